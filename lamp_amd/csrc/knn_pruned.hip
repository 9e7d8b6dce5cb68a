// knn_pruned.hip — exact 1-NN correspondence search that skips provably irrelevant targets.
//
// Same contract as the exhaustive kernel in knn.hip (reference: the kd-tree lookups of
// lamp_utils/include/lamp_utils/gicp.hpp:463-498 and the two lookup passes at
// loop_closure/src/IcpLoopComputation.cc:553-566,586): FLANN L2_Simple<float> arithmetic
// d = (dx*dx + dy*dy) + dz*dz with every operation rounded separately, lowest target index
// among exactly equidistant targets.  What changes is how much of the target is looked at.
//
// Layout.  Every cloud carries a spatial index built once (spatial_index.hip): its points in
// k-d partition order cut into tiles of SI_TILE points, each point stored with its original index,
// plus the axis-aligned box of every tile and the same visiting order of the cloud's own
// points.  One wave owns 64 consecutive queries of that order — a compact patch of surface — and keeps
// the box of those of them that still search (all of them in the first passes, see
// "Certificates").  The float box-to-box gap, squared and summed in the same operation order as
// a point distance, never exceeds the computed distance of any query/target pair drawn from
// the two boxes (rounding is monotonic), so a tile whose gap exceeds every lane's current best
// cannot improve or tie any lane and is skipped without touching its points.
//
// Sweep.  Tiles are visited in rounds of doubling radius: round k looks at tiles whose gap to
// the patch lies in (r_{k-1}^2, r_k^2]; the sweep ends when every lane's best is <= r_k^2
// (nothing further out can matter) or, for the bounded search of the GICP loop, when r_k^2
// reaches the correspondence threshold (the loop ignores matches at or beyond it,
// gicp.hpp:476).  A round first tests the boxes of the blocks of 64 consecutive tiles, 64 blocks at
// once, and goes into the blocks that can hold a tile of the round.  Inside a block the 64 lanes
// test its 64 tile boxes at once; for each survivor the
// box is broadcast and every lane tests its own query against it, and the tile is walked only
// if some lane can still improve or tie.  Tile coordinates arrive through scalar loads, two
// targets per packed instruction.  As in the exhaustive kernel a lane tracks only the minimum of each group
// of 8 targets and the first group that lowered it; the index is recovered afterwards, and the
// rare exact tie between groups is resolved by a second walk.
//
// Certificates.  Late in the GICP loop the transformation barely moves and almost every query
// keeps its match.  A sweep therefore also leaves, per query, the position it was made from and a
// lower bound of the distance from there to every target other than the match: the second
// smallest distance it evaluated, the gaps of the tiles it ruled out for the lane, and the
// radius beyond which the wave looked at nothing.  In a later pass a query that has moved by
// delta and measures d to its old match keeps that match for certain when
// sqrt(d) + delta < bound (triangle inequality, with a margin far above the rounding of the
// float distance); such lanes are answered at once and take no part in the sweep, which then
// only visits the tiles the remaining lanes need -- none at all when every lane is certain.
#include "lgicp_device.h"

namespace lgicp {

namespace {

// max over the wave of a non-negative float (bit patterns order like the values)
__device__ __forceinline__ float wave_max_nonneg(float v) {
  unsigned u = __float_as_uint(v);
  // row_shr 1,2,4,8 within rows of 16, then row_bcast 15 and 31 across rows
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x111, 0xf, 0xf, false));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x112, 0xf, 0xf, false));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x114, 0xf, 0xf, false));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x118, 0xf, 0xf, false));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x142, 0xa, 0xf, false));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x143, 0xc, 0xf, false));
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 63));
}

__device__ __forceinline__ float wave_min_f(float v) {  // NaN lanes are ignored by fminf
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off));
  return v;
}

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int GROUP = 8;  // targets whose minimum is tracked together
// Tiles are selected up to this factor beyond the squared distance the sweep needs for an exact
// answer: without it the lane with the largest best distance of a wave would get a certificate
// with no margin at all, every time.
constexpr float WIDEN = 2.25f;

// tile coordinates are read through the constant address space: with wave-uniform addresses
// the compiler fetches them with scalar loads and feeds them to the vector ALU as scalar
// operands, so a target costs no vector register, LDS bandwidth or vector-memory instruction
typedef const __attribute__((address_space(4))) float cfloat;

__device__ __forceinline__ float rdlane(float v, int l) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

}  // namespace

__global__ __launch_bounds__(NNP_THREADS) void nn_pruned_kernel(
    const CloudDev* __restrict__ clouds, const PairDev* __restrict__ pairs,
    const PairState* __restrict__ st, const int4* __restrict__ bmap, int final_pass, float cap,
    float r0sq, int use_seed, unsigned long long* __restrict__ counters) {
  const int4 bm = bmap[blockIdx.x];
  const int pair = bm.x;
  const PairState& S = st[pair];
  if (!final_pass && !S.active) return;  // block-uniform
  const PairDev& P = pairs[pair];
  const CloudDev& TG = clouds[P.tgt_cloud];
  const CloudDev& SC = clouds[P.src_cloud];
  const int lane = threadIdx.x & 63;
  const int slot = (bm.y * (NNP_THREADS / 64) + (threadIdx.x >> 6)) * 64 + lane;
  if (slot - lane >= SC.n_q) return;  // wave-uniform: this wave has no queries
  const float4* __restrict__ qsrc = final_pass ? SC.xyz : P.out;
  const float* Tm = final_pass ? S.Tfinal : S.T;

  const int qi = slot < SC.n_q ? SC.qorder[slot] : -1;
  float x = __builtin_nanf(""), y = x, z = x;
  if (qi >= 0) {
    const float4 p = qsrc[qi];
    x = ((Tm[0] * p.x + Tm[1] * p.y) + Tm[2] * p.z) + Tm[3];
    y = ((Tm[4] * p.x + Tm[5] * p.y) + Tm[6] * p.z) + Tm[7];
    z = ((Tm[8] * p.x + Tm[9] * p.y) + Tm[10] * p.z) + Tm[11];
  }
  const bool finite = __builtin_isfinite(x) && __builtin_isfinite(y) && __builtin_isfinite(z);
  // optional; kept in the order of the query slots (this kernel is its only reader and writer):
  // 64 consecutive 16-byte entries per wave instead of a gather through qorder
  float4* const cert = P.cert;
  if (__ballot(finite) == 0) {  // nothing to look up
    if (qi >= 0) {
      P.keys[qi] = NNKEY_NONE;
      if (cert) cert[slot] = make_float4(x, y, z, 0.f);  // bound 0: never certain
    }
    return;
  }
  const v2f x2 = {x, x}, y2 = {y, y}, z2 = {z, z};

  const float4* __restrict__ tmin = TG.tmin;
  const float4* __restrict__ tmax = TG.tmax;
  const int n_tiles = TG.n_tiles;
  cfloat* const tiles = (cfloat*)(uintptr_t)TG.sxyz;

  float bestd = __builtin_inff();  // smallest finite distance so far
  int bgrp = -1;                   // first group (tile * 4 + k) that reached it
  bool tie = false;                // another group reached exactly the same distance
  // Seed: the match of the previous pass, re-measured under the current transformation, is an
  // upper bound of this pass's nearest distance.  Tiles that cannot reach it are then dropped by
  // the per-lane test before any of their points is looked at.  The seed's own tile always
  // passes that test, so the sweep re-finds the seed (or something at least as close) itself.
  // need: the lane takes part in the sweep.  A lane whose certificate still holds is answered
  // from it (kept) and only rides along.
  bool need = finite;
  nnkey_t kept = NNKEY_NONE;
  if (use_seed && finite) {
    const nnkey_t prev = P.keys[qi];
    const unsigned pi = (unsigned)(prev & 0xFFFFFFFFu);
    const bool matched = prev != NNKEY_NONE && pi < (unsigned)TG.n;
    float d = __builtin_inff();
    if (matched) {
      const float4 p = TG.xyz[pi];
      const float dx = x - p.x, dy = y - p.y, dz = z - p.z;
      d = (dx * dx + dy * dy) + dz * dz;
      if (d <= cap) bestd = d;  // false for NaN
    }
    if (cert) {
      // c.xyz: where the query stood when the bound c.w was established
      const float4 c = cert[slot];
      const float mx = x - c.x, my = y - c.y, mz = z - c.z;
      const float moved = __builtin_amdgcn_sqrtf((mx * mx + my * my) + mz * mz);
      // the float distance arithmetic is off by a few 1e-7 relative; 1e-5 and an absolute 1e-6
      // (squares far from underflow) leave two orders of magnitude to spare.  NaNs fail.
      if (matched) {
        const float reach = (__builtin_amdgcn_sqrtf(d) + moved) * 1.00001f + 1.0e-6f;
        if (reach < c.w) need = false, kept = ((nnkey_t)__float_as_uint(d) << 32) | pi;
      } else if (prev == NNKEY_NONE && !final_pass) {
        // nothing within the cap last time and nothing can have come within it since
        const float reach = (__builtin_amdgcn_sqrtf(cap) + moved) * 1.00001f + 1.0e-6f;
        if (reach < c.w) need = false;
      }
    }
  }
  if (__ballot(need) == 0) {  // every lane is certain (or has no query)
    if (qi >= 0) P.keys[qi] = kept;
    return;
  }
  // the box of the queries that take part in the sweep: the whole patch in the first passes, a
  // few scattered queries later on (tiles are then selected for those alone)
  const float nanv = __builtin_nanf("");
  const float fx = need ? x : nanv, fy = need ? y : nanv, fz = need ? z : nanv;
  const float qx0 = wave_min_f(fx), qy0 = wave_min_f(fy), qz0 = wave_min_f(fz);
  const float qx1 = wave_max_f(fx), qy1 = wave_max_f(fy), qz1 = wave_max_f(fz);
  unsigned visited = 0;
  float second = __builtin_inff();  // lower bound of the evaluated distances other than the best
  float ruled = __builtin_inff();   // smallest gap of a tile that was ruled out without a walk
  // wave-uniform: max over the sweeping lanes of bestd, refreshed after every round.  With every
  // lane seeded the first round already reaches as far as any lane can need.
  float worst = wave_max_nonneg(need ? bestd : 0.f);
  float wround = worst;  // the value of `worst` the current round's tile selection used
  float lo = -1.0f;
  // tiles are looked at somewhat beyond the cap as well (their gaps only, never their points): a
  // query with nothing inside the cap then learns how far away the nearest candidate tile is,
  // and can be certain later on that still nothing has come within the cap
  const float capsel = cap * WIDEN;
  float hi = __builtin_fminf(worst < __builtin_inff() ? __builtin_fmaxf(r0sq, worst * WIDEN) : r0sq, capsel);
  const int n_steps = (n_tiles + 63) / 64;

  const float4* __restrict__ const gmin = tmin + n_tiles;  // boxes of the blocks of 64 tiles
  const float4* __restrict__ const gmax = tmax + n_tiles;
  for (;;) {
    wround = worst;
    const float sel = __builtin_fminf(hi, wround * WIDEN);  // no tile beyond this gap is selected
    // The blocks of 64 consecutive tiles first, 64 of them at once: a block whose box lies beyond
    // `sel` holds no tile of this round (its gap bounds the gap of every tile inside from below),
    // and in k-d order most blocks of a cloud are far from the patch.
    for (int sb0 = 0; sb0 < n_steps; sb0 += 64) {
      bool take = false;
      if (sb0 + lane < n_steps) {
        const float4 g0 = gmin[sb0 + lane], g1 = gmax[sb0 + lane];
        const float gx = __builtin_fmaxf(0.f, __builtin_fmaxf(g0.x - qx1, qx0 - g1.x));
        const float gy = __builtin_fmaxf(0.f, __builtin_fmaxf(g0.y - qy1, qy0 - g1.y));
        const float gz = __builtin_fmaxf(0.f, __builtin_fmaxf(g0.z - qz1, qz0 - g1.z));
        take = (gx * gx + gy * gy) + gz * gz <= sel;  // false for an empty block (gap +inf)
      }
      unsigned long long smask = __ballot(take);
      // tile boxes of the next block to visit are fetched while the current one is worked on
      int step = -1;
      float4 nb0 = make_float4(0.f, 0.f, 0.f, 0.f), nb1 = nb0;
      if (smask) {
        step = sb0 + __builtin_ctzll(smask);
        smask &= smask - 1;
        if (step * 64 + lane < n_tiles) nb0 = tmin[step * 64 + lane], nb1 = tmax[step * 64 + lane];
      }
      while (step >= 0) {
        const float4 b0 = nb0, b1 = nb1;
        const int t = step * 64 + lane;
        const int cur = step;
        step = -1;
        if (smask) {
          step = sb0 + __builtin_ctzll(smask);
          smask &= smask - 1;
          if (step * 64 + lane < n_tiles) nb0 = tmin[step * 64 + lane], nb1 = tmax[step * 64 + lane];
        }
        // patch-to-tile gap: which of these 64 tiles belong to this round at all
        const float gx = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.x - qx1, qx0 - b1.x));
        const float gy = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.y - qy1, qy0 - b1.y));
        const float gz = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.z - qz1, qz0 - b1.z));
        const float lb = (gx * gx + gy * gy) + gz * gz;
        unsigned long long mask = __ballot(t < n_tiles && lb > lo && lb <= hi && lb <= wround * WIDEN);
        while (mask) {
          const int b = __builtin_ctzll(mask);
          mask &= mask - 1;
          // query-to-tile gap per lane: the tile is skipped unless it can improve or tie some
          // lane's current best (and, in the bounded search, lies inside the cap for it)
          const float x0 = rdlane(b0.x, b), y0 = rdlane(b0.y, b), z0 = rdlane(b0.z, b);
          const float x1 = rdlane(b1.x, b), y1 = rdlane(b1.y, b), z1 = rdlane(b1.z, b);
          const float hx = __builtin_fmaxf(0.f, __builtin_fmaxf(x0 - x, x - x1));
          const float hy = __builtin_fmaxf(0.f, __builtin_fmaxf(y0 - y, y - y1));
          const float hz = __builtin_fmaxf(0.f, __builtin_fmaxf(z0 - z, z - z1));
          const float gl = (hx * hx + hy * hy) + hz * hz;
#ifdef NNP_COUNT_CONSIDERED  // diagnostics build: the counter reports tiles tested, not tiles walked
          ++visited;
#endif
          if (__ballot(need && gl <= bestd && gl <= cap) == 0) {
            ruled = __builtin_fminf(ruled, gl);
            continue;
          }
          const int tile = cur * 64 + b;
          cfloat* const tx = tiles + (size_t)tile * (4 * SI_TILE);
          cfloat* const ty = tx + SI_TILE;
          cfloat* const tz = tx + 2 * SI_TILE;
#pragma unroll
          for (int g = 0; g < SI_TILE / GROUP; ++g) {
            float m = __builtin_inff();
#pragma unroll
            for (int j = 0; j < GROUP; j += 2) {
              const int kk = g * GROUP + j;
              const v2f px = {tx[kk], tx[kk + 1]}, py = {ty[kk], ty[kk + 1]}, pz = {tz[kk], tz[kk + 1]};
              const v2f dx = x2 - px, dy = y2 - py, dz = z2 - pz;
              const v2f d = (dx * dx + dy * dy) + dz * dz;
              m = __builtin_fminf(__builtin_fminf(m, d.x), d.y);
            }
            // +inf and NaN distances (padding, overflow) never count
            // a seeded bound counts as reached by the first group that equals it
            const bool lower = m < bestd || (m == bestd && bgrp < 0);
            // the loser of the comparison is a target other than the final match (a seeded bound
            // that no group has reached yet is the match itself or shows up again later)
            second = __builtin_fminf(second, lower ? (bgrp >= 0 ? bestd : __builtin_inff()) : m);
            tie = lower ? false : (tie || m == bestd);
            bestd = lower ? m : bestd;
            bgrp = lower ? tile * (SI_TILE / GROUP) + g : bgrp;
          }
#ifndef NNP_COUNT_CONSIDERED
          ++visited;
#endif
        }
      }
    }
    worst = wave_max_nonneg(need ? bestd : 0.f);
    if (hi >= capsel || worst * WIDEN <= hi) break;
    lo = hi;
    hi = __builtin_fminf(hi * 4.0f, capsel);
    if (!(hi < 1.0e30f)) hi = capsel;
  }

  // lowest original index inside the winning group
  const int* __restrict__ tidx = reinterpret_cast<const int*>(TG.sxyz);
  const float* __restrict__ tf = reinterpret_cast<const float*>(TG.sxyz);
  int idx = 0x7FFFFFFF;
  if (finite && bgrp >= 0) {
    const size_t o = (size_t)(bgrp / (SI_TILE / GROUP)) * (4 * SI_TILE) + (bgrp % (SI_TILE / GROUP)) * GROUP;
    int n_best = 0;
    float others = __builtin_inff();
#pragma unroll
    for (int j = 0; j < GROUP; ++j) {
      const float dx = x - tf[o + j], dy = y - tf[o + SI_TILE + j], dz = z - tf[o + 2 * SI_TILE + j];
      const float d = (dx * dx + dy * dy) + dz * dz;
      const int id = tidx[o + 3 * SI_TILE + j];
      idx = (d == bestd && id < idx) ? id : idx;
      n_best += d == bestd;
      others = d == bestd ? others : __builtin_fminf(others, d);
    }
    second = __builtin_fminf(second, n_best > 1 ? bestd : others);
  }
  // Rare: another group holds a target at exactly the same distance.  Re-walk every tile that
  // can contain one and take the lowest index among the exact matches.
  tie = tie && finite && bgrp >= 0;
  if (__ballot(tie)) {
    const float tworst = wave_max_nonneg(tie ? bestd : 0.f);
    for (int base = 0; base < n_tiles; base += 64) {
      const int t = base + lane;
      float lb = __builtin_inff();
      if (t < n_tiles) {
        const float4 b0 = tmin[t], b1 = tmax[t];
        const float gx = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.x - qx1, qx0 - b1.x));
        const float gy = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.y - qy1, qy0 - b1.y));
        const float gz = __builtin_fmaxf(0.f, __builtin_fmaxf(b0.z - qz1, qz0 - b1.z));
        lb = (gx * gx + gy * gy) + gz * gz;
      }
      unsigned long long mask = __ballot(t < n_tiles && lb <= tworst);
      while (mask) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1;
        const size_t o = (size_t)(base + b) * (4 * SI_TILE);
        for (int j = 0; j < SI_TILE; ++j) {
          const float dx = x - tf[o + j], dy = y - tf[o + SI_TILE + j], dz = z - tf[o + 2 * SI_TILE + j];
          const float d = (dx * dx + dy * dy) + dz * dz;
          const int id = tidx[o + 3 * SI_TILE + j];
          idx = (tie && d == bestd && id >= 0 && id < idx) ? id : idx;
        }
      }
    }
  }

  if (qi >= 0) {
    const bool found = finite && bgrp >= 0 && idx != 0x7FFFFFFF;
    if (need || !finite) {
      P.keys[qi] = found ? (((nnkey_t)__float_as_uint(bestd) << 32) | (unsigned)idx) : NNKEY_NONE;
      if (cert) {
        // Every target other than the match was either evaluated (second), lies in a tile ruled
        // out for this lane (ruled), or in a tile the wave never selected: those are farther
        // from the whole patch than the round's radius or the round's largest best distance.
        // Without a match the bound covers every target.
        float lb = __builtin_fminf(__builtin_fminf(second, ruled), __builtin_fminf(hi, wround * WIDEN));
        if (!found) lb = __builtin_fminf(lb, bestd);
        cert[slot] = make_float4(x, y, z, finite ? __builtin_amdgcn_sqrtf(lb) * 0.99999f : 0.f);
      }
    } else {
      P.keys[qi] = kept;
    }
  }
  // spread over NN_COUNTER_SLOTS addresses: one hot address would serialise every wave of the
  // launch in a single L2 channel
  if (counters && lane == 0)
    atomicAdd(&counters[(blockIdx.x * (NNP_THREADS / 64) + (threadIdx.x >> 6)) & (NN_COUNTER_SLOTS - 1)],
              (unsigned long long)visited * SI_TILE * 64ull);
}

hipError_t launch_nn_pruned(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                            const int4* bmap, int n_blocks, bool final_pass, float cap, float r0sq,
                            bool use_seed, unsigned long long* counters, hipStream_t stream) {
  if (n_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(nn_pruned_kernel, dim3(n_blocks), dim3(NNP_THREADS), 0, stream, clouds, pairs,
                     st, bmap, (int)final_pass, cap, r0sq, (int)use_seed, counters);
  return hipGetLastError();
}

}  // namespace lgicp
