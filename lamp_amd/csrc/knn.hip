// knn.hip — tiled exhaustive 1-NN correspondence search for gfx950.
//
// Replaces the OpenMP kd-tree lookup loop of the reference
// (lamp_utils/include/lamp_utils/gicp.hpp:463-498, gicp.h:390-397) and the two serial
// lookup passes after the alignment (pcl getFitnessScore called at
// loop_closure/src/IcpLoopComputation.cc:586 and the correspondence pass :553-566).
//
// Arithmetic contract (bit-exact with the reference's FLANN L2_Simple<float> distance on an
// x86-64 build without FMA): q = ((m0*x + m1*y) + m2*z) + m3 per row, d = (dx*dx + dy*dy)
// + dz*dz, every operation rounded separately (this file is compiled with
// -ffp-contract=off).  Among exactly equidistant targets the lowest index wins.
//
// Mapping: one workgroup = 256 lanes x Q queries held in registers; the target cloud streams
// through LDS in tiles of NN_TILE points stored as three float arrays, read back with
// wave-uniform (broadcast) ds_read_b128.  A lane never compares indices in the hot loop: it
// keeps the minimum distance of each group of NN_GROUP targets with v_min3_f32 and remembers
// the first group that lowered its running minimum; the exact index is recovered afterwards by
// recomputing that one group.  When the batch is too small to fill the chip the target is
// split over several workgroups and the partial results merge through a 64-bit atomic min of
// (distance bits, index).
#include "lgicp_device.h"

namespace lgicp {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float tx, float ty,
                                       float tz) {
  const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
  return (dx * dx + dy * dy) + dz * dz;
}

template <int Q>
__global__ __launch_bounds__(NN_THREADS) void nn_kernel(const CloudDev* __restrict__ clouds,
                                                        const PairDev* __restrict__ pairs,
                                                        const PairState* __restrict__ st,
                                                        const int4* __restrict__ bmap,
                                                        int final_pass) {
  static_assert(Q % 2 == 0, "queries are processed in packed pairs");
  __shared__ __attribute__((aligned(16))) float lds[3 * NN_TILE];
  float* const lx = lds;
  float* const ly = lds + NN_TILE;
  float* const lz = lds + 2 * NN_TILE;

  const int4 bm = bmap[blockIdx.x];
  const int pair = bm.x;
  const PairState& S = st[pair];
  if (!final_pass && !S.active) return;  // wave-uniform
  const PairDev& P = pairs[pair];
  const CloudDev& TG = clouds[P.tgt_cloud];
  const float4* __restrict__ tgt = TG.xyz;
  const int n_src = P.n_src;
  const float4* __restrict__ qsrc = final_pass ? clouds[P.src_cloud].xyz : P.out;
  const float* Tm = final_pass ? S.Tfinal : S.T;
  float t[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) t[i] = Tm[i];

  const int tid = threadIdx.x;
  const int qbase = bm.y * (NN_THREADS * Q);
  v2f qx[Q / 2], qy[Q / 2], qz[Q / 2];
  float best[Q];
  int bgrp[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * NN_THREADS + tid;
    float x = __builtin_nanf(""), y = x, z = x;
    if (i < n_src) {
      const float4 p = qsrc[i];
      x = ((t[0] * p.x + t[1] * p.y) + t[2] * p.z) + t[3];
      y = ((t[4] * p.x + t[5] * p.y) + t[6] * p.z) + t[7];
      z = ((t[8] * p.x + t[9] * p.y) + t[10] * p.z) + t[11];
    }
    qx[q / 2][q % 2] = x;
    qy[q / 2][q % 2] = y;
    qz[q / 2][q % 2] = z;
    best[q] = __builtin_inff();
    bgrp[q] = -1;
  }

  const int n_tiles = TG.n_pad / NN_TILE;
  const int tile0 = (int)(((long long)n_tiles * bm.z) / bm.w);
  const int tile1 = (int)(((long long)n_tiles * (bm.z + 1)) / bm.w);

  for (int tile = tile0; tile < tile1; ++tile) {
    __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
    for (int k = 0; k < NN_TILE / NN_THREADS; ++k) {
      const int j = k * NN_THREADS + tid;
      const float4 p = tgt[(size_t)tile * NN_TILE + j];
      lx[j] = p.x;
      ly[j] = p.y;
      lz[j] = p.z;
    }
    __syncthreads();
    const int gbase = tile * (NN_TILE / NN_GROUP);
#pragma unroll 2
    for (int g = 0; g < NN_TILE / NN_GROUP; ++g) {
      // wave-uniform addresses: each ds_read_b128 is a broadcast
      const float4 xa = *reinterpret_cast<const float4*>(lx + g * NN_GROUP);
      const float4 xb = *reinterpret_cast<const float4*>(lx + g * NN_GROUP + 4);
      const float4 ya = *reinterpret_cast<const float4*>(ly + g * NN_GROUP);
      const float4 yb = *reinterpret_cast<const float4*>(ly + g * NN_GROUP + 4);
      const float4 za = *reinterpret_cast<const float4*>(lz + g * NN_GROUP);
      const float4 zb = *reinterpret_cast<const float4*>(lz + g * NN_GROUP + 4);
      const float tx[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      const float ty[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
      const float tz[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
      for (int p = 0; p < Q / 2; ++p) {
        v2f d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const v2f dx = qx[p] - tx[j];
          const v2f dy = qy[p] - ty[j];
          const v2f dz = qz[p] - tz[j];
          d[j] = (dx * dx + dy * dy) + dz * dz;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float m = __builtin_fminf(__builtin_fminf(d[0][h], d[1][h]), d[2][h]);
          m = __builtin_fminf(__builtin_fminf(m, d[3][h]), d[4][h]);
          m = __builtin_fminf(__builtin_fminf(m, d[5][h]), d[6][h]);
          m = __builtin_fminf(m, d[7][h]);
          const int q = 2 * p + h;
          const bool lower = m < best[q];
          best[q] = lower ? m : best[q];
          bgrp[q] = lower ? gbase + g : bgrp[q];
        }
      }
    }
  }

  // recover the index inside the winning group (first target that reproduces the minimum)
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * NN_THREADS + tid;
    if (i >= n_src) continue;
    nnkey_t key = NNKEY_NONE;
    if (bgrp[q] >= 0) {
      const float x = qx[q / 2][q % 2], y = qy[q / 2][q % 2], z = qz[q / 2][q % 2];
      int idx = -1;
#pragma unroll
      for (int j = NN_GROUP - 1; j >= 0; --j) {
        const float4 p = tgt[(size_t)bgrp[q] * NN_GROUP + j];
        const float d = dist2(x, y, z, p.x, p.y, p.z);
        idx = (d == best[q]) ? bgrp[q] * NN_GROUP + j : idx;
      }
      if (idx >= 0 && idx < TG.n)
        key = ((nnkey_t)__float_as_uint(best[q]) << 32) | (unsigned)idx;
    }
    if (bm.w == 1)
      P.keys[i] = key;
    else if (key != NNKEY_NONE)
      atomicMin(&P.keys[i], key);
  }
}

int nn_queries_per_block(int variant) {
  switch (variant) {
    case 2: return NN_THREADS * 2;
    case 8: return NN_THREADS * 8;
    default: return NN_THREADS * 4;
  }
}

hipError_t launch_nn_search(const CloudDev* clouds, const PairDev* pairs, const PairState* st,
                            const int4* bmap, int n_blocks, bool final_pass, bool any_split,
                            int variant, hipStream_t stream) {
  (void)any_split;
  if (n_blocks <= 0) return hipSuccess;
  const dim3 grid(n_blocks), block(NN_THREADS);
  switch (variant) {
    case 2:
      hipLaunchKernelGGL(nn_kernel<2>, grid, block, 0, stream, clouds, pairs, st, bmap,
                         (int)final_pass);
      break;
    case 8:
      hipLaunchKernelGGL(nn_kernel<8>, grid, block, 0, stream, clouds, pairs, st, bmap,
                         (int)final_pass);
      break;
    default:
      hipLaunchKernelGGL(nn_kernel<4>, grid, block, 0, stream, clouds, pairs, st, bmap,
                         (int)final_pass);
      break;
  }
  return hipGetLastError();
}

// keys <- NONE before a pass whose target is split over several workgroups
__global__ void keys_fill_kernel(const PairDev* __restrict__ pairs,
                                 const PairState* __restrict__ st, int final_pass) {
  const int pair = blockIdx.y;
  if (!final_pass && !st[pair].active) return;
  const PairDev& P = pairs[pair];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n_src; i += gridDim.x * blockDim.x)
    P.keys[i] = NNKEY_NONE;
}

hipError_t launch_keys_fill(const PairDev* pairs, const PairState* st, int n_pairs, int max_n,
                            bool final_pass, hipStream_t stream) {
  if (n_pairs <= 0 || max_n <= 0) return hipSuccess;
  int bx = (max_n + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(keys_fill_kernel, dim3(bx, n_pairs), dim3(256), 0, stream, pairs, st,
                     (int)final_pass);
  return hipGetLastError();
}

__global__ void keys_unpack_kernel(const nnkey_t* __restrict__ keys, int n,
                                   int32_t* __restrict__ idx, float* __restrict__ d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const nnkey_t k = keys[i];
  if (k == NNKEY_NONE) {
    idx[i] = -1;
    d2[i] = __builtin_inff();
  } else {
    idx[i] = (int32_t)(k & 0xFFFFFFFFu);
    d2[i] = __uint_as_float((unsigned)(k >> 32));
  }
}

hipError_t launch_keys_unpack(const nnkey_t* keys, int n, int32_t* idx, float* d2,
                              hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(keys_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n,
                     idx, d2);
  return hipGetLastError();
}

}  // namespace lgicp
