// lgicp_primitives.hip — kernel-level entry points of the C ABI (host buffers in and out).
// Each one uploads its operands, runs exactly the kernel the batch path runs, and downloads
// the result; they exist so that every kernel can be checked on its own.
#include <algorithm>
#include <cstring>

#include "lgicp_internal.h"
#include "sac_ia.h"

namespace {

struct Dev {  // frees everything it handed out
  std::vector<void*> ptrs;
  ~Dev() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  T* alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T*)p;
  }
};

inline int round_up(int n, int m) { return ((n + m - 1) / m) * m; }

// packed xyz -> padded float4 on the device
float4* upload_points(Dev& d, const float* xyz, size_t n, int n_pad, float w) {
  std::vector<float4> h((size_t)n_pad);
  for (size_t i = 0; i < n; ++i) h[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], w);
  for (size_t i = n; i < (size_t)n_pad; ++i) h[i] = make_float4(PAD_COORD, PAD_COORD, PAD_COORD, w);
  float4* p = d.alloc<float4>((size_t)n_pad);
  if (!p) return nullptr;
  if (hipMemcpy(p, h.data(), sizeof(float4) * (size_t)n_pad, hipMemcpyHostToDevice) != hipSuccess)
    return nullptr;
  return p;
}

void expand_cov(const std::vector<double>& c6, size_t n, double* cov9) {
  for (size_t i = 0; i < n; ++i) {
    const double* c = &c6[6 * i];
    double* o = cov9 + 9 * i;
    o[0] = c[0], o[1] = c[1], o[2] = c[2];
    o[3] = c[1], o[4] = c[3], o[5] = c[4];
    o[6] = c[2], o[7] = c[4], o[8] = c[5];
  }
}

}  // namespace

extern "C" {

int lgicp_nn_search(lgicp_ctx* ctx, const float T[16], const float* src_xyz, size_t n_src,
                    const float* tgt_xyz, size_t n_tgt, int32_t* idx, float* d2, int n_splits) {
  if (!ctx || !T || (n_src && (!src_xyz || !idx || !d2)) || (n_tgt && !tgt_xyz))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n_src > (size_t)1 << 28 || n_tgt > (size_t)1 << 28)
    return fail(ctx, LGICP_E_INVALID, "cloud too large");
  if (n_src == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = ctx->stream;
  Dev d;
  // both clouds go up like keyframes (padded float4 points + spatial index, nothing derived)
  const KfInput in[2] = {{0, src_xyz, nullptr, n_src, 3}, {1, tgt_xyz, nullptr, n_tgt, 3}};
  Keyframe kf[2];
  {
    const int rc = register_keyframes(ctx, in, 2, kf, /*derive=*/false);
    if (rc != LGICP_OK) {
      (void)hipStreamSynchronize(st);
      return rc;
    }
  }
  const int n_pad_t = kf[1].n_pad;
  float4* d_src = kf[0].d_xyz;
  nnkey_t* d_keys = d.alloc<nnkey_t>(n_src);
  CloudDev* d_clouds = d.alloc<CloudDev>(2);
  PairDev* d_pair = d.alloc<PairDev>(1);
  PairState* d_state = d.alloc<PairState>(1);
  int32_t* d_idx = d.alloc<int32_t>(n_src);
  float* d_d2 = d.alloc<float>(n_src);
  if (!d_keys || !d_clouds || !d_pair || !d_state || !d_idx || !d_d2) {
    (void)hipStreamSynchronize(st);
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  }
  CloudDev hc[2];
  std::memset(hc, 0, sizeof(hc));
  for (int c = 0; c < 2; ++c) {
    hc[c].xyz = kf[c].d_xyz, hc[c].n = kf[c].n, hc[c].n_pad = kf[c].n_pad;
    hc[c].sxyz = kf[c].d_sxyz, hc[c].tmin = kf[c].d_tmin, hc[c].tmax = kf[c].d_tmax;
    hc[c].qorder = kf[c].d_qorder, hc[c].n_tiles = kf[c].n_tiles, hc[c].n_q = kf[c].n_q;
  }
  PairDev hp;
  std::memset(&hp, 0, sizeof(hp));
  hp.src_cloud = 0, hp.tgt_cloud = 1, hp.scan1_cloud = 0, hp.scan2_cloud = 1;
  hp.out = d_src, hp.keys = d_keys, hp.n_src = (int)n_src, hp.n_tgt = (int)n_tgt;
  PairState hs;
  std::memset(&hs, 0, sizeof(hs));
  std::memcpy(hs.T, T, sizeof(hs.T));
  hs.active = 1;
  const int QB = ctx->nn_exhaustive ? nn_queries_per_block(ctx->nn_variant) : NNP_THREADS;
  const int qt = ((int)n_src + QB - 1) / QB, tiles = n_pad_t / NN_TILE;
  int sp = n_splits;
  if (!ctx->nn_exhaustive) {
    sp = 1;
  } else if (sp <= 0) {
    sp = 1;
    while ((long long)qt * sp < 1024 && sp < 64) sp *= 2;
  }
  sp = std::max(1, std::min(sp, tiles));
  std::vector<int4> bmap;
  for (int t = 0; t < qt; ++t)
    for (int s = 0; s < sp; ++s) bmap.push_back(make_int4(0, t, s, sp));
  int4* d_bmap = d.alloc<int4>(bmap.size());
  if (!d_bmap) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, hipMemcpy(d_clouds, hc, sizeof(hc), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_pair, &hp, sizeof(hp), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_bmap, bmap.data(), sizeof(int4) * bmap.size(), hipMemcpyHostToDevice));
  if (!ctx->nn_exhaustive) {
    HIP_TRY(ctx, launch_nn_pruned(d_clouds, d_pair, d_state, d_bmap, (int)bmap.size(), false,
                                  __builtin_inff(), ctx->nn_r0sq, false, nullptr, st));
  } else {
    if (sp > 1) HIP_TRY(ctx, launch_keys_fill(d_pair, d_state, 1, (int)n_src, false, st));
    HIP_TRY(ctx, launch_nn_search(d_clouds, d_pair, d_state, d_bmap, (int)bmap.size(), false,
                                  sp > 1, ctx->nn_variant, st));
  }
  HIP_TRY(ctx, launch_keys_unpack(d_keys, (int)n_src, d_idx, d_d2, st));
  HIP_TRY(ctx, hipMemcpyAsync(idx, d_idx, sizeof(int32_t) * n_src, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(d2, d_d2, sizeof(float) * n_src, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return LGICP_OK;
}

int lgicp_covariances_from_normals(lgicp_ctx* ctx, const float* normals, size_t n, double* cov9) {
  if (!ctx || (n && (!normals || !cov9))) return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  float4* d_n = upload_points(d, normals, n, (int)n, 0.f);
  double* d_c = d.alloc<double>(6 * n);
  if (!d_n || !d_c) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, launch_cov_from_normals(d_n, (int)n, d_c, ctx->stream));
  std::vector<double> h(6 * n);
  HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_c, sizeof(double) * 6 * n, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  expand_cov(h, n, cov9);
  return LGICP_OK;
}

int lgicp_covariances_knn(lgicp_ctx* ctx, const float* xyz, size_t n, int k, double eps,
                          double* cov9) {
  if (!ctx || (n && (!xyz || !cov9))) return fail(ctx, LGICP_E_INVALID, "null argument");
  if (k < 1 || k > 32) return fail(ctx, LGICP_E_INVALID, "k must be in 1..32");
  if ((size_t)k > n)  // the reference leaves the covariances empty (gicp.hpp:72-79)
    return fail(ctx, LGICP_E_INVALID, "cloud has fewer points (%zu) than k (%d)", n, k);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  const int n_pad = std::max(NN_TILE, round_up((int)n, NN_TILE));
  float4* d_x = upload_points(d, xyz, n, n_pad, 1.f);
  double* d_c = d.alloc<double>(6 * n);
  if (!d_x || !d_c) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, launch_cov_knn(d_x, (int)n, n_pad, k, eps, d_c, ctx->stream));
  std::vector<double> h(6 * n);
  HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_c, sizeof(double) * 6 * n, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  expand_cov(h, n, cov9);
  return LGICP_OK;
}

int lgicp_compute_ap(lgicp_ctx* ctx, const float* query_xyz, size_t n_query,
                     const float* ref_normals, size_t n_ref, const int32_t* corr, size_t n_corr,
                     const float T[16], int normalize_query, double Ap[36]) {
  if (!ctx || !T || !Ap || (n_query && !query_xyz) || (n_ref && !ref_normals) ||
      (n_corr && !corr))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  for (int i = 0; i < 36; ++i) Ap[i] = 0.0;
  const size_t n = std::min(n_query, n_corr);
  if (n == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  hipStream_t st = ctx->stream;
  Dev d;
  float4* d_q = upload_points(d, query_xyz, n_query, (int)n_query, 1.f);
  float4* d_n = upload_points(d, ref_normals, n_ref, (int)std::max<size_t>(n_ref, 1), 0.f);
  std::vector<nnkey_t> hk(n);
  for (size_t i = 0; i < n; ++i)
    hk[i] = corr[i] < 0 ? NNKEY_NONE : (nnkey_t)(unsigned)corr[i];
  nnkey_t* d_keys = d.alloc<nnkey_t>(n);
  CloudDev* d_clouds = d.alloc<CloudDev>(2);
  PairDev* d_pair = d.alloc<PairDev>(1);
  PairState* d_state = d.alloc<PairState>(1);
  int4* d_bmap = d.alloc<int4>(RED_MAX_CHUNKS);
  int* d_ids = d.alloc<int>(1);
  double* d_part = d.alloc<double>((size_t)RED_MAX_CHUNKS * NPOST);
  if (!d_q || !d_n || !d_keys || !d_clouds || !d_pair || !d_state || !d_bmap || !d_ids || !d_part)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  CloudDev hc[2];
  std::memset(hc, 0, sizeof(hc));
  hc[0].xyz = d_q, hc[0].n = (int)n_query, hc[0].n_pad = (int)n_query, hc[0].factor = 1.f;
  hc[1].nrm = d_n, hc[1].nrm_x = d_n, hc[1].n = (int)n_ref;
  PairDev hp;
  std::memset(&hp, 0, sizeof(hp));
  hp.src_cloud = 0, hp.tgt_cloud = 1, hp.scan1_cloud = 0, hp.scan2_cloud = 1;
  hp.keys = d_keys, hp.n_src = (int)n;
  PairState hs;
  std::memset(&hs, 0, sizeof(hs));
  std::memcpy(hs.Tfinal, T, sizeof(hs.Tfinal));
  const int chunks = std::max(1, std::min(RED_MAX_CHUNKS, ((int)n + 4095) / 4096));
  int4 hb[RED_MAX_CHUNKS];
  for (int c = 0; c < chunks; ++c) hb[c] = make_int4(0, c, chunks, 0);
  const int zero = 0;
  HIP_TRY(ctx, hipMemcpy(d_keys, hk.data(), sizeof(nnkey_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_clouds, hc, sizeof(hc), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_pair, &hp, sizeof(hp), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_state, &hs, sizeof(hs), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_bmap, hb, sizeof(int4) * chunks, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_ids, &zero, sizeof(int), hipMemcpyHostToDevice));
  if (normalize_query) HIP_TRY(ctx, launch_normalize_constants(d_clouds, d_ids, 1, st));
  HIP_TRY(ctx, launch_post_reduce(d_clouds, d_pair, d_state, d_bmap, chunks, d_part, false, st));
  std::vector<double> hpart((size_t)chunks * NPOST);
  HIP_TRY(ctx, hipMemcpyAsync(hpart.data(), d_part, sizeof(double) * hpart.size(),
                              hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  double post[NPOST] = {0};
  for (int c = 0; c < chunks; ++c)
    for (int k = 0; k < NPOST; ++k) post[k] += hpart[(size_t)c * NPOST + k];
  int k = 3;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) Ap[a * 6 + b] = Ap[b * 6 + a] = post[k++];
  return LGICP_OK;
}

int lgicp_fpfh(lgicp_ctx* ctx, const float* surface_xyz, const float* surface_normals, size_t n,
               const float* keypoints_xyz, size_t n_key, double radius, float* out) {
  if (!ctx || (n && (!surface_xyz || !surface_normals)) || (n_key && (!keypoints_xyz || !out)))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  if (n > (size_t)1 << 28 || n_key > (size_t)1 << 24) return fail(ctx, LGICP_E_INVALID, "too large");
  if (n_key == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  float4* d_s = upload_points(d, surface_xyz, n, (int)std::max<size_t>(n, 1), 1.f);
  float4* d_n = upload_points(d, surface_normals, n, (int)std::max<size_t>(n, 1), 0.f);
  float4* d_k = upload_points(d, keypoints_xyz, n_key, (int)n_key, 1.f);
  unsigned char* d_need = d.alloc<unsigned char>(n);
  float* d_spfh = d.alloc<float>(n * 33);
  float* d_out = d.alloc<float>(n_key * 33);
  float4* d_boxes = d.alloc<float4>(2 * ((n + 63) / 64) + 2);
  if (!d_s || !d_n || !d_k || !d_need || !d_spfh || !d_out || !d_boxes)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, launch_chunk_boxes(d_s, (int)n, d_boxes, ctx->stream));
  HIP_TRY(ctx, launch_fpfh(d_s, d_n, (int)n, d_k, (int)n_key, (float)(radius * radius), d_need,
                           d_spfh, d_out, d_boxes, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * n_key * 33, hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LGICP_OK;
}

int lgicp_harris_keypoints(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                           int32_t* idx_out, size_t capacity, size_t* count) {
  if (!ctx || !count || (n && !xyz) || (capacity && !idx_out))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  *count = 0;
  if (n == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  float4* d_x = upload_points(d, xyz, n, (int)n, 1.f);
  float4* d_n = normals ? upload_points(d, normals, n, (int)n, 0.f) : d.alloc<float4>(n);
  if (!d_x || !d_n) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  if (!normals) {
    // the overload without normals (PointCloudUtils.cc:61-76): HarrisKeypoint3D estimates them
    // itself over a radius search with the Harris radius
    const float r2 = (float)(ctx->cfg.harris_radius * ctx->cfg.harris_radius);
    HIP_TRY(ctx, launch_normals_radius(d_x, (int)n, r2, d_n, ctx->stream));
  }
  CloudFeatures cf;
  const int rc = compute_cloud_features(ctx, d_x, d_n, (int)n, &cf);
  if (rc != LGICP_OK) return rc;
  *count = cf.idx.size();
  for (size_t i = 0; i < cf.idx.size() && i < capacity; ++i) idx_out[i] = cf.idx[i];
  return LGICP_OK;
}

int lgicp_sac_initial_alignment(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals,
                                size_t n_src, const float* tgt_xyz, const float* tgt_normals,
                                size_t n_tgt, const float guess[16], float T_out[16],
                                double* fitness_score) {
  if (!ctx || !guess || !T_out || !fitness_score || !src_xyz || !src_normals || !tgt_xyz ||
      !tgt_normals)
    return fail(ctx, LGICP_E_INVALID, "null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  float4* d_sx = upload_points(d, src_xyz, n_src, (int)std::max<size_t>(n_src, 1), 1.f);
  float4* d_sn = upload_points(d, src_normals, n_src, (int)std::max<size_t>(n_src, 1), 0.f);
  float4* d_tx = upload_points(d, tgt_xyz, n_tgt, (int)std::max<size_t>(n_tgt, 1), 1.f);
  float4* d_tn = upload_points(d, tgt_normals, n_tgt, (int)std::max<size_t>(n_tgt, 1), 0.f);
  if (!d_sx || !d_sn || !d_tx || !d_tn) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  CloudFeatures fs, ft;
  int rc = compute_cloud_features(ctx, d_sx, d_sn, (int)n_src, &fs);
  if (rc == LGICP_OK) rc = compute_cloud_features(ctx, d_tx, d_tn, (int)n_tgt, &ft);
  if (rc != LGICP_OK) return rc;
  return sac_align_device(ctx, fs, ft, guess, 0x5AC1A000u, T_out, fitness_score);
}

int lgicp_compute_normals(lgicp_ctx* ctx, const float* xyz, size_t n, int k, float* normals,
                          float* curvature) {
  if (!ctx || (n && (!xyz || !normals))) return fail(ctx, LGICP_E_INVALID, "null argument");
  if (k < 1 || k > 32) return fail(ctx, LGICP_E_INVALID, "k must be in 1..32");
  if (n > (size_t)1 << 28) return fail(ctx, LGICP_E_INVALID, "cloud too large");
  if (n == 0) return LGICP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  const int n_pad = std::max(NN_TILE, round_up((int)n, NN_TILE));
  float4* d_x = upload_points(d, xyz, n, n_pad, 1.f);
  float4* d_o = d.alloc<float4>(n);
  if (!d_x || !d_o) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, launch_normals_knn(d_x, (int)n, n_pad, (int)std::min<size_t>((size_t)k, n), d_o,
                                  ctx->stream));
  std::vector<float4> h(n);
  HIP_TRY(ctx, hipMemcpyAsync(h.data(), d_o, sizeof(float4) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; ++i) {
    normals[3 * i] = h[i].x, normals[3 * i + 1] = h[i].y, normals[3 * i + 2] = h[i].z;
    if (curvature) curvature[i] = h[i].w;
  }
  return LGICP_OK;
}

int lgicp_icp_observability(lgicp_ctx* ctx, const float* xyz, const float* normals, size_t n,
                            double eigenvalues[3]) {
  if (!ctx || !eigenvalues || (n && (!xyz || !normals)))
    return fail(ctx, LGICP_E_INVALID, "null argument");
  eigenvalues[0] = eigenvalues[1] = eigenvalues[2] = 0.0;
  if (n == 0) return LGICP_OK;
  // ExtractNormals (PointCloudUtils.cc:17-35): stored normals unless the first one is all zero
  std::vector<float> computed;
  const float* nrm = normals;
  if (normals[0] == 0.f && normals[1] == 0.f && normals[2] == 0.f) {
    computed.resize(3 * n);
    const int rc = lgicp_compute_normals(ctx, xyz, n, 10, computed.data(), nullptr);
    if (rc != LGICP_OK) return rc;
    nrm = computed.data();
  }
  // every point corresponds with itself, identity transform (PointCloudUtils.cc:176-186)
  std::vector<int32_t> corr(n);
  for (size_t i = 0; i < n; ++i) corr[i] = (int32_t)i;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double Ap[36];
  const int rc = lgicp_compute_ap(ctx, xyz, n, nrm, n, corr.data(), n, I, 1, Ap);
  if (rc != LGICP_OK) return rc;
  double B[9], ev[3], V[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) B[a * 3 + b] = Ap[(3 + a) * 6 + 3 + b];
  lgicp::sac::eig3(B, ev, V);  // ascending, as Eigen's SelfAdjointEigenSolver reports them
  for (int a = 0; a < 3; ++a) eigenvalues[a] = ev[a];
  return LGICP_OK;
}

int lgicp_teaser_initial_alignment(lgicp_ctx* ctx, const float* src_xyz, const float* src_normals,
                                   size_t n_src, const float* tgt_xyz, const float* tgt_normals,
                                   size_t n_tgt, const float guess[16], float T_out[16],
                                   int32_t* used_teaser, int32_t* n_matches, int32_t* n_clique) {
  if (!ctx || !guess || !T_out || !used_teaser || !src_xyz || !src_normals || !tgt_xyz ||
      !tgt_normals)
    return fail(ctx, LGICP_E_INVALID, "null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  Dev d;
  float4* d_sx = upload_points(d, src_xyz, n_src, (int)std::max<size_t>(n_src, 1), 1.f);
  float4* d_sn = upload_points(d, src_normals, n_src, (int)std::max<size_t>(n_src, 1), 0.f);
  float4* d_tx = upload_points(d, tgt_xyz, n_tgt, (int)std::max<size_t>(n_tgt, 1), 1.f);
  float4* d_tn = upload_points(d, tgt_normals, n_tgt, (int)std::max<size_t>(n_tgt, 1), 0.f);
  if (!d_sx || !d_sn || !d_tx || !d_tn) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  CloudFeatures fs, ft;
  int rc = compute_cloud_features(ctx, d_sx, d_sn, (int)n_src, &fs);
  if (rc == LGICP_OK) rc = compute_cloud_features(ctx, d_tx, d_tn, (int)n_tgt, &ft);
  if (rc != LGICP_OK) return rc;
  int used = 0, nm = 0, nc = 0;
  rc = teaser_align_device(ctx, fs, ft, guess, T_out, &used, &nm, &nc);
  *used_teaser = used;
  if (n_matches) *n_matches = nm;
  if (n_clique) *n_clique = nc;
  return rc;
}

int lgicp_teaser_clique(lgicp_ctx* ctx, const uint64_t* adjacency, size_t n, int32_t* clique_out,
                        int32_t* n_clique) {
  if (!ctx || !adjacency || !clique_out || !n_clique) return fail(ctx, LGICP_E_INVALID, "null argument");
  *n_clique = 0;
  if (n == 0) return LGICP_OK;
  if (n > (size_t)clique_device_max_vertices())
    return fail(ctx, LGICP_E_UNSUPPORTED, "graph of %zu vertices: the device search takes at most %d", n,
                clique_device_max_vertices());
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
  const size_t words = (n + 63) / 64;
  unsigned long long* d_bits = nullptr;
  if (hipMalloc((void**)&d_bits, sizeof(uint64_t) * n * words) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  std::vector<int> clique;
  bool ok = hipMemcpy(d_bits, adjacency, sizeof(uint64_t) * n * words, hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && greedy_clique_device(&ctx->clique_bufs[0], ctx->cfg.device, ctx->stream, d_bits, (int)n, (int)words,
                                  &clique);
  (void)hipFree(d_bits);
  if (!ok) return fail(ctx, LGICP_E_HIP, "clique search on the device failed (%s)", hipGetErrorString(hipGetLastError()));
  for (size_t i = 0; i < clique.size(); ++i) clique_out[i] = clique[i];
  *n_clique = (int32_t)clique.size();
  return LGICP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
namespace lgicp_detail {

int compute_cloud_features(lgicp_ctx* ctx, const float4* d_xyz, const float4* d_nrm, int n,
                           CloudFeatures* out, int lane) {
  out->idx.clear();
  out->xyz.clear();
  out->feat.clear();
  out->dev.reset();
  out->d_feat = out->d_kxyz = nullptr;
  if (n <= 0) return LGICP_OK;
  const lgicp_config& cfg = ctx->cfg;
  if (cfg.harris_response != 1 || cfg.harris_refine || !cfg.harris_suppression)
    return fail(ctx, LGICP_E_UNSUPPORTED,
                "only the HARRIS response with non-maxima suppression and no refinement is built");
  if (lane < 0 || lane >= lgicp_ctx::FEATURE_SETS * FEATURE_LANES || lane >= ctx->n_slots) lane = 0;
  hipStream_t st = ctx->slot[lane].stream;
  FeatureScratch& d = ctx->feature_scratch.get()[lane];
  if (d.resp.reserve((size_t)n) != hipSuccess || d.key.reserve((size_t)n) != hipSuccess ||
      d.boxes.reserve(2 * (((size_t)n + 63) / 64)) != hipSuccess)
    return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  float* d_resp = d.resp.p;
  unsigned char* d_key = d.key.p;
  // boxes of the cloud's 64-point chunks: both radius sweeps below skip what lies out of reach
  // (reserved[5] bit 2: sweep everything, for cross-checks -- the results are the same bits)
  const float4* d_boxes = (cfg.reserved[5] & 4) ? nullptr : d.boxes.p;
  if (d_boxes) HIP_TRY(ctx, launch_chunk_boxes(d_xyz, n, d.boxes.p, st));
  const float hr2 = (float)(cfg.harris_radius * cfg.harris_radius);
  HIP_TRY(ctx, launch_harris(d_xyz, d_nrm, n, hr2, (float)cfg.harris_threshold, d_resp, d_key, d_boxes, st));
  std::vector<unsigned char> flags((size_t)n);
  std::vector<float4> pts((size_t)n);
  HIP_TRY(ctx, hipMemcpyAsync(flags.data(), d_key, (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i)
    if (flags[i]) out->idx.push_back(i);
  const size_t k = out->idx.size();
  if (k == 0) return LGICP_OK;
  // keypoint coordinates: gather on the host from one download of the cloud
  HIP_TRY(ctx, hipMemcpyAsync(pts.data(), d_xyz, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  std::vector<float4> kp(k);
  out->xyz.resize(3 * k);
  for (size_t q = 0; q < k; ++q) {
    kp[q] = pts[out->idx[q]];
    out->xyz[3 * q] = kp[q].x, out->xyz[3 * q + 1] = kp[q].y, out->xyz[3 * q + 2] = kp[q].z;
  }
  const bool scratch_ok = d.kp.reserve(k) == hipSuccess && d.need.reserve((size_t)n) == hipSuccess &&
                          d.spfh.reserve((size_t)n * 33) == hipSuccess;
  float4* d_kp = scratch_ok ? d.kp.p : nullptr;
  unsigned char* d_need = d.need.p;
  float* d_spfh = d.spfh.p;
  // descriptors and keypoint coordinates stay on the device for the matcher
  auto block = std::make_shared<DevBlock>();
  if (hipMalloc((void**)&block->p, sizeof(float) * 36 * k) != hipSuccess) block->p = nullptr;
  float* d_out = reinterpret_cast<float*>(block->p);
  if (!scratch_ok || !d_out) return fail(ctx, LGICP_E_NOMEM, "device allocation failed");
  HIP_TRY(ctx, hipMemcpyAsync(d_out + 33 * k, out->xyz.data(), sizeof(float) * 3 * k, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_kp, kp.data(), sizeof(float4) * k, hipMemcpyHostToDevice, st));
  const float fr2 = (float)(cfg.sac_features_radius * cfg.sac_features_radius);
  HIP_TRY(ctx, launch_fpfh(d_xyz, d_nrm, n, d_kp, (int)k, fr2, d_need, d_spfh, d_out, d_boxes, st));
  out->feat.resize(33 * k);
  HIP_TRY(ctx, hipMemcpyAsync(out->feat.data(), d_out, sizeof(float) * 33 * k, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  out->dev = block;
  out->d_feat = d_out;
  out->d_kxyz = d_out + 33 * k;
  return LGICP_OK;
}

}  // namespace lgicp_detail
