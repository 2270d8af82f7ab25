// GroupNorm forward / backward on NHWC 16-bit activations (gfx950).
//
// Replaces nn.GroupNorm(get_group_gn(planes), planes) as built by models/utils/layers.py:50-54,138-154 (32 groups)
// and applied after the convs of models/backbone/resnet.py (use_gn=True: :42-59, :97-119, :254-257) and of ConvModule
// (layers.py:122-135, necks with normalize=GN), together with the residual add and ReLU that follow it.
// Unlike eval-mode BatchNorm the statistics depend on the sample, so GN cannot be folded into the conv epilogue:
//   forward   z = conv(x) (raw, 16-bit)  ->  [1] per-channel partial (sum, sum of squares) over pixel chunks
//             -> [2] per (sample, group) mean / rstd, expanded to per-(sample, channel) affine (a, b)
//             -> [3] y = relu?(z*a + b (+ addend))      (relu: 0 none, 1 ReLU, 2 ReLU6)
//   backward  g = dL/dy (ReLU-masked) -> [1'] per-channel partial (sum g, sum g*xhat) -> [2'] dgamma, dbeta and the
//             per-(sample, channel) coefficients of dz = g*A + z*B + C  ->  [3'] dz, which then feeds the ordinary
//             conv dgrad / wgrad kernels.
// All three passes are HBM-bound streams, 16 bytes (8 channels) per lane; reductions are fixed-order (deterministic).
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct GnGeom {
  int N, HW, C, G, cpg;      // cpg = C / G channels per group
  int C8;                    // C / 8 lanes per pixel
  int ppp;                   // pixels per pass of one 256-thread block
  int chunks, chunk_px;      // pixel chunks per sample, pixels per chunk
};

// [1] / [1']: partial per-channel sums over one pixel chunk of one sample.
//   MODE 0: (sum z, sum z^2)            MODE 1: (sum g, sum g * xhat), xhat = z*rstd - mu*rstd from `stats`
// part layout: [n][chunk][2][C]
template <int MODE, bool F16>
__global__ __launch_bounds__(kThreads) void gn_partial_kernel(const bf16_t* __restrict__ z,
                                                               const bf16_t* __restrict__ g,
                                                               const float* __restrict__ stats, GnGeom ge,
                                                               float* __restrict__ part) {
  __shared__ float red[kThreads][17];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int cl = tid % ge.C8, pl = tid / ge.C8;   // channel lane (8 channels), pixel lane
  const int p0 = chunk * ge.chunk_px;
  const int p1 = min(ge.HW, p0 + ge.chunk_px);
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
  float mu_r[8], rs[8];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mu = stats[((int64_t)n * ge.C + cl * 8 + e) * 2], r = stats[((int64_t)n * ge.C + cl * 8 + e) * 2 + 1];
      rs[e] = r;
      mu_r[e] = mu * r;
    }
  }
  const int64_t base = (int64_t)n * ge.HW * ge.C + cl * 8;
  for (int p = p0 + pl; p < p1; p += ge.ppp) {
    const bf16x8_t zv = *(const bf16x8_t*)(z + base + (int64_t)p * ge.C);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = elem_to_f32<F16>(zv[e]);
        a0[e] += v;
        a1[e] += v * v;
      }
    } else {
      const bf16x8_t gv = *(const bf16x8_t*)(g + base + (int64_t)p * ge.C);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = elem_to_f32<F16>(gv[e]);
        const float xh = elem_to_f32<F16>(zv[e]) * rs[e] - mu_r[e];
        a0[e] += gg;
        a1[e] += gg * xh;
      }
    }
  }
  // combine the pixel lanes in lane order (fixed order -> deterministic)
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid][e] = a0[e]; red[tid][8 + e] = a1[e]; }
  __syncthreads();
  if (pl == 0) {
    for (int j = 1; j < ge.ppp; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[tid][e] += red[j * ge.C8 + cl][e];
    }
    float* o = part + (((int64_t)n * ge.chunks + chunk) * 2) * ge.C + cl * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = red[tid][e]; o[ge.C + e] = red[tid][8 + e]; }
  }
}

// Sum of the per-chunk partials of channel c of sample n, both planes, by the KL chunk-lanes of a block (thread
// (cl, kl) takes chunks kl, kl+KL, ...; lanes combined through LDS in lane order -> deterministic).  CB channels per
// block, CB * KL = kThreads.  Returns the totals to every thread with kl == 0.
__device__ __forceinline__ void chunk_totals(const float* __restrict__ part, const GnGeom& ge, int n, int c, int cl,
                                             int kl, int CB, int KL, bool live, double* red /*[2][kThreads]*/,
                                             double& s, double& ss) {
  double a = 0.0, b = 0.0;
  if (live) {
    for (int k = kl; k < ge.chunks; k += KL) {
      const float* q = part + (((int64_t)n * ge.chunks + k) * 2) * ge.C + c;
      a += (double)q[0];
      b += (double)q[ge.C];
    }
  }
  red[kl * CB + cl] = a;
  red[kThreads + kl * CB + cl] = b;
  __syncthreads();
  s = 0.0; ss = 0.0;
  if (kl == 0) {
    for (int j = 0; j < KL; ++j) { s += red[j * CB + cl]; ss += red[kThreads + j * CB + cl]; }
  }
  __syncthreads();
}

// [2]: block (sample n, channel block of CB = max(32, cpg) channels — whole groups); chunk sums -> group mean / rstd
// (double), then per-channel stats[n][c] = (mu, rstd) and coef[n][c] = (a, b) with y = z*a + b.
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const float* __restrict__ part, GnGeom ge, int CB,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ stats, float* __restrict__ coef) {
  __shared__ double red[2 * kThreads];
  __shared__ double tot[2 * kThreads];
  const int n = blockIdx.y;
  const int KL = kThreads / CB;
  const int cl = threadIdx.x % CB, kl = threadIdx.x / CB;
  const int c = blockIdx.x * CB + cl;
  double s, ss;
  chunk_totals(part, ge, n, c, cl, kl, CB, KL, c < ge.C, red, s, ss);
  if (kl == 0) { tot[cl] = s; tot[kThreads + cl] = ss; }
  __syncthreads();
  if (kl == 0 && c < ge.C) {
    const int g0 = (cl / ge.cpg) * ge.cpg;
    double gs = 0.0, gss = 0.0;
    for (int j = 0; j < ge.cpg; ++j) { gs += tot[g0 + j]; gss += tot[kThreads + g0 + j]; }
    const double cnt = (double)ge.HW * ge.cpg;
    const double mu = gs / cnt;
    double var = gss / cnt - mu * mu;       // biased variance, like nn.GroupNorm
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    stats[((int64_t)n * ge.C + c) * 2] = muf;
    stats[((int64_t)n * ge.C + c) * 2 + 1] = rstd;
    const float a = rstd * gamma[c];
    coef[((int64_t)n * ge.C + c) * 2] = a;
    coef[((int64_t)n * ge.C + c) * 2 + 1] = beta[c] - muf * a;
  }
}

// [3]: y = relu?(z*a + b (+ addend));  up_w > 0: the addend is the coarser FPN level (H/2 x W/2), read with
// nearest-neighbour 2x upsampling (fpn.py:98-100), W = 2*up_w
template <bool F16>
__global__ void gn_apply_kernel(const bf16_t* __restrict__ z, const float* __restrict__ coef,
                                const bf16_t* __restrict__ addend, int relu, int up_w, GnGeom ge,
                                bf16_t* __restrict__ y) {
  const int64_t per_n = (int64_t)ge.HW * ge.C8, total = per_n * ge.N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per_n);
    const int cl = (int)(i % ge.C8);
    const bf16x8_t zv = *(const bf16x8_t*)(z + i * 8);
    const float* cf = coef + ((int64_t)n * ge.C + cl * 8) * 2;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = elem_to_f32<F16>(zv[e]) * cf[2 * e] + cf[2 * e + 1];
    if (addend) {
      int64_t ai = i;
      if (up_w > 0) {
        const int W = 2 * up_w;
        const int64_t pix = (i - (int64_t)n * per_n) / ge.C8;
        const int h = (int)(pix / W), w = (int)(pix - (int64_t)h * W);
        ai = (((int64_t)n * (ge.HW / 4)) + (int64_t)(h >> 1) * up_w + (w >> 1)) * ge.C8 + cl;
      }
      const bf16x8_t av = *(const bf16x8_t*)(addend + ai * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += elem_to_f32<F16>(av[e]);
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = relu ? fmaxf(v[e], 0.f) : v[e];
      if (relu == 2) t = relu6_top<F16>(t);       // nn.ReLU6
      o[e] = f32_to_elem<F16>(t);
    }
    *(bf16x8_t*)(y + i * 8) = o;
  }
}

// [2']: block = channel block of CB = max(32, cpg) channels (whole groups), all samples in turn: chunk sums ->
// (s1, s2), group means m1 = sum_c gamma*s1 / cnt, m2 = sum_c gamma*s2 / cnt, coefficients of dz = g*A + z*B + Cc;
// across samples: dgamma = sum_n s2, dbeta = sum_n s1.
__global__ __launch_bounds__(kThreads) void gn_bwd_coef_kernel(const float* __restrict__ part, GnGeom ge, int CB,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ stats,
                                                               float* __restrict__ coef3, float* dgamma, float* dbeta,
                                                               float acc) {
  __shared__ double red[2 * kThreads];
  __shared__ float t1[kThreads], t2[kThreads];
  const int KL = kThreads / CB;
  const int cl = threadIdx.x % CB, kl = threadIdx.x / CB;
  const int c = blockIdx.x * CB + cl;
  const bool live = c < ge.C;
  const float gm = live ? gamma[c] : 0.f;
  float dg = 0.f, db = 0.f;
  const float cnt = (float)ge.HW * (float)ge.cpg;
  for (int n = 0; n < ge.N; ++n) {
    double d1, d2;
    chunk_totals(part, ge, n, c, cl, kl, CB, KL, live, red, d1, d2);
    const float s1 = (float)d1, s2 = (float)d2;
    if (kl == 0) {
      dg += s2;
      db += s1;
      t1[cl] = gm * s1;
      t2[cl] = gm * s2;
    }
    __syncthreads();
    if (kl == 0 && live) {
      const int l0 = (cl / ge.cpg) * ge.cpg;
      float m1 = 0.f, m2 = 0.f;
      for (int j = 0; j < ge.cpg; ++j) { m1 += t1[l0 + j]; m2 += t2[l0 + j]; }
      m1 /= cnt;
      m2 /= cnt;
      const float mu = stats[((int64_t)n * ge.C + c) * 2], r = stats[((int64_t)n * ge.C + c) * 2 + 1];
      float* o = coef3 + ((int64_t)n * ge.C + c) * 3;
      o[0] = r * gm;                       // A
      o[1] = -r * r * m2;                  // B
      o[2] = -r * m1 + mu * r * r * m2;    // C
    }
    __syncthreads();
  }
  if (kl == 0 && live) {
    dgamma[c] = (acc != 0.f) ? acc * dgamma[c] + dg : dg;
    dbeta[c] = (acc != 0.f) ? acc * dbeta[c] + db : db;
  }
}

// [3']: dz = g*A + z*B + C
template <bool F16>
__global__ void gn_bwd_apply_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ z,
                                    const float* __restrict__ coef3, GnGeom ge, bf16_t* __restrict__ dz) {
  const int64_t per_n = (int64_t)ge.HW * ge.C8, total = per_n * ge.N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per_n);
    const int cl = (int)(i % ge.C8);
    const bf16x8_t gv = *(const bf16x8_t*)(g + i * 8);
    const bf16x8_t zv = *(const bf16x8_t*)(z + i * 8);
    const float* cf = coef3 + ((int64_t)n * ge.C + cl * 8) * 3;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = f32_to_elem<F16>(elem_to_f32<F16>(gv[e]) * cf[3 * e] + elem_to_f32<F16>(zv[e]) * cf[3 * e + 1] +
                              cf[3 * e + 2]);
    *(bf16x8_t*)(dz + i * 8) = o;
  }
}

// ---- training-mode BatchNorm2d (batch statistics): the same three passes with the statistics taken per channel over
// the whole batch (N, H, W) instead of per (sample, group).  Only the two small middle kernels differ; stats / coef
// are still written per (sample, channel) — identical for every sample — so the element-wise passes are shared. ----
// totals over ALL samples and chunks of channel c, by the KL lanes of a block of CB = 32 channels
__device__ __forceinline__ void batch_totals(const float* __restrict__ part, const GnGeom& ge, int c, int cl, int kl,
                                             int CB, int KL, bool live, double* red, double& s, double& ss) {
  double a = 0.0, b = 0.0;
  if (live) {
    for (int k = kl; k < ge.N * ge.chunks; k += KL) {   // [n][chunk] is one flat index of the partials
      const float* q = part + ((int64_t)k * 2) * ge.C + c;
      a += (double)q[0];
      b += (double)q[ge.C];
    }
  }
  red[kl * CB + cl] = a;
  red[kThreads + kl * CB + cl] = b;
  __syncthreads();
  s = 0.0; ss = 0.0;
  if (kl == 0) {
    for (int j = 0; j < KL; ++j) { s += red[j * CB + cl]; ss += red[kThreads + j * CB + cl]; }
  }
}

__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const float* __restrict__ part, GnGeom ge,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float momentum,
                                                            float* running_mean, float* running_var,
                                                            float* __restrict__ stats, float* __restrict__ coef) {
  __shared__ double red[2 * kThreads];
  constexpr int CB = 32, KL = kThreads / CB;
  const int cl = threadIdx.x % CB, kl = threadIdx.x / CB;
  const int c = blockIdx.x * CB + cl;
  double s, ss;
  batch_totals(part, ge, c, cl, kl, CB, KL, c < ge.C, red, s, ss);
  if (kl != 0 || c >= ge.C) return;
  const double cnt = (double)ge.N * ge.HW;
  const double mu = s / cnt;
  double var = ss / cnt - mu * mu;             // biased variance normalises (nn.BatchNorm2d, training)
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  const float a = rstd * gamma[c], b = beta[c] - muf * a;
  for (int n = 0; n < ge.N; ++n) {
    stats[((int64_t)n * ge.C + c) * 2] = muf;
    stats[((int64_t)n * ge.C + c) * 2 + 1] = rstd;
    coef[((int64_t)n * ge.C + c) * 2] = a;
    coef[((int64_t)n * ge.C + c) * 2 + 1] = b;
  }
  if (running_mean && running_var) {           // running statistics: unbiased variance, momentum update
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * muf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_coef_kernel(const float* __restrict__ part, GnGeom ge,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ stats,
                                                               float* __restrict__ coef3, float* dgamma, float* dbeta,
                                                               float acc) {
  __shared__ double red[2 * kThreads];
  constexpr int CB = 32, KL = kThreads / CB;
  const int cl = threadIdx.x % CB, kl = threadIdx.x / CB;
  const int c = blockIdx.x * CB + cl;
  double d1, d2;
  batch_totals(part, ge, c, cl, kl, CB, KL, c < ge.C, red, d1, d2);
  if (kl != 0 || c >= ge.C) return;
  const float s1 = (float)d1, s2 = (float)d2;
  const float gm = gamma[c];
  const float cnt = (float)ge.N * (float)ge.HW;
  const float m1 = gm * s1 / cnt, m2 = gm * s2 / cnt;
  const float mu = stats[(int64_t)c * 2], r = stats[(int64_t)c * 2 + 1];
  for (int n = 0; n < ge.N; ++n) {
    float* o = coef3 + ((int64_t)n * ge.C + c) * 3;
    o[0] = r * gm;
    o[1] = -r * r * m2;
    o[2] = -r * m1 + mu * r * r * m2;
  }
  dgamma[c] = (acc != 0.f) ? acc * dgamma[c] + s2 : s2;
  dbeta[c] = (acc != 0.f) ? acc * dbeta[c] + s1 : s1;
}

int make_geom(GnGeom& ge, int N, int H, int W, int C, int G) {
  TDN_CHECK(N > 0 && H > 0 && W > 0, "GroupNorm: bad shape N=%d H=%d W=%d", N, H, W);
  TDN_CHECK(C >= 64 && C <= 2048 && (C & (C - 1)) == 0, "GroupNorm: C=%d must be a power of two in 64..2048", C);
  TDN_CHECK(G > 0 && C % G == 0, "GroupNorm: %d groups do not divide %d channels", G, C);
  ge.N = N; ge.HW = H * W; ge.C = C; ge.G = G; ge.cpg = C / G;
  TDN_CHECK((ge.cpg & (ge.cpg - 1)) == 0 && ge.cpg <= 256, "GroupNorm: %d channels per group not supported", ge.cpg);
  ge.C8 = C / 8;
  ge.ppp = kThreads / ge.C8;
  int chunks = ceil_div(1024, N);                       // ~1024 blocks in flight
  const int max_chunks = ceil_div(ge.HW, ge.ppp * 4);   // at least 4 passes per block
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  ge.chunk_px = ceil_div(ge.HW, chunks);
  ge.chunks = ceil_div(ge.HW, ge.chunk_px);
  return 0;
}

// workspace: part [N][chunks][2][C] floats | coef [N][C][3] floats
int64_t ws_floats(const GnGeom& ge) { return (int64_t)ge.N * ge.chunks * 2 * ge.C + (int64_t)ge.N * ge.C * 3; }

}  // namespace

extern "C" int64_t tdn_gn_workspace(int N, int H, int W, int C, int G) {
  GnGeom ge;
  if (make_geom(ge, N, H, W, C, G)) return -1;
  return ws_floats(ge) * 4 + 256;
}

extern "C" int tdn_gn_fwd(const void* z, const float* gamma, const float* beta, int N, int H, int W, int C, int G,
                          float eps, const void* addend, int addend_mode, int relu, void* y, float* stats,
                          void* workspace,
                          int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(z && gamma && beta && y && stats && workspace, "tdn_gn_fwd: NULL pointer");
  TDN_CHECK(!addend || addend_mode == TDN_ADD_SAME || addend_mode == TDN_ADD_UP2X,
            "tdn_gn_fwd: addend_mode %d (TDN_ADD_SAME or TDN_ADD_UP2X)", addend_mode);
  TDN_CHECK(!(addend && addend_mode == TDN_ADD_UP2X) || (H % 2 == 0 && W % 2 == 0),
            "tdn_gn_fwd: UP2X addend needs even H, W (got %dx%d)", H, W);
  const int up_w = (addend && addend_mode == TDN_ADD_UP2X) ? W / 2 : 0;
  GnGeom ge;
  if (make_geom(ge, N, H, W, C, G)) return -1;
  TDN_CHECK(workspace_bytes >= ws_floats(ge) * 4 && ((uintptr_t)workspace & 15) == 0,
            "tdn_gn_fwd: workspace too small or misaligned");
  float* part = (float*)workspace;
  float* coef = part + (int64_t)ge.N * ge.chunks * 2 * ge.C;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gp(ge.chunks, N);
  if (dtype == TDN_F16)
    TDN_LAUNCH((gn_partial_kernel<0, true>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, nullptr, nullptr, ge, part);
  else
    TDN_LAUNCH((gn_partial_kernel<0, false>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, nullptr, nullptr, ge, part);
  const int CB = ge.cpg > 32 ? ge.cpg : 32;
  TDN_LAUNCH(gn_stats_kernel, dim3(ceil_div(C, CB), N), dim3(kThreads), 0, st, part, ge, CB, gamma, beta, eps,
                     stats, coef);
  const int64_t total = (int64_t)N * ge.HW * ge.C8;
  int grid = (int)((total + kThreads - 1) / kThreads);
  if (grid > 8192) grid = 8192;
  TDN_LAUNCH_T(gn_apply_kernel, dtype, dim3(grid), dim3(kThreads), st, (const bf16_t*)z, coef, (const bf16_t*)addend,
               relu, up_w, ge, (bf16_t*)y);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_gn_bwd(const void* g, const void* z, const float* stats, const float* gamma, int N, int H, int W,
                          int C, int G, void* dz, float* dgamma, float* dbeta, float acc, void* workspace,
                          int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(g && z && stats && gamma && dz && dgamma && dbeta && workspace, "tdn_gn_bwd: NULL pointer");
  GnGeom ge;
  if (make_geom(ge, N, H, W, C, G)) return -1;
  TDN_CHECK(workspace_bytes >= ws_floats(ge) * 4 && ((uintptr_t)workspace & 15) == 0,
            "tdn_gn_bwd: workspace too small or misaligned");
  float* part = (float*)workspace;
  float* coef3 = part + (int64_t)ge.N * ge.chunks * 2 * ge.C;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gp(ge.chunks, N);
  if (dtype == TDN_F16)
    TDN_LAUNCH((gn_partial_kernel<1, true>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, (const bf16_t*)g, stats, ge, part);
  else
    TDN_LAUNCH((gn_partial_kernel<1, false>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, (const bf16_t*)g, stats, ge, part);
  const int CB = ge.cpg > 32 ? ge.cpg : 32;
  TDN_LAUNCH(gn_bwd_coef_kernel, dim3(ceil_div(C, CB)), dim3(kThreads), 0, st, part, ge, CB, gamma, stats,
                     coef3, dgamma, dbeta, acc);
  const int64_t total = (int64_t)N * ge.HW * ge.C8;
  int grid = (int)((total + kThreads - 1) / kThreads);
  if (grid > 8192) grid = 8192;
  TDN_LAUNCH_T(gn_bwd_apply_kernel, dtype, dim3(grid), dim3(kThreads), st, (const bf16_t*)g, (const bf16_t*)z, coef3,
               ge, (bf16_t*)dz);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Training-mode nn.BatchNorm2d (models/utils/layers.py:50-54 with ResNet(bn_eval=False), resnet.py:270-276): batch
// statistics over (N, H, W), running statistics updated in place (momentum; unbiased variance) when given.
// Workspace: tdn_gn_workspace(N, H, W, C, C).
extern "C" int tdn_bn_train_fwd(const void* z, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, int N, int H, int W, int C, float eps,
                                const void* addend, int addend_mode, int relu, void* y, float* stats, void* workspace,
                                int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(z && gamma && beta && y && stats && workspace, "tdn_bn_train_fwd: NULL pointer");
  TDN_CHECK((running_mean == nullptr) == (running_var == nullptr), "tdn_bn_train_fwd: give both running stats or none");
  TDN_CHECK(!addend || addend_mode == TDN_ADD_SAME || addend_mode == TDN_ADD_UP2X,
            "tdn_bn_train_fwd: addend_mode %d (TDN_ADD_SAME or TDN_ADD_UP2X)", addend_mode);
  TDN_CHECK(!(addend && addend_mode == TDN_ADD_UP2X) || (H % 2 == 0 && W % 2 == 0),
            "tdn_bn_train_fwd: UP2X addend needs even H, W (got %dx%d)", H, W);
  const int up_w = (addend && addend_mode == TDN_ADD_UP2X) ? W / 2 : 0;
  GnGeom ge;
  if (make_geom(ge, N, H, W, C, C)) return -1;
  TDN_CHECK(workspace_bytes >= ws_floats(ge) * 4 && ((uintptr_t)workspace & 15) == 0,
            "tdn_bn_train_fwd: workspace too small or misaligned");
  float* part = (float*)workspace;
  float* coef = part + (int64_t)ge.N * ge.chunks * 2 * ge.C;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gp(ge.chunks, N);
  if (dtype == TDN_F16)
    TDN_LAUNCH((gn_partial_kernel<0, true>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, nullptr, nullptr, ge, part);
  else
    TDN_LAUNCH((gn_partial_kernel<0, false>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, nullptr, nullptr, ge, part);
  TDN_LAUNCH(bn_stats_kernel, dim3(ceil_div(C, 32)), dim3(kThreads), 0, st, part, ge, gamma, beta, eps,
                     momentum, running_mean, running_var, stats, coef);
  const int64_t total = (int64_t)N * ge.HW * ge.C8;
  int grid = (int)((total + kThreads - 1) / kThreads);
  if (grid > 8192) grid = 8192;
  TDN_LAUNCH_T(gn_apply_kernel, dtype, dim3(grid), dim3(kThreads), st, (const bf16_t*)z, coef, (const bf16_t*)addend,
               relu, up_w, ge, (bf16_t*)y);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_bn_train_bwd(const void* g, const void* z, const float* stats, const float* gamma, int N, int H,
                                int W, int C, void* dz, float* dgamma, float* dbeta, float acc, void* workspace,
                                int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(g && z && stats && gamma && dz && dgamma && dbeta && workspace, "tdn_bn_train_bwd: NULL pointer");
  GnGeom ge;
  if (make_geom(ge, N, H, W, C, C)) return -1;
  TDN_CHECK(workspace_bytes >= ws_floats(ge) * 4 && ((uintptr_t)workspace & 15) == 0,
            "tdn_bn_train_bwd: workspace too small or misaligned");
  float* part = (float*)workspace;
  float* coef3 = part + (int64_t)ge.N * ge.chunks * 2 * ge.C;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gp(ge.chunks, N);
  if (dtype == TDN_F16)
    TDN_LAUNCH((gn_partial_kernel<1, true>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, (const bf16_t*)g, stats, ge, part);
  else
    TDN_LAUNCH((gn_partial_kernel<1, false>), gp, dim3(kThreads), 0, st, (const bf16_t*)z, (const bf16_t*)g, stats, ge, part);
  TDN_LAUNCH(bn_bwd_coef_kernel, dim3(ceil_div(C, 32)), dim3(kThreads), 0, st, part, ge, gamma, stats,
                     coef3, dgamma, dbeta, acc);
  const int64_t total = (int64_t)N * ge.HW * ge.C8;
  int grid = (int)((total + kThreads - 1) / kThreads);
  if (grid > 8192) grid = 8192;
  TDN_LAUNCH_T(gn_bwd_apply_kernel, dtype, dim3(grid), dim3(kThreads), st, (const bf16_t*)g, (const bf16_t*)z, coef3,
               ge, (bf16_t*)dz);
  TDN_LAUNCH_CHECK();
  return 0;
}
