// HBM-bound helper kernels of the ResNet/FPN path: weight packing, BN folding, image staging,
// max-pool (resnet.py:218,258), stride-2 subsample (fpn.py:116), ReLU-mask/add, layout converters.
// All are 16-byte-per-lane streaming kernels over NHWC bf16.
#include "common.h"
#include <limits.h>
#include <string.h>

// Flat index -> coordinates, innermost extent first: i = ((a * D1 + b) * D2 + c) * D3 + d.  32-bit arithmetic whenever
// the index fits: a 64-bit quotient costs ~100 instructions on this ISA (it is what paced the pairwise-IoU kernel
// before its 2-D decomposition), a 32-bit one ~25.
__device__ __forceinline__ void split_idx(int64_t i, int D3, int D2, int D1, int& d, int& c, int& b, int& a) {
  if (i < (1ll << 31)) {
    const unsigned u = (unsigned)i;
    const unsigned q = u / (unsigned)D3;
    d = (int)(u - q * (unsigned)D3);
    const unsigned q2 = q / (unsigned)D2;
    c = (int)(q - q2 * (unsigned)D2);
    const unsigned q3 = q2 / (unsigned)D1;
    b = (int)(q2 - q3 * (unsigned)D1);
    a = (int)q3;
  } else {
    d = (int)(i % D3);
    int64_t r = i / D3;
    c = (int)(r % D2);
    r /= D2;
    b = (int)(r % D1);
    a = (int)(r / D1);
  }
}
__device__ __forceinline__ void split_idx(int64_t i, int D3, int D2, int& d, int& c, int& b) {
  if (i < (1ll << 31)) {
    const unsigned u = (unsigned)i;
    const unsigned q = u / (unsigned)D3;
    d = (int)(u - q * (unsigned)D3);
    const unsigned q2 = q / (unsigned)D2;
    c = (int)(q - q2 * (unsigned)D2);
    b = (int)q2;
  } else {
    d = (int)(i % D3);
    const int64_t r = i / D3;
    c = (int)(r % D2);
    b = (int)(r / D2);
  }
}


static thread_local char g_err[512] = "";

void tdn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tdn_last_error(void) { return g_err; }
extern "C" int tdn_version(void) { return TDN_VERSION; }

static inline int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- BN fold ---------------------------------------------------------------------------------
// 1 / sqrt(var + eps) of an eval-mode BatchNorm2d.  One out-of-line copy: the per-layer fold, the grouped fold and the
// grouped pack (which recomputes the scale instead of waiting for the fold) must produce the same bits, whatever
// instruction selection each caller's context would have led to.
__device__ __noinline__ float bn_invstd(float var, float eps) { return 1.0f / sqrtf(var + eps); }

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               float eps, int C, float* scale, float* shift, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = bn_invstd(var[c], eps);
  const float s = gamma[c] * is;
  scale[c] = s;
  shift[c] = beta[c] - mean[c] * s;
  invstd[c] = is;
}

extern "C" int tdn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                           float eps, int C, float* scale, float* shift, float* invstd, void* stream) {
  TDN_CHECK(gamma && beta && mean && var && scale && shift && invstd, "tdn_bn_fold: NULL pointer");
  TDN_CHECK(C > 0, "tdn_bn_fold: C=%d", C);
  TDN_LAUNCH(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     mean, var, eps, C, scale, shift, invstd);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- weight packing --------------------------------------------------------------------------
template <bool F16>
__global__ void pack_weight_kernel(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int Cout,
                                   int Cin, int kh, int kw, const float* scale, bf16_t* w_fwd, bf16_t* w_dgrad) {
  const int64_t total = (int64_t)Cout * Cin * kh * kw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates the fwd layout [co][h][w][ci]
    const int ci = (int)(i % Cin);
    int64_t r = i / Cin;
    const int x = (int)(r % kw);
    r /= kw;
    const int y = (int)(r % kh);
    const int co = (int)(r / kh);
    const float v = w[co * s_o + ci * s_i + y * s_h + x * s_w];
    const bf16_t vb = f32_to_elem<F16>(v);
    w_fwd[i] = vb;
    if (w_dgrad) {
      const float sc = scale ? scale[co] : 1.f;
      // dgrad operand uses the bf16-rounded forward weight times the fp32 scale
      w_dgrad[(((int64_t)ci * kh + y) * kw + x) * Cout + co] =
          f32_to_elem<F16>(mul_f32_rounded(elem_to_f32<F16>(vb), sc));
    }
  }
}

extern "C" int tdn_pack_conv_weight(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int Cout,
                                    int Cin, int kh, int kw, const float* scale, void* w_fwd, void* w_dgrad,
                                    int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(w && w_fwd, "tdn_pack_conv_weight: NULL pointer");
  const int64_t total = (int64_t)Cout * Cin * kh * kw;
  TDN_CHECK(total > 0, "tdn_pack_conv_weight: empty weight");
  TDN_LAUNCH_T(pack_weight_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream, w, s_o, s_i,
                     s_h, s_w, Cout, Cin, kh, kw, scale, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Grouped conv weight [C][cpg][kh][kw] (resnext.py:26-28,82-83) -> block-diagonal operands [C][kh][kw][64]:
//   w_fwd[co][tap][j]   = w[co][j % cpg][tap]                 if input slot j (channel 64*(co/64) + j) is in co's group
//   w_dgrad[ci][tap][j] = scale[co'] * w[co'][ci % cpg][tap]  with co' = 64*(ci/64) + j, if co' is in ci's group
// zeros elsewhere.
template <bool F16>
__global__ void pack_gconv_kernel(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int C, int cpg,
                                  int kh, int kw, const float* scale, bf16_t* w_fwd, bf16_t* w_dgrad) {
  const int64_t total = (int64_t)C * kh * kw * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 63);
    int64_t r = i >> 6;
    const int x = (int)(r % kw);
    r /= kw;
    const int y = (int)(r % kh);
    const int c = (int)(r / kh);   // row channel: output channel for w_fwd, input channel for w_dgrad
    const bool same = (j / cpg) == ((c & 63) / cpg);
    float vf = 0.f, vd = 0.f;
    if (same) {
      const float wf = w[c * s_o + (j % cpg) * s_i + y * s_h + x * s_w];
      vf = wf;
      const int co = (c & ~63) + j;   // the output channel this dgrad entry multiplies
      const float wd = w[co * s_o + (c % cpg) * s_i + y * s_h + x * s_w];
      vd = mul_f32_rounded(elem_to_f32<F16>(f32_to_elem<F16>(wd)), scale ? scale[co] : 1.f);
    }
    w_fwd[i] = f32_to_elem<F16>(vf);
    if (w_dgrad) w_dgrad[i] = f32_to_elem<F16>(vd);
  }
}

extern "C" int tdn_pack_gconv_weight(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int C,
                                     int groups, int kh, int kw, const float* scale, void* w_fwd, void* w_dgrad,
                                     int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(w && w_fwd, "tdn_pack_gconv_weight: NULL pointer");
  TDN_CHECK(groups > 0 && C % groups == 0 && C % 64 == 0 && (C / groups) <= 64 && 64 % (C / groups) == 0,
            "grouped conv: need C %% 64 == 0 and channels per group dividing 64 (C=%d, groups=%d)", C, groups);
  const int64_t total = (int64_t)C * kh * kw * 64;
  TDN_LAUNCH_T(pack_gconv_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream, w, s_o, s_i, s_h,
               s_w, C, C / groups, kh, kw, scale, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- grouped operand preparation: BN fold + weight pack of many conv units in ONE launch -------------------------
// A training step re-derives the 16-bit operands of every conv from the fp32 parameters (the optimizer changed them):
// per unit that was one bn_fold and one pack launch — 113 launches (0.55 ms) per ResNet-50-FPN step.  Here the work
// list spans all units of a net (descriptors in the kernel-argument block, <= PREP_MAXI per launch):
//   * pack blocks: one 64 (co) x 64 (ci) tile of one tap — read fp32 (coalesced along whichever of ci / co / tap is
//     the unit-stride axis of the parameter's memory format), round to the element type, store w_fwd rows (ci fastest),
//     transpose through LDS, store the scaled w_dgrad rows (co fastest).  The BN scale is recomputed inline from
//     gamma / var with the arithmetic of bn_fold_kernel (bit-identical), so packing does not wait for the fold.
//   * fold blocks: 256 channels of scale / shift / invstd each.
constexpr int PREP_MAXI = 30;
struct PrepItem {
  const float* w;
  bf16_t* w_fwd;
  bf16_t* w_dgrad;
  const float* gamma;     // eval-mode BN behind the conv (NULL: none — scale 1, no fold outputs)
  const float* bnbeta;
  const float* mean;
  const float* var;
  float* fold;            // [3][Cout]: scale, shift, invstd
  long long s_o, s_i, s_h, s_w;
  int Cout, Cin, kh, kw;
  float eps;
  int tiles;              // pack blocks of this item; fold blocks follow them
};
struct PrepGroup {
  int nitems, reserved;
  int blk_start[PREP_MAXI + 2];
  PrepItem it[PREP_MAXI];
};
static_assert(sizeof(PrepGroup) <= 4096, "kernel-argument block too large");

template <bool F16>
__global__ __launch_bounds__(256) void prepare_group_kernel(const PrepGroup grp) {
  __shared__ float tile[64][65];
  int idx = 0;
#pragma unroll
  for (int i = 1; i < PREP_MAXI; ++i) idx += ((int)blockIdx.x >= grp.blk_start[i]) ? 1 : 0;
  const PrepItem& p = grp.it[idx];
  const int lb = (int)blockIdx.x - grp.blk_start[idx];
  const int tid = threadIdx.x;
  const int Cout = p.Cout, Cin = p.Cin;
  if (lb >= p.tiles) {   // ---- fold block ----
    const int c = (lb - p.tiles) * 256 + tid;
    if (c >= Cout || !p.gamma) return;
    const float is = bn_invstd(p.var[c], p.eps);
    const float sc = p.gamma[c] * is;
    p.fold[c] = sc;
    p.fold[Cout + c] = p.bnbeta[c] - p.mean[c] * sc;
    p.fold[2 * Cout + c] = is;
    return;
  }
  // ---- pack block: tile (to, ti) of tap (y, x) ----
  const int tiles_i = Cin >> 6, tiles_o = Cout >> 6;
  const int ti = lb % tiles_i;
  int r = lb / tiles_i;
  const int to = r % tiles_o;
  const int tap = r / tiles_o;
  const int y = tap / p.kw, x = tap - y * p.kw;
  const int o0 = to * 64, i0 = ti * 64;
  const float* src = p.w + (long long)y * p.s_h + (long long)x * p.s_w;
  // read 64 x 64 with the unit-stride axis across the lanes
  if (p.s_i == 1) {
    const int c4 = (tid & 15) * 4, r0 = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int o = r0 + rr * 16;
      const float* q = src + (long long)(o0 + o) * p.s_o + (i0 + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[o][c4 + e] = q[e];
    }
  } else {
    // ci is strided (contiguous OIHW 3x3: s_i = kh*kw): lanes walk co... no axis is unit-stride within a tap either,
    // so take ci across the lanes and accept the strided gather (these parameters are small or 1x1)
    const int c = tid & 63, r0 = tid >> 6;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int o = r0 + rr * 4;
      tile[o][c] = src[(long long)(o0 + o) * p.s_o + (long long)(i0 + c) * p.s_i];
    }
  }
  __syncthreads();
  const int ntap = p.kh * p.kw;
  {   // w_fwd[co][tap][ci]: 4 consecutive ci per thread
    const int c4 = (tid & 15) * 4, r0 = tid >> 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int o = r0 + rr * 16;
      bf16x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = f32_to_elem<F16>(tile[o][c4 + e]);
      *(bf16x4_t*)(p.w_fwd + ((long long)(o0 + o) * ntap + tap) * Cin + i0 + c4) = v;
    }
  }
  if (p.w_dgrad) {   // w_dgrad[ci][tap][co] = elem(elem(w) * scale[co]): 4 consecutive co per thread
    const int o4 = (tid & 15) * 4, r0 = tid >> 4;
    float sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = o0 + o4 + e;
      sc[e] = p.gamma ? p.gamma[co] * bn_invstd(p.var[co], p.eps) : 1.f;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = r0 + rr * 16;
      bf16x4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = f32_to_elem<F16>(mul_f32_rounded(elem_to_f32<F16>(f32_to_elem<F16>(tile[o4 + e][i])), sc[e]));
      *(bf16x4_t*)(p.w_dgrad + ((long long)(i0 + i) * ntap + tap) * Cout + o0 + o4) = v;
    }
  }
}

extern "C" int tdn_prepare_group(const tdn_prep_item* items, int n, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(items != nullptr && n > 0, "tdn_prepare_group: no items");
  for (int base = 0; base < n; base += PREP_MAXI) {
    PrepGroup g;
    memset(&g, 0, sizeof(g));
    const int cnt = (n - base < PREP_MAXI) ? n - base : PREP_MAXI;
    g.nitems = cnt;
    for (int j = 0; j < PREP_MAXI + 2; ++j) g.blk_start[j] = INT_MAX;
    int blk = 0;
    for (int j = 0; j < cnt; ++j) {
      const tdn_prep_item& s = items[base + j];
      TDN_CHECK(s.w && s.w_fwd, "tdn_prepare_group: item %d: NULL weight pointer", base + j);
      TDN_CHECK(s.Cout > 0 && s.Cin > 0 && s.Cout % 64 == 0 && s.Cin % 64 == 0 && s.kh > 0 && s.kw > 0 && s.kh <= 7 &&
                    s.kw <= 7,
                "tdn_prepare_group: item %d: channels must be multiples of 64 (Cout=%d Cin=%d k=%dx%d)", base + j,
                s.Cout, s.Cin, s.kh, s.kw);
      TDN_CHECK(!s.gamma || (s.beta && s.mean && s.var && s.fold),
                "tdn_prepare_group: item %d: BN fold needs beta, mean, var and the (3, Cout) output", base + j);
      PrepItem& d = g.it[j];
      d.w = s.w; d.w_fwd = (bf16_t*)s.w_fwd; d.w_dgrad = (bf16_t*)s.w_dgrad;
      d.gamma = s.gamma; d.bnbeta = s.beta; d.mean = s.mean; d.var = s.var; d.fold = s.fold;
      d.s_o = s.s_o; d.s_i = s.s_i; d.s_h = s.s_h; d.s_w = s.s_w;
      d.Cout = s.Cout; d.Cin = s.Cin; d.kh = s.kh; d.kw = s.kw; d.eps = s.eps;
      d.tiles = (s.Cout / 64) * (s.Cin / 64) * s.kh * s.kw;
      g.blk_start[j] = blk;
      blk += d.tiles + (s.gamma ? ceil_div(s.Cout, 256) : 0);
    }
    TDN_LAUNCH_T(prepare_group_kernel, dtype, dim3(blk), dim3(256), (hipStream_t)stream, g);
    TDN_LAUNCH_CHECK();
  }
  return 0;
}

template <bool F16>
__global__ void pack_stem_kernel(const float* w, int Cout, bf16_t* w_fwd) {
  const int total = Cout * 7 * 8 * 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, co = i / 224;
    float v = 0.f;
    if (c < 3 && kw < 7) v = w[co * 147 + c * 49 + kh * 7 + kw];
    w_fwd[i] = f32_to_elem<F16>(v);
  }
}

extern "C" int tdn_pack_stem_weight(const float* w, int Cout, void* w_fwd, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(w && w_fwd && Cout > 0, "tdn_pack_stem_weight: bad arguments");
  TDN_LAUNCH_T(pack_stem_kernel, dtype, dim3(grid_for(Cout * 224, 256)), dim3(256), (hipStream_t)stream, w, Cout,
                     (bf16_t*)w_fwd);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- image staging: NCHW fp32 -> zero-padded NHWC4 bf16 [N][H+6][W+8][4] ----------------------
template <bool F16>
__global__ void stage_image_kernel(const float* img, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N,
                                   int H, int W, bf16_t* xp) {
  const int Hp = H + 6, Wp = W + 8;
  const int64_t total = (int64_t)N * Hp * Wp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int wp, hp, n;
    split_idx(i, Wp, Hp, wp, hp, n);
    const int h = hp - 3, w = wp - 3;
    bf16x4_t v = {f32_to_elem<F16>(0.f), f32_to_elem<F16>(0.f), f32_to_elem<F16>(0.f), f32_to_elem<F16>(0.f)};
    if (h >= 0 && h < H && w >= 0 && w < W) {
      const float* s = img + n * s_n + h * s_h + w * s_w;
      v[0] = f32_to_elem<F16>(s[0]);
      v[1] = f32_to_elem<F16>(s[s_c]);
      v[2] = f32_to_elem<F16>(s[2 * s_c]);
    }
    *(bf16x4_t*)(xp + i * 4) = v;
  }
}

extern "C" int tdn_stage_image(const float* img, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N, int H,
                               int W, void* xp, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(img && xp && N > 0 && H > 0 && W > 0, "tdn_stage_image: bad arguments");
  const int64_t total = (int64_t)N * (H + 6) * (W + 8);
  TDN_LAUNCH_T(stage_image_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream, img, s_n,
                     s_c, s_h, s_w, N, H, W, (bf16_t*)xp);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- max pool 3x3 s2 p1 (NHWC, 8 channels per lane) --------------------------------------------
template <bool F16>
__global__ void maxpool_fwd_kernel(const bf16_t* x, bf16_t* y, uint8_t* idx, int N, int H, int W, int C, int Ho,
                                   int Wo) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, wo, ho, n;
    split_idx(i, C8, Wo, Ho, c8, wo, ho, n);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if (w < 0 || w >= W) continue;
        const bf16x8_t v = *(const bf16x8_t*)(x + (((int64_t)n * H + h) * W + w) * C + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = elem_to_f32<F16>(v[e]);
          // PyTorch rule: first maximum in (kh, kw) scan order wins; NaN propagates
          if (bi[e] < 0 || f > best[e] || f != f) { best[e] = f; bi[e] = kh * 3 + kw; }
        }
      }
    }
    bf16x8_t o;
    uint64_t packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = f32_to_elem<F16>(best[e]); packed |= (uint64_t)(uint8_t)bi[e] << (8 * e); }
    *(bf16x8_t*)(y + i * 8) = o;
    *(uint64_t*)(idx + i * 8) = packed;
  }
}

extern "C" int tdn_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype,
                                    void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "tdn_maxpool3x3s2_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
  TDN_LAUNCH_T(maxpool_fwd_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, Ho, Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Gather form of the adjoint: each input element sums dy of the (<= 4) windows that selected it.
// ReLU in front of the pool: `mask` is the pool's full-size input (gradient passes where it is > 0), or `ypool` is the
// pool's OUTPUT — a window's value is the value of the element it selected, so "selected element > 0" can be read
// from the 4x smaller tensor (same result bit for bit: all windows that selected one element carry its value).
// A thread owns a 2 x 2 quad of input pixels (8 channels): its four pixels can only have been selected by the four
// windows (k, k+1) x (m, m+1), whose idx / dy / y words are loaded ONCE (2.25 window loads per pixel before), and each
// pixel sums its windows in ascending (ho, wo) order — PyTorch's accumulation order.
template <bool F16>
__global__ void maxpool_bwd_kernel(const bf16_t* dy, const uint8_t* idx, const bf16_t* mask, const bf16_t* ypool,
                                   bf16_t* dx, int N, int H, int W, int C, int Ho, int Wo) {
  const int C8 = C >> 3;
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const int64_t total = (int64_t)N * H2 * W2 * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, m, k, n;
    split_idx(i, C8, W2, H2, c8, m, k, n);
    // window (a, b) = (k + a, m + b): words, with code 0xff (matches no position) where the window does not exist
    uint64_t pk[2][2];
    bf16x8_t g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = k + a, wo = m + b;
        pk[a][b] = ~0ull;
        g[a][b] = (bf16x8_t){};
        if (ho < Ho && wo < Wo) {
          const int64_t o = ((((int64_t)n * Ho + ho) * Wo + wo) * C8 + c8) * 8;
          uint64_t p8 = *(const uint64_t*)(idx + o);
          g[a][b] = *(const bf16x8_t*)(dy + o);
          if (ypool) {   // ReLU in front of the pool: a window whose value is <= 0 passes nothing
            const bf16x8_t yv = *(const bf16x8_t*)(ypool + o);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (!(elem_to_f32<F16>(yv[e]) > 0.f)) p8 |= 0xffull << (8 * e);
          }
          pk[a][b] = p8;
        }
      }
    // pixel (2k + r, 2m + s) is position (kh, kw) = (r + 1 - 2a, s + 1 - 2b) of window (a, b) when that is in [0, 2]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const int h = 2 * k + r, w = 2 * m + s_;
        if (h >= H || w >= W) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int kh = r + 1 - 2 * a, kw = s_ + 1 - 2 * b;
            if (kh < 0 || kw < 0) continue;            // compile-time after unrolling
            const int code = kh * 3 + kw;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if ((int)((pk[a][b] >> (8 * e)) & 0xff) == code) acc[e] += elem_to_f32<F16>(g[a][b][e]);
          }
        const int64_t px = (((int64_t)n * H + h) * W + w) * C8 + c8;
        bf16x8_t o8;
        if (mask) {
          const bf16x8_t mk = *(const bf16x8_t*)(mask + px * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = f32_to_elem<F16>((elem_to_f32<F16>(mk[e]) > 0.f) ? acc[e] : 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = f32_to_elem<F16>(acc[e]);
        }
        *(bf16x8_t*)(dx + px * 8) = o8;
      }
  }
}

extern "C" int tdn_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, const void* mask_src, void* dx, int N,
                                    int H, int W, int C, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "tdn_maxpool3x3s2_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);   // one thread per 2 x 2 input quad
  TDN_LAUNCH_T(maxpool_bwd_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream,
                     (const bf16_t*)dy, idx, (const bf16_t*)mask_src, (const bf16_t*)nullptr, (bf16_t*)dx, N, H, W, C, Ho,
                     Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_maxpool3x3s2_relu_bwd(const void* dy, const uint8_t* idx, const void* y_pooled, void* dx, int N,
                                         int H, int W, int C, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(dy && idx && y_pooled && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0,
            "tdn_maxpool3x3s2_relu_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);   // one thread per 2 x 2 input quad
  TDN_LAUNCH_T(maxpool_bwd_kernel, dtype, dim3(grid_for(total, 256)), dim3(256), (hipStream_t)stream,
                     (const bf16_t*)dy, idx, (const bf16_t*)nullptr, (const bf16_t*)y_pooled, (bf16_t*)dx, N, H, W, C, Ho,
                     Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- stride-2 subsample (F.max_pool2d(x, 1, stride=2)) -------------------------------------------
__global__ void subsample_fwd_kernel(const bf16_t* x, bf16_t* y, int N, int H, int W, int C, int Ho, int Wo) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)N * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, wo, ho, n;
    split_idx(i, C8, Wo, Ho, c8, wo, ho, n);
    *(bf16x8_t*)(y + i * 8) = *(const bf16x8_t*)(x + (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c8 * 8);
  }
}

extern "C" int tdn_subsample2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "tdn_subsample2_fwd: bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  TDN_LAUNCH(subsample_fwd_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <bool F16>
__global__ void subsample_bwd_kernel(const bf16_t* dy, const bf16_t* dx_in, bf16_t* dx, int N, int H, int W, int C,
                                     int Ho, int Wo) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)N * H * W * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, w, h, n;
    split_idx(i, C8, W, H, c8, w, h, n);
    float acc[8];
    if (dx_in) {
      const bf16x8_t v = *(const bf16x8_t*)(dx_in + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = elem_to_f32<F16>(v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    }
    if (((h | w) & 1) == 0) {
      const bf16x8_t g = *(const bf16x8_t*)(dy + ((((int64_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C8 + c8) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += elem_to_f32<F16>(g[e]);
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f32_to_elem<F16>(acc[e]);
    *(bf16x8_t*)(dx + i * 8) = o;
  }
}

extern "C" int tdn_subsample2_bwd(const void* dy, const void* dx_in, void* dx, int N, int H, int W, int C, int dtype,
                                  void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "tdn_subsample2_bwd: bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  TDN_LAUNCH_T(subsample_bwd_kernel, dtype, dim3(grid_for((int64_t)N * H * W * (C / 8), 256)), dim3(256), (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)dx_in, (bf16_t*)dx, N, H, W, C, Ho, Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- out = (a + b) masked by mask > 0 ---------------------------------------------------------------
template <bool F16>
__global__ void add_relu_mask_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* mask, bf16_t* out, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8_t va = *(const bf16x8_t*)(a + i * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = elem_to_f32<F16>(va[e]);
    if (b) {
      const bf16x8_t vb = *(const bf16x8_t*)(b + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += elem_to_f32<F16>(vb[e]);
    }
    if (mask) {
      const bf16x8_t mk = *(const bf16x8_t*)(mask + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (elem_to_f32<F16>(mk[e]) > 0.f) ? v[e] : 0.f;
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f32_to_elem<F16>(v[e]);
    *(bf16x8_t*)(out + i * 8) = o;
  }
}

extern "C" int tdn_add_relu_mask(const void* a, const void* b, const void* mask_src, void* out, int64_t n, int dtype,
                                 void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(a && out && n > 0 && n % 8 == 0, "tdn_add_relu_mask: bad arguments (n=%lld)", (long long)n);
  TDN_LAUNCH_T(add_relu_mask_kernel, dtype, dim3(grid_for(n / 8, 256)), dim3(256), (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)mask_src, (bf16_t*)out, n / 8);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- ConvModule activation / pre-activation pieces (layers.py:57-135 of the reference) ----------------
// ReLU6 (nn.ReLU6 = hardtanh(0, 6)): the conv epilogues clamp at 0; the upper clamp and the backward mask
// (gradient passes where 0 < y < hi, read from the saved OUTPUT) are these two element-wise passes.
template <bool F16>
__global__ void clamp_max_kernel(bf16_t* y, float hi, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    bf16x8_t v = *(const bf16x8_t*)(y + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f32_to_elem<F16>(fminf(elem_to_f32<F16>(v[e]), hi));
    *(bf16x8_t*)(y + i * 8) = v;
  }
}

extern "C" int tdn_clamp_max(void* y, float hi, int64_t n, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(y && n > 0 && n % 8 == 0, "tdn_clamp_max: bad arguments (n=%lld)", (long long)n);
  TDN_LAUNCH_T(clamp_max_kernel, dtype, dim3(grid_for(n / 8, 256)), dim3(256), (hipStream_t)stream, (bf16_t*)y, hi,
               n / 8);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <bool F16>
__global__ void act_mask_kernel(const bf16_t* g, const bf16_t* y, bf16_t* out, float hi, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const bf16x8_t vg = *(const bf16x8_t*)(g + i * 8);
    const bf16x8_t vy = *(const bf16x8_t*)(y + i * 8);
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = elem_to_f32<F16>(vy[e]);
      o[e] = (t > 0.f && t < hi) ? vg[e] : f32_to_elem<F16>(0.f);
    }
    *(bf16x8_t*)(out + i * 8) = o;
  }
}

extern "C" int tdn_act_mask(const void* g, const void* y, void* out, float hi, int64_t n, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(g && y && out && n > 0 && n % 8 == 0, "tdn_act_mask: bad arguments (n=%lld)", (long long)n);
  TDN_LAUNCH_T(act_mask_kernel, dtype, dim3(grid_for(n / 8, 256)), dim3(256), (hipStream_t)stream, (const bf16_t*)g,
               (const bf16_t*)y, (bf16_t*)out, hi, n / 8);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Pre-activation order (activate_last=False): y = act(x * scale[c] + shift[c]) on the conv's INPUT — BatchNorm2d in
// eval mode folded to a per-channel affine, act = none / ReLU / ReLU6.  NHWC, C % 8 == 0.
template <bool F16>
__global__ void channel_affine_kernel(const bf16_t* x, const float* scale, const float* shift, bf16_t* y,
                                      int64_t n8, int C8, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (i < (1ll << 31) ? (int)((unsigned)i % (unsigned)C8) : (int)(i % C8)) * 8;
    const bf16x8_t v = *(const bf16x8_t*)(x + i * 8);
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = elem_to_f32<F16>(v[e]) * scale[c + e] + shift[c + e];
      if (act >= 1) t = fmaxf(t, 0.f);
      if (act == 2) t = relu6_top<F16>(t);
      o[e] = f32_to_elem<F16>(t);
    }
    *(bf16x8_t*)(y + i * 8) = o;
  }
}

extern "C" int tdn_channel_affine_fwd(const void* x, const float* scale, const float* shift, void* y, int64_t npix,
                                      int C, int act, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(x && scale && shift && y && npix > 0 && C > 0 && C % 8 == 0 && act >= 0 && act <= 2,
            "tdn_channel_affine_fwd: bad arguments (npix=%lld C=%d act=%d)", (long long)npix, C, act);
  const int64_t n8 = npix * (C / 8);
  TDN_LAUNCH_T(channel_affine_kernel, dtype, dim3(grid_for(n8, 256)), dim3(256), (hipStream_t)stream,
               (const bf16_t*)x, scale, shift, (bf16_t*)y, n8, C / 8, act);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Backward of the same affine from g = dL/dy already masked by the activation:
//   dx = g * scale[c];   dbeta[c] = sum_p g;   dgamma[c] = invstd[c] * sum_p g * (x - mean[c])
// Pass 1: one block per chunk of CHUNK pixels, threads = channel lanes (8 channels each) x pixel lanes, the pixel
// lanes combined through LDS in lane order, partial sums to the workspace [chunks][2][C].  Pass 2 adds the chunks in
// order (deterministic).
static constexpr int kAffineChunk = 512;

template <bool F16>
__global__ __launch_bounds__(256) void channel_affine_bwd_kernel(const bf16_t* g, const bf16_t* x, const float* scale,
                                                                 const float* mean, bf16_t* dx, float* partial,
                                                                 int64_t npix, int C) {
  __shared__ float red[256 * 16];
  const int cl = C / 8;                       // channel lanes
  const int pl = cl >= 256 ? 1 : 256 / cl;    // pixel lanes
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * kAffineChunk;
  const int64_t p1 = p0 + kAffineChunk < npix ? p0 + kAffineChunk : npix;
  for (int cbase = 0; cbase < cl; cbase += 256) {   // C > 2048: several rounds of channel lanes
    const int lane_c = cbase + tid % (cl < 256 ? cl : 256);
    const int lane_p = tid / (cl < 256 ? cl : 256);
    const bool active = lane_p < pl && lane_c < cl;
    float s0[8], s1[8], sc[8], mu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[e] = scale[lane_c * 8 + e];
        mu[e] = mean[lane_c * 8 + e];
      }
      for (int64_t p = p0 + lane_p; p < p1; p += pl) {
        const bf16x8_t vg = *(const bf16x8_t*)(g + (p * cl + lane_c) * 8);
        const bf16x8_t vx = *(const bf16x8_t*)(x + (p * cl + lane_c) * 8);
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gv = elem_to_f32<F16>(vg[e]);
          s0[e] += gv;
          s1[e] += gv * (elem_to_f32<F16>(vx[e]) - mu[e]);
          o[e] = f32_to_elem<F16>(gv * sc[e]);
        }
        *(bf16x8_t*)(dx + (p * cl + lane_c) * 8) = o;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[tid * 16 + e] = s0[e];
      red[tid * 16 + 8 + e] = s1[e];
    }
    __syncthreads();
    if (active && lane_p == 0) {
      const int w = cl < 256 ? cl : 256;
      for (int j = 1; j < pl; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s0[e] += red[(j * w + tid) * 16 + e];
          s1[e] += red[(j * w + tid) * 16 + 8 + e];
        }
      float* out = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        out[lane_c * 8 + e] = s0[e];
        out[C + lane_c * 8 + e] = s1[e];
      }
    }
    __syncthreads();
  }
}

__global__ void channel_affine_bwd_reduce_kernel(const float* partial, int chunks, int C, const float* invstd,
                                                 float* dgamma, float* dbeta, float beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < chunks; ++k) {
    s0 += partial[(int64_t)k * 2 * C + c];
    s1 += partial[(int64_t)k * 2 * C + C + c];
  }
  const float dg = s1 * invstd[c];
  dbeta[c] = beta != 0.f ? beta * dbeta[c] + s0 : s0;
  dgamma[c] = beta != 0.f ? beta * dgamma[c] + dg : dg;
}

extern "C" int64_t tdn_channel_affine_bwd_workspace(int64_t npix, int C) {
  return ((npix + kAffineChunk - 1) / kAffineChunk) * 2 * C * 4;
}

extern "C" int tdn_channel_affine_bwd(const void* g, const void* x, const float* scale, const float* mean,
                                      const float* invstd, void* dx, float* dgamma, float* dbeta, float beta,
                                      int64_t npix, int C, void* workspace, int64_t workspace_bytes, int dtype,
                                      void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(g && x && scale && mean && invstd && dx && dgamma && dbeta && workspace && npix > 0 && C > 0 && C % 8 == 0,
            "tdn_channel_affine_bwd: bad arguments (npix=%lld C=%d)", (long long)npix, C);
  TDN_CHECK(workspace_bytes >= tdn_channel_affine_bwd_workspace(npix, C), "tdn_channel_affine_bwd: workspace too small");
  const int chunks = (int)((npix + kAffineChunk - 1) / kAffineChunk);
  TDN_LAUNCH_T(channel_affine_bwd_kernel, dtype, dim3(chunks), dim3(256), (hipStream_t)stream, (const bf16_t*)g,
               (const bf16_t*)x, scale, mean, (bf16_t*)dx, (float*)workspace, npix, C);
  TDN_LAUNCH_CHECK();
  TDN_LAUNCH(channel_affine_bwd_reduce_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, chunks, C, invstd, dgamma, dbeta, beta);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- boundary layout converters -----------------------------------------------------------------------
template <bool F16>
__global__ void nchw_to_nhwc_kernel(const float* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N,
                                    int C, int H, int W, bf16_t* dst) {
  // tile transpose through LDS: 32 pixels x 32 channels per block step
  __shared__ float t[32][33];
  const int HW = H * W;
  const int tiles_p = ceil_div(HW, 32), tiles_c = ceil_div(C, 32);
  const int64_t ntiles = (int64_t)N * tiles_p * tiles_c;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tc = (int)(tile % tiles_c);
    int64_t r = tile / tiles_c;
    const int tp = (int)(r % tiles_p);
    const int n = (int)(r / tiles_p);
    for (int k = ty; k < 32; k += 8) {
      const int c = tc * 32 + k, pix = tp * 32 + tx;
      float v = 0.f;
      if (c < C && pix < HW) {
        const int h = pix / W, w = pix - h * W;
        v = src[n * s_n + c * s_c + h * s_h + w * s_w];
      }
      t[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int pix = tp * 32 + k, c = tc * 32 + tx;
      if (c < C && pix < HW) dst[((int64_t)n * HW + pix) * C + c] = f32_to_elem<F16>(t[tx][k]);
    }
    __syncthreads();
  }
}

extern "C" int tdn_nchw_f32_to_nhwc(const float* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N,
                                    int C, int H, int W, void* dst, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "tdn_nchw_f32_to_nhwc: bad arguments");
  const int64_t ntiles = (int64_t)N * ceil_div(H * W, 32) * ceil_div(C, 32);
  TDN_LAUNCH_T(nchw_to_nhwc_kernel, dtype, dim3((int)(ntiles < 8192 ? ntiles : 8192)), dim3(256), (hipStream_t)stream, src, s_n, s_c, s_h, s_w, N, C, H, W, (bf16_t*)dst);
  TDN_LAUNCH_CHECK();
  return 0;
}

// The same transpose for a source that already holds 16-bit elements (a cotangent or feature map handed in as a
// plain NCHW-contiguous tensor): a pure move of 2-byte values, dtype-agnostic.  Keeps the whole boundary conversion
// inside the library, so a recorded launch plan (plan.hip) contains it.
__global__ void nchw16_to_nhwc_kernel(const unsigned short* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w,
                                      int N, int C, int H, int W, unsigned short* dst) {
  __shared__ unsigned short t[32][34];
  const int HW = H * W;
  const int tiles_p = ceil_div(HW, 32), tiles_c = ceil_div(C, 32);
  const int64_t ntiles = (int64_t)N * tiles_p * tiles_c;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tc = (int)(tile % tiles_c);
    int64_t r = tile / tiles_c;
    const int tp = (int)(r % tiles_p);
    const int n = (int)(r / tiles_p);
    for (int k = ty; k < 32; k += 8) {
      const int c = tc * 32 + k, pix = tp * 32 + tx;
      unsigned short v = 0;
      if (c < C && pix < HW) {
        const int h = pix / W, w = pix - h * W;
        v = src[n * s_n + c * s_c + h * s_h + w * s_w];
      }
      t[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int pix = tp * 32 + k, c = tc * 32 + tx;
      if (c < C && pix < HW) dst[((int64_t)n * HW + pix) * C + c] = t[tx][k];
    }
    __syncthreads();
  }
}

extern "C" int tdn_nchw16_to_nhwc(const void* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N, int C,
                                  int H, int W, void* dst, void* stream) {
  TDN_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "tdn_nchw16_to_nhwc: bad arguments");
  const int64_t ntiles = (int64_t)N * ceil_div(H * W, 32) * ceil_div(C, 32);
  TDN_LAUNCH(nchw16_to_nhwc_kernel, dim3((int)(ntiles < 8192 ? ntiles : 8192)), dim3(256), 0, (hipStream_t)stream,
             (const unsigned short*)src, s_n, s_c, s_h, s_w, N, C, H, W, (unsigned short*)dst);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <bool F16>
__global__ void nhwc_to_nchw_kernel(const bf16_t* src, int N, int C, int H, int W, float* dst) {
  __shared__ float t[32][33];
  const int HW = H * W;
  const int tiles_p = ceil_div(HW, 32), tiles_c = ceil_div(C, 32);
  const int64_t ntiles = (int64_t)N * tiles_p * tiles_c;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tc = (int)(tile % tiles_c);
    int64_t r = tile / tiles_c;
    const int tp = (int)(r % tiles_p);
    const int n = (int)(r / tiles_p);
    for (int k = ty; k < 32; k += 8) {
      const int pix = tp * 32 + k, c = tc * 32 + tx;
      t[k][tx] = (c < C && pix < HW) ? elem_to_f32<F16>(src[((int64_t)n * HW + pix) * C + c]) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int c = tc * 32 + k, pix = tp * 32 + tx;
      if (c < C && pix < HW) dst[((int64_t)n * C + c) * HW + pix] = t[tx][k];
    }
    __syncthreads();
  }
}

extern "C" int tdn_nhwc_to_nchw_f32(const void* src, int N, int C, int H, int W, float* dst, int dtype,
                                    void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "tdn_nhwc_to_nchw_f32: bad arguments");
  const int64_t ntiles = (int64_t)N * ceil_div(H * W, 32) * ceil_div(C, 32);
  TDN_LAUNCH_T(nhwc_to_nchw_kernel, dtype, dim3((int)(ntiles < 8192 ? ntiles : 8192)), dim3(256), (hipStream_t)stream, (const bf16_t*)src, N, C, H, W, dst);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- image batch staging: normalize + horizontal flip + zero pad + HWC -> (NCHW f32 | staged NHWC4) + collate ----
// Replaces, for already-resized pixels, datasets/utils/image.py:87-105 (img_normalize), :220-249 (img_flip),
// :300-347 (img_pad_size_divisor), dataset_transforms.py:44 (HWC -> CHW) and loader/collate.py:42-63 (pad to the
// batch maximum, padding_value 0, stack).  (x - mean) / std as two IEEE fp32 operations: bit-identical to numpy.
struct CollateArgs {
  const void* img[TDN_COLLATE_MAX];
  int h[TDN_COLLATE_MAX], w[TDN_COLLATE_MAX];
  int flip[TDN_COLLATE_MAX];
  float mean[3], stdv[3];
  int n, Hb, Wb;
};

template <bool SRC_F32>
__device__ __forceinline__ void collate_pixel(const CollateArgs& a, int n, int y, int x, float v[3]) {
  v[0] = v[1] = v[2] = 0.f;
  if (y < a.h[n] && x < a.w[n]) {
    const int sx = a.flip[n] ? a.w[n] - 1 - x : x;
    const int64_t o = ((int64_t)y * a.w[n] + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = SRC_F32 ? ((const float*)a.img[n])[o + c] : (float)((const uint8_t*)a.img[n])[o + c];
      v[c] = __fdiv_rn(__fsub_rn(s, a.mean[c]), a.stdv[c]);
    }
  }
}

template <bool SRC_F32>
__global__ void collate_nchw_kernel(const CollateArgs a, float* out) {
  const int64_t plane = (int64_t)a.Hb * a.Wb, total = plane * a.n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / plane);
    const int64_t r = i - n * plane;
    const int y = (int)(r / a.Wb), x = (int)(r - (int64_t)y * a.Wb);
    float v[3];
    collate_pixel<SRC_F32>(a, n, y, x, v);
    float* o = out + (int64_t)n * 3 * plane + r;
    o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
  }
}

template <bool SRC_F32, bool F16>
__global__ void collate_staged_kernel(const CollateArgs a, bf16_t* xp) {
  const int Hp = a.Hb + 6, Wp = a.Wb + 8;
  const int64_t total = (int64_t)a.n * Hp * Wp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int wp, hp, n;
    split_idx(i, Wp, Hp, wp, hp, n);
    const int y = hp - 3, x = wp - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (y >= 0 && y < a.Hb && x >= 0 && x < a.Wb) collate_pixel<SRC_F32>(a, n, y, x, v);
    bf16x4_t o = {f32_to_elem<F16>(v[0]), f32_to_elem<F16>(v[1]), f32_to_elem<F16>(v[2]), f32_to_elem<F16>(0.f)};
    *(bf16x4_t*)(xp + i * 4) = o;
  }
}

extern "C" int tdn_collate_images(const void* const* imgs, const int32_t* hw, const uint8_t* flip, int N,
                                  int src_kind, const float* mean3, const float* std3, int Hb, int Wb, void* out,
                                  int out_kind, int dtype, void* stream) {
  TDN_CHECK(imgs && hw && mean3 && std3 && out, "tdn_collate_images: NULL pointer");
  TDN_CHECK(N > 0 && N <= TDN_COLLATE_MAX, "tdn_collate_images: N=%d outside 1..%d (split the batch)", N,
            TDN_COLLATE_MAX);
  TDN_CHECK(src_kind == 0 || src_kind == 1, "tdn_collate_images: src_kind %d (0 = u8 HWC, 1 = f32 HWC)", src_kind);
  TDN_CHECK(out_kind == 0 || out_kind == 1, "tdn_collate_images: out_kind %d (0 = f32 NCHW, 1 = staged)", out_kind);
  TDN_CHECK(Hb > 0 && Wb > 0, "tdn_collate_images: batch size %dx%d", Hb, Wb);
  if (out_kind == 1) TDN_CHECK_DTYPE(dtype);
  CollateArgs a;
  for (int i = 0; i < N; ++i) {
    TDN_CHECK(imgs[i] != nullptr, "tdn_collate_images: image %d is NULL", i);
    a.img[i] = imgs[i];
    a.h[i] = hw[2 * i];
    a.w[i] = hw[2 * i + 1];
    a.flip[i] = flip ? flip[i] : 0;
    TDN_CHECK(a.h[i] > 0 && a.w[i] > 0 && a.h[i] <= Hb && a.w[i] <= Wb,
              "tdn_collate_images: image %d is %dx%d, batch is %dx%d", i, a.h[i], a.w[i], Hb, Wb);
  }
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
  a.n = N; a.Hb = Hb; a.Wb = Wb;
  hipStream_t st = (hipStream_t)stream;
  if (out_kind == 0) {
    const int64_t total = (int64_t)N * Hb * Wb;
    if (src_kind) TDN_LAUNCH(collate_nchw_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, st, a, (float*)out);
    else TDN_LAUNCH(collate_nchw_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, st, a, (float*)out);
  } else {
    const int64_t total = (int64_t)N * (Hb + 6) * (Wb + 8);
    const dim3 g(grid_for(total, 256)), b(256);
    const bool f16 = dtype == TDN_F16;
    if (src_kind && f16) TDN_LAUNCH((collate_staged_kernel<true, true>), g, b, 0, st, a, (bf16_t*)out);
    else if (src_kind) TDN_LAUNCH((collate_staged_kernel<true, false>), g, b, 0, st, a, (bf16_t*)out);
    else if (f16) TDN_LAUNCH((collate_staged_kernel<false, true>), g, b, 0, st, a, (bf16_t*)out);
    else TDN_LAUNCH((collate_staged_kernel<false, false>), g, b, 0, st, a, (bf16_t*)out);
  }
  TDN_LAUNCH_CHECK();
  return 0;
}
