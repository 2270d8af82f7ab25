// Stem of the backbone in ONE launch: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded eval-mode BatchNorm + ReLU +
// MaxPool2d(3, stride 2, padding 1)   (models/backbone/resnet.py:214-218, 254-258 of the reference).
//
// Separately the two launches write and re-read the stem's full-size activation (N x H/2 x W/2 x 64: 69 MB at the
// BASELINE batch) at the very head of the step, where nothing runs beside them; the backward pass needs only the
// pooled output, the window indices and the staged image (the ReLU mask is read from the pooled output,
// tdn_maxpool3x3s2_relu_bwd).  Here a workgroup owns a 4 x 16 patch of POOLED pixels: it computes the 9 x 33 conv
// pixels under it on the matrix cores (one 16x16x32 MFMA per kernel row and 16 pixels x 16 channels — the same
// K order and operand layout as the generic kernel's stem instantiation, so every conv value is the same bit
// pattern), applies BN + ReLU, rounds to the 16-bit element type (what the separate conv stores), keeps that patch
// in LDS and pools it — values and first-maximum indices identical to tdn_stem_conv_fwd + tdn_maxpool3x3s2_fwd.
// The conv patch overlaps its neighbours by one row / column: 1.16x the conv work for none of its HBM traffic.
#include "common.h"

namespace {
constexpr int SP_PH = 4, SP_PW = 16;                          // pooled patch
constexpr int SP_CH = 2 * SP_PH + 1, SP_CW = 2 * SP_PW + 1;   // conv patch: 9 x 33
constexpr int SP_NPIX = SP_CH * SP_CW;                        // 297
constexpr int SP_NFRAG = (SP_NPIX + 15) / 16;                 // 19 fragments of 16 pixels
constexpr int SP_WAVES = 4;
constexpr int SP_ROWB = 64 * 2;                               // bytes of one conv pixel (64 channels) in LDS
constexpr int SP_IH = 2 * SP_CH + 5;                          // staged-image rows under the conv patch: 23
constexpr int SP_ICH = SP_CW + 3;                             // 16-byte chunks (2 pixels x 4 channels) per row: 36
constexpr int SP_IROWB = SP_ICH * 16;                         // 576 bytes
}  // namespace

template <bool F16>
__global__ __launch_bounds__(SP_WAVES * 64) void stem_pool_fwd_kernel(const bf16_t* __restrict__ xp,
                                                                     const bf16_t* __restrict__ w,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     bf16_t* __restrict__ y, uint8_t* __restrict__ idx,
                                                                     int H, int W, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) char sC[SP_NFRAG * 16 * SP_ROWB];   // [pixel][64 channels], 38 KB
  __shared__ __attribute__((aligned(16))) char sX[SP_IH * SP_IROWB];          // staged image under the patch, 13 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.z, ph0 = blockIdx.y * SP_PH, pw0 = blockIdx.x * SP_PW;
  const int Hc = H >> 1, Wc = W >> 1;                  // conv output grid
  const int Hp = H + 6, Wp = W + 8;                    // staged image
  const int fr = lane & 15, q = lane >> 4;

  // ---- the staged image under the conv patch -> LDS, every load in flight at once.  Row r is image row 4*ph0 - 2 + r,
  // chunk c is pixels 4*pw0 - 2 + 2c, +1; positions outside the staged image (they only feed conv pixels outside the
  // conv grid, which the pool never reads) are clamped to a valid address ----
  {
    const int row0 = 4 * ph0 - 2, chunk0 = 2 * pw0 - 1, nchunk = Wp >> 1;
    for (int i = tid; i < SP_IH * SP_ICH; i += SP_WAVES * 64) {
      const int r = i / SP_ICH, c = i - r * SP_ICH;
      const int gr = min(max(row0 + r, 0), Hp - 1), gc = min(max(chunk0 + c, 0), nchunk - 1);
      *(bf16x8_t*)(sX + r * SP_IROWB + c * 16) = *(const bf16x8_t*)(xp + (((size_t)n * Hp + gr) * Wp + 2 * gc) * 4);
    }
  }

  // ---- weights: all 7 kernel rows x 4 channel fragments in registers.  MFMA row R of fragment i is channel
  // 16 * (R >> 2) + 4 * i + (R & 3): lane (fr, q) then ends up with channels 16q .. 16q+15 of its pixel ----
  bf16x8_t wf[7][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = 16 * (fr >> 2) + 4 * i + (fr & 3);
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) wf[kh][i] = *(const bf16x8_t*)(w + (size_t)ch * 224 + kh * 32 + q * 8);
  }

  // lane's pixel of fragment f: LDS address of its kernel-row-0 operand (2 pixels x 4 channels = 16 bytes per lane);
  // pixel (py, px) of the patch reads image rows 2py + kh, chunks px + q
  auto pix_src = [&](int f) -> const char* {
    const int pid = min(f * 16 + fr, SP_NPIX - 1);     // the last fragment's padding lanes repeat its last pixel
    const int py = pid / SP_CW, px = pid - py * SP_CW;
    return sX + (2 * py) * SP_IROWB + (px + q) * 16;
  };
  auto load_frag = [&](const char* src, bf16x8_t (&xf)[7]) {
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) xf[kh] = *(const bf16x8_t*)(src + kh * SP_IROWB);
  };

  f32x4_t sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = *(const f32x4_t*)(scale + 16 * q + 4 * i);
    sh[i] = *(const f32x4_t*)(shift + 16 * q + 4 * i);
  }

  // ---- conv patch: fragments wave, wave + 4, ...; the next fragment's operands are in flight during the MFMAs ----
  __syncthreads();                                     // the image patch is in LDS
  bf16x8_t xa[7], xb[7];
  load_frag(pix_src(wave), xa);                        // SP_WAVES <= SP_NFRAG: every wave has a first fragment
  constexpr int ROUNDS = (SP_NFRAG + SP_WAVES - 1) / SP_WAVES;
#pragma unroll
  for (int it = 0; it < ROUNDS; ++it) {                // fully unrolled: the two operand sets alternate statically
    const int f = wave + it * SP_WAVES;                // wave-uniform
    if (f < SP_NFRAG) {
      bf16x8_t(&cur)[7] = (it & 1) ? xb : xa;
      bf16x8_t(&nxt)[7] = (it & 1) ? xa : xb;
      if (f + SP_WAVES < SP_NFRAG) load_frag(pix_src(f + SP_WAVES), nxt);
      f32x4_t acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma16<F16>(wf[kh][i], cur[kh], acc[i]);
      bf16x8_t o[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4_t v = acc[i] * sc[i] + sh[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i >> 1][(i & 1) * 4 + e] = f32_to_elem<F16>(fmaxf(v[e], 0.f));
      }
      char* dst = sC + (f * 16 + fr) * SP_ROWB + q * 32;
      *(bf16x8_t*)dst = o[0];
      *(bf16x8_t*)(dst + 16) = o[1];
    }
  }
  __syncthreads();

  // ---- pool: PyTorch's rule (first maximum in (kh, kw) scan order, NaN propagates), as tdn_maxpool3x3s2_fwd ----
  for (int it = tid; it < SP_PH * SP_PW * 8; it += SP_WAVES * 64) {
    const int c8 = it & 7, pp = it >> 3;
    const int pwl = pp & (SP_PW - 1), phl = pp / SP_PW;
    const int ph = ph0 + phl, pw = pw0 + pwl;
    if (ph >= Ho || pw >= Wo) continue;
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ph - 1 + kh;
      if (h < 0 || h >= Hc) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wq = 2 * pw - 1 + kw;
        if (wq < 0 || wq >= Wc) continue;
        const int pid = (2 * phl + kh) * SP_CW + 2 * pwl + kw;
        const bf16x8_t v = *(const bf16x8_t*)(sC + pid * SP_ROWB + c8 * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = elem_to_f32<F16>(v[e]);
          if (bi[e] < 0 || f > best[e] || f != f) { best[e] = f; bi[e] = kh * 3 + kw; }
        }
      }
    }
    bf16x8_t o;
    uint64_t packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = f32_to_elem<F16>(best[e]); packed |= (uint64_t)(uint8_t)bi[e] << (8 * e); }
    const size_t at = ((((size_t)n * Ho + ph) * Wo + pw) * 8 + c8) * 8;
    *(bf16x8_t*)(y + at) = o;
    *(uint64_t*)(idx + at) = packed;
  }
}

extern "C" int tdn_stem_pool_fwd(const void* xp, const void* w_stem, const float* scale, const float* shift, void* y,
                                 uint8_t* idx, int N, int H, int W, int Cout, int dtype, void* stream) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(xp && w_stem && scale && shift && y && idx, "tdn_stem_pool_fwd: NULL pointer");
  TDN_CHECK(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "tdn_stem_pool_fwd: needs even H, W (got %dx%d)", H, W);
  TDN_CHECK(Cout == 64, "tdn_stem_pool_fwd: the fused stem is built for 64 output channels (got %d)", Cout);
  TDN_CHECK(N <= 65535, "tdn_stem_pool_fwd: batch too large");
  const int Hc = H / 2, Wc = W / 2;
  const int Ho = (Hc + 2 - 3) / 2 + 1, Wo = (Wc + 2 - 3) / 2 + 1;
  const dim3 grid((Wo + SP_PW - 1) / SP_PW, (Ho + SP_PH - 1) / SP_PH, N);
  TDN_LAUNCH_T(stem_pool_fwd_kernel, dtype, grid, dim3(SP_WAVES * 64), (hipStream_t)stream, (const bf16_t*)xp,
               (const bf16_t*)w_stem, scale, shift, (bf16_t*)y, idx, H, W, Ho, Wo);
  TDN_LAUNCH_CHECK();
  return 0;
}
