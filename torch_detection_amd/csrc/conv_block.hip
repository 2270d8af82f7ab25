// One residual Bottleneck in ONE launch (gfx950): conv1 1x1 -> conv2 3x3 -> conv3 1x1 (+ residual), forward or the
// input-gradient chain, with the two intermediate activations handed from GEMM to GEMM through LDS.
//
// Replaces the per-conv launches of Bottleneck.forward, models/backbone/resnet.py:97-119
//   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + x)          (stride 1, no downsample: :110-118)
// and of its autograd input gradient
//   g2 = mask(h2) . conv3^T(g);  g1 = mask(h1) . conv2^T(g2);  dx = mask(x) . (conv1^T(g1) + g)
// (BN scales folded into the dgrad weight packs, as everywhere in this library).  Both passes have the same shape —
// a 4C -> C 1x1 GEMM on a haloed pixel patch, a C -> C 3x3 GEMM, a C -> 4C 1x1 GEMM plus the phase-1 input as the
// addend — so one kernel template serves both; only the epilogues differ (affine + ReLU / ReLU mask of a saved tensor)
// and the 3x3 tap order is mirrored.
//
// Why one launch: in layer1 / layer2 these three convs are HBM-bound and each of them re-reads what the previous one
// has just written (x twice, h1, h2): 137 MB per image and block at 200x336 against 86 MB when x is read once and
// h1 / h2 are only written (they are still needed by the weight gradients).  Why LDS and not flags between
// workgroups: a cross-CU hand-off on this chip goes through HBM-side coherence (per-XCD L2s are not coherent) and
// costs as much as the kernel boundary it replaces (MI355X_MICROARCH.md, "handoff-flag" 2-5 us vs "boundary" 1.5 us);
// inside one workgroup the hand-off is a ds_write + s_barrier.
//
// Workgroup = 8 x 16 output pixels (256 threads, two workgroups per CU, 80 KB of LDS each):
//   phase 1  H1[10x18 halo pixels][C]  = epi1(W1[C][4C] . A[halo][4C])      A streamed global -> LDS by LDS-DMA, K-steps of 64
//   phase 2  H2[8x16][C]               = epi2(sum_taps W2[C][tap][C] . H1[pixel + tap][C])   H1 read in place with tap shifts
//   phase 3  OUT[8x16][4C]             = epi3(W3[4C][C] . H2 + A[pixel])
// K order and arithmetic of every phase equal those of conv_gemm_kernel (channel chunks in order; taps in the order
// build_fwd / build_dgrad list them; fma(acc, scale, shift), + addend, ReLU / mask, one rounding to 16 bit), and the
// intermediates are rounded to 16 bit before they are consumed, exactly as when they travel through HBM: results are
// bit-identical to the three separate launches (tests/test_gpu_block.py).
//
// LDS images are [row][64 channels] with 128-byte rows and a 16-byte-chunk XOR swizzle f(row) = (row >> 1) & 7 applied
// to the SOURCE address of the LDS-DMA (or to the ds_write address) and to the fragment reads.  The 3x3 phase reads 16
// consecutive patch rows per fragment at an arbitrary row offset (the tap shift); MFMA column r is therefore dealt to
// patch pixel pi(r) — even pixels to the lanes that read with k-chunk kq, odd pixels to those that read with kq + 1 —
// which keeps every ds_read_b128 lane group on 16 distinct 16-byte slots for every shift (tile width 16 = fragment
// width: a fragment never straddles two patch rows, cf. the halo kernel's residual conflicts).
#include "common.h"
#include <string.h>

struct BlockParams {
  const bf16_t* a;     // [N][H][W][4C]   forward: x; backward: g
  const bf16_t* w1;    // [C][4C]         forward: conv1 w_fwd; backward: conv3 w_dgrad
  const bf16_t* w2;    // [C][9][C]       conv2 w_fwd / w_dgrad
  const bf16_t* w3;    // [4C][C]         forward: conv3 w_fwd; backward: conv1 w_dgrad
  const float* sc1; const float* sh1;   // forward: folded BN of conv1 / conv2 / conv3 (NULL: 1 / 0)
  const float* sc2; const float* sh2;
  const float* sc3; const float* sh3;
  const bf16_t* m1;    // backward: ReLU-mask sources of the three outputs ([..C] h2, [..C] h1, [..4C] x; NULL: none)
  const bf16_t* m2;
  const bf16_t* m3;
  bf16_t* o1;          // [N][H][W][C]    h1 / g2
  bf16_t* o2;          // [N][H][W][C]    h2 / g1
  bf16_t* o3;          // [N][H][W][4C]   out / dx
  int N, H, W;
  int tiles_x, tiles_y, ntiles, nwg_pad;
};

template <bool BWD, bool F16>
__global__ __launch_bounds__(256, 2) void bottleneck64_kernel(const BlockParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 64, C4 = 256, TH = 8, TW = 16, HWD = TW + 2, PH = (TH + 2) * HWD /* 180 */, PHP = 192;
  constexpr int ROWB = 128;
  constexpr int X_BYTES = PHP * ROWB;          // 24576: one 64-channel K-step of the haloed patch
  constexpr int W1_BYTES = C * ROWB;           //  8192
  constexpr int STAGE1 = X_BYTES + W1_BYTES;   // 32768; phase 1 ring: [0, 65536)
  constexpr int TAP_BYTES = C * ROWB;          //  8192
  constexpr int W2_LO = 24576;                 // taps 2..6: [24576, 65536)   (issued after phase 1)
  constexpr int W2_HI = 65536;                 // taps 0, 1 (issued at kernel start), later taps 7, 8: [65536, 81920)
  constexpr int W3_OFF = 24576;                // [24576, 57344): issued once taps 2..6 are consumed
  // H1 lives at [0, 24576) during phase 2, H2 at [0, 16384) during phase 3

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int lrow = lane >> 3, lchunk = lane & 7;

  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.nwg_pad >> 3) + (bid >> 3);   // XCD x owns a contiguous run of tiles (shared halos)
  if (tile >= p.ntiles) return;
  const int H = p.H, W = p.W;
  const int tpi = p.tiles_x * p.tiles_y;
  const int img = tile / tpi;
  const int trem = tile - img * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t img_pix0 = (int64_t)img * H * W;

  const char* zero = (const char*)g_zero_page + lchunk * 16;

  auto swz_w8 = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };    // 8 consecutive channels per lane
  auto swz_w16 = [](int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); };   // 16

  // ---- conv2 weights, taps 0 and 1: nothing else uses [65536, 81920) during phase 1 ----
  auto load_tap = [&](int t, int dst_off) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g8 = it * 4 + wave;
      const int n = g8 * 8 + lrow;
      const char* src = (const char*)p.w2 + ((int64_t)n * (9 * C) + t * C) * 2 + ((lchunk ^ swz_w8(n)) * 16);
      glds16_async(src, smem + dst_off + g8 * 8 * ROWB);
    }
  };
  load_tap(0, W2_HI);
  load_tap(1, W2_HI + TAP_BYTES);

  // ---- phase 1 loader state: 6 patch rows and 2 weight rows per lane and K-step ----
  const char* xsrc[6];
  unsigned xok = 0;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int R = (it * 4 + wave) * 8 + lrow;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    const int swz = (R >> 1) & 7;
    xsrc[it] = (const char*)p.a + ((img_pix0 + (int64_t)y * W + x) * C4 + ((lchunk ^ swz) * 8)) * 2;
    xok |= ok ? (1u << it) : 0u;
  }
  const char* w1src[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int n = (it * 4 + wave) * 8 + lrow;
    w1src[it] = (const char*)p.w1 + ((int64_t)n * C4) * 2 + ((lchunk ^ swz_w8(n)) * 16);
  }
  auto load_stage1 = [&](int kc) {
    char* sX = smem + (kc & 1) * STAGE1;
    char* sW = sX + X_BYTES;
#pragma unroll
    for (int it = 0; it < 6; ++it)
      glds16_async((xok >> it) & 1u ? xsrc[it] + kc * 128 : zero, sX + (it * 4 + wave) * 8 * ROWB);
#pragma unroll
    for (int it = 0; it < 2; ++it) glds16_async(w1src[it] + kc * 128, sW + (it * 4 + wave) * 8 * ROWB);
  };

  // ---- fragment read constants ----
  const int f_rd = (fr >> 1) & 7;                               // swizzle of a pixel row whose index is fr (mod 16)
  const int f_rd_w = ((fr & 3) >> 1) | ((fr >> 2) << 1);        // swizzle of this lane's weight rows
  const int wrow8 = (wn * 32 + (fr >> 2) * 8 + (fr & 3)) * ROWB;     // + 4i rows: 8 consecutive channels per lane
  // even / odd deal of the 3x3 phase: MFMA column r <-> pixel pi(r) of the 16-pixel tile row
  const int pi = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * (fr - 8) : 2 * (fr - 4) + 1);

  // ================= phase 1: H1[halo][C] =================
  f32x4_t acc1[2][6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  load_stage1(0);
#pragma unroll 1
  for (int kc = 0; kc < C4 / 64; ++kc) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // step kc landed; the other stage is free again
    if (kc + 1 < C4 / 64) load_stage1(kc + 1);
    const char* sX = smem + (kc & 1) * STAGE1 + (wm * 96 + fr) * ROWB;
    const char* sW = smem + (kc & 1) * STAGE1 + X_BYTES + wrow8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[6];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 6; ++j) xf[j] = lds_read_b128(sX + j * 16 * ROWB + (((kk * 4 + fq) ^ f_rd) * 16));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc1[i][j] = mfma16<F16>(wf[i], xf[j], acc1[i][j]);
    }
  }
  __builtin_amdgcn_s_barrier();   // b0: every wave is done reading the ring
  // conv2 taps 2..6 into [24576, 65536): lands while the epilogue below runs
#pragma unroll
  for (int t = 2; t < 7; ++t) load_tap(t, W2_LO + (t - 2) * TAP_BYTES);

  const int cb8 = wn * 32 + fq * 8;     // this lane's 8 consecutive channels of the C-wide outputs
  {
    f32x4_t sc[2], sh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sc[i] = p.sc1 ? *(const f32x4_t*)(p.sc1 + cb8 + 4 * i) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
      sh[i] = p.sh1 ? *(const f32x4_t*)(p.sh1 + cb8 + 4 * i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int R = wm * 96 + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc1[i][j] * sc[i] + sh[i];
      if constexpr (BWD) {
        if (p.m1 && ok) {
          const bf16x8_t mk = *(const bf16x8_t*)(p.m1 + pix * C + cb8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = (elem_to_f32<F16>(mk[e]) > 0.f) ? v[0][e] : 0.f;
            v[1][e] = (elem_to_f32<F16>(mk[4 + e]) > 0.f) ? v[1][e] : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f32_to_elem<F16>(ok ? v[0][e] : 0.f);       // zero padding of the 3x3 conv: outside the image H1 is 0
        o[4 + e] = f32_to_elem<F16>(ok ? v[1][e] : 0.f);
      }
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + R * ROWB + (((wn * 4 + fq) ^ ((R >> 1) & 7)) * 16)) = o;
      if (ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) *(bf16x8_t*)(p.o1 + pix * C + cb8) = o;
    }
  }
  // b1: H1 complete (LDS writes of every wave), taps 0 / 1 landed long ago (waited with phase 1's loads)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ================= phase 2: H2[8x16][C] =================
  f32x4_t acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto tap_compute = [&](int t, int slot_off) {
    const int ky = t / 3, kx = t - ky * 3;
    const int oy = BWD ? 2 - ky : ky, ox = BWD ? 2 - kx : kx;   // patch offset of the tap (build_fwd / build_dgrad order)
    const char* sW = smem + slot_off + wrow8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int R = (wm * 4 + j + oy) * HWD + pi + ox;
        xf[j] = lds_read_b128(smem + R * ROWB + (((kk * 4 + fq) ^ ((R >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = mfma16<F16>(wf[i], xf[j], acc2[i][j]);
    }
  };
  tap_compute(0, W2_HI);
  tap_compute(1, W2_HI + TAP_BYTES);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // b2: taps 2..6 landed; [65536, 81920) is free
  load_tap(7, W2_HI);
  load_tap(8, W2_HI + TAP_BYTES);
#pragma unroll
  for (int t = 2; t < 7; ++t) tap_compute(t, W2_LO + (t - 2) * TAP_BYTES);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // b3: taps 7, 8 landed; [24576, 65536) is free
  // conv3 weights [4C][C] into [24576, 57344) while taps 7 and 8 are multiplied
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int g8 = it * 4 + wave;
    const int n = g8 * 8 + lrow;
    glds16_async((const char*)p.w3 + (int64_t)n * C * 2 + ((lchunk ^ swz_w16(n)) * 16), smem + W3_OFF + g8 * 8 * ROWB);
  }
  tap_compute(7, W2_HI);
  tap_compute(8, W2_HI + TAP_BYTES);
  __builtin_amdgcn_s_barrier();   // b4: every wave is done with H1
  {
    f32x4_t sc[2], sh[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sc[i] = p.sc2 ? *(const f32x4_t*)(p.sc2 + cb8 + 4 * i) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
      sh[i] = p.sh2 ? *(const f32x4_t*)(p.sh2 + cb8 + 4 * i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = wm * 4 + j;
      const int y = y0 + yy, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc2[i][j] * sc[i] + sh[i];
      if constexpr (BWD) {
        if (p.m2 && ok) {
          const bf16x8_t mk = *(const bf16x8_t*)(p.m2 + pix * C + cb8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = (elem_to_f32<F16>(mk[e]) > 0.f) ? v[0][e] : 0.f;
            v[1][e] = (elem_to_f32<F16>(mk[4 + e]) > 0.f) ? v[1][e] : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = f32_to_elem<F16>(v[0][e]);
        o[4 + e] = f32_to_elem<F16>(v[1][e]);
      }
      const int pr = yy * TW + pi;
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + pr * ROWB + (((wn * 4 + fq) ^ ((pr >> 1) & 7)) * 16)) = o;
      if (ok) *(bf16x8_t*)(p.o2 + pix * C + cb8) = o;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // b5: H2 complete, conv3 weights landed

  // ================= phase 3: OUT[8x16][4C], two passes of 128 channels =================
  const int wrow16 = (wn * 64 + (fr >> 2) * 16 + (fr & 3)) * ROWB;   // + 4i rows: 16 consecutive channels per lane
#pragma unroll 1
  for (int nc = 0; nc < 2; ++nc) {
    const int ch0 = nc * 128 + wn * 64 + fq * 16;
    // addend (and mask) of this lane's outputs: requested before the MFMAs, consumed after them
    bf16x8_t ad[4][2], mk[4][2];
    bool okp[4];
    int64_t pixp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + wm * 4 + j, x = x0 + fr;
      okp[j] = (y < H) && (x < W);
      pixp[j] = img_pix0 + (int64_t)y * W + x;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ad[j][h] = okp[j] ? *(const bf16x8_t*)(p.a + pixp[j] * C4 + ch0 + h * 8) : bf16x8_t{};
        if constexpr (BWD) mk[j][h] = (okp[j] && p.m3) ? *(const bf16x8_t*)(p.m3 + pixp[j] * C4 + ch0 + h * 8) : bf16x8_t{};
      }
    }
    f32x4_t acc3[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* sW = smem + W3_OFF + nc * 128 * ROWB + wrow16;
    const char* sX = smem + (wm * 64 + fr) * ROWB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = lds_read_b128(sW + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = lds_read_b128(sX + j * 16 * ROWB + (((kk * 4 + fq) ^ f_rd) * 16));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = mfma16<F16>(wf[i], xf[j], acc3[i][j]);
    }
    f32x4_t sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sc[i] = p.sc3 ? *(const f32x4_t*)(p.sc3 + ch0 + 4 * i) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
      sh[i] = p.sh3 ? *(const f32x4_t*)(p.sh3 + ch0 + 4 * i) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!okp[j]) continue;
      f32x4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc3[i][j] * sc[i] + sh[i];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * h][e] += elem_to_f32<F16>(ad[j][h][e]);
          v[2 * h + 1][e] += elem_to_f32<F16>(ad[j][h][4 + e]);
        }
      if constexpr (BWD) {
        if (p.m3) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * h][e] = (elem_to_f32<F16>(mk[j][h][e]) > 0.f) ? v[2 * h][e] : 0.f;
              v[2 * h + 1][e] = (elem_to_f32<F16>(mk[j][h][4 + e]) > 0.f) ? v[2 * h + 1][e] : 0.f;
            }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = f32_to_elem<F16>(v[2 * h][e]);
          o[4 + e] = f32_to_elem<F16>(v[2 * h + 1][e]);
        }
        *(bf16x8_t*)(p.o3 + pixp[j] * C4 + ch0 + h * 8) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <bool BWD, bool F16>
static int launch_block64(BlockParams& p, hipStream_t stream) {
  constexpr int lds = 81920;
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)bottleneck64_kernel<BWD, F16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
    attr_set[dev] = true;
    if (getenv("TDN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)bottleneck64_kernel<BWD, F16>, 256, lds);
      fprintf(stderr, "[tdn] bottleneck64<%d,%d>: %d B LDS, %d workgroups/CU\n", (int)BWD, (int)F16, lds, nb);
    }
  }
  TDN_LAUNCH((bottleneck64_kernel<BWD, F16>), dim3(p.nwg_pad), dim3(256), lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_bottleneck_supported(int H, int W, int C, int stride, int dilation) {
  return (C == 64 && stride == 1 && dilation == 1 && H > 0 && W > 0) ? 1 : 0;
}

static int block_common(BlockParams& p, const tdn_bottleneck_args* a, int dtype) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(a != nullptr, "bottleneck: NULL argument block");
  TDN_CHECK(a->N > 0 && a->H > 0 && a->W > 0, "bottleneck: bad tensor shape N=%d H=%d W=%d", a->N, a->H, a->W);
  TDN_CHECK(tdn_bottleneck_supported(a->H, a->W, a->C, 1, 1), "bottleneck: C=%d is not built (64)", a->C);
  TDN_CHECK(a->in && a->w1 && a->w2 && a->w3 && a->out1 && a->out2 && a->out3, "bottleneck: NULL tensor pointer");
  TDN_CHECK((int64_t)a->N * a->H * a->W < (1ll << 31) / 4, "tensor too large for 32-bit pixel indexing");
  memset(&p, 0, sizeof(p));
  p.a = (const bf16_t*)a->in; p.w1 = (const bf16_t*)a->w1; p.w2 = (const bf16_t*)a->w2; p.w3 = (const bf16_t*)a->w3;
  p.o1 = (bf16_t*)a->out1; p.o2 = (bf16_t*)a->out2; p.o3 = (bf16_t*)a->out3;
  p.N = a->N; p.H = a->H; p.W = a->W;
  p.tiles_x = ceil_div(a->W, 16); p.tiles_y = ceil_div(a->H, 8);
  p.ntiles = a->N * p.tiles_x * p.tiles_y;
  p.nwg_pad = (p.ntiles + 7) & ~7;
  return 0;
}

extern "C" int tdn_bottleneck_fwd(const tdn_bottleneck_args* a, int dtype, void* stream) {
  BlockParams p;
  if (block_common(p, a, dtype)) return -1;
  p.sc1 = a->scale1; p.sh1 = a->shift1; p.sc2 = a->scale2; p.sh2 = a->shift2; p.sc3 = a->scale3; p.sh3 = a->shift3;
  if (dtype == TDN_F16) return launch_block64<false, true>(p, (hipStream_t)stream);
  return launch_block64<false, false>(p, (hipStream_t)stream);
}

extern "C" int tdn_bottleneck_dgrad(const tdn_bottleneck_args* a, int dtype, void* stream) {
  BlockParams p;
  if (block_common(p, a, dtype)) return -1;
  p.m1 = (const bf16_t*)a->mask1; p.m2 = (const bf16_t*)a->mask2; p.m3 = (const bf16_t*)a->mask3;
  if (dtype == TDN_F16) return launch_block64<true, true>(p, (hipStream_t)stream);
  return launch_block64<true, false>(p, (hipStream_t)stream);
}
