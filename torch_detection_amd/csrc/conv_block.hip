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
#include <type_traits>

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
  // ReLU bit planes (1 bit per element, bit c % 32 of word c / 32 of the pixel's run of words): written by the forward
  // pass when given, read by the backward pass INSTEAD of the 16-bit mask sources m1 / m2 / m3
  unsigned* b1;        // [N][H][W][C / 32]    h1 > 0   (backward: mask of its phase 2 output)
  unsigned* b2;        // [N][H][W][C / 32]    h2 > 0   (backward: mask of its phase 1 output, read on the halo too)
  unsigned* b3;        // [N][H][W][4C / 32]   x > 0    (backward: mask of its phase 3 output)
  // head variant (first block of layer1: 1x1 downsample on the residual branch, C input channels):
  const bf16_t* ad;    // forward: [N][H][W][4C] downsample branch (when wd is NULL); backward: [N][H][W][C] its input gradient
  const bf16_t* wd;    // forward: downsample w_fwd [4C][C]: the branch is computed in phase 3 (ad unused)
  const float* scd; const float* shd;
  int N, H, W;
  int tiles_x, tiles_y, ntiles, nwg_pad;
  unsigned long long* trace;   // libtdn_trace.so only: 16 x 8-byte stamps per workgroup (scripts/block_trace.py)
};

#ifdef TDN_TRACE_BUILD
#define BLK_STAMP(slot)                                                                                   \
  do {                                                                                                    \
    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter();   \
  } while (0)
#define BLK_STAMP_RT(slot)                                                                                \
  do {                                                                                                    \
    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define BLK_STAMP(slot) do { } while (0)
#define BLK_STAMP_RT(slot) do { } while (0)
#endif

// Store target of output pixels that lie outside the image (ragged right / bottom tiles): the epilogue's loads and
// stores are then unconditional — address selects, no branches — so the compiler can count what is in flight behind a
// load instead of waiting vmcnt(0), which would drain the stores issued before it.
static __device__ __attribute__((aligned(256))) unsigned char g_blk_sink[512];
static __device__ __attribute__((aligned(256))) unsigned char g_blk_zero[512];   // load source of such pixels (a lane reads up to 288 B behind its base)

// bit e of the result: element e of v is > 0 (what the ReLU masks of the backward pass test).  Both 16-bit float formats
// are sign-magnitude, so "> 0" is the signed 16-bit integer test: max(min(h, 1), 0) is 1 for a positive half-word and 0
// otherwise — two packed 16-bit instructions per pair of elements instead of convert + compare + select + or per
// element (the plane costs the forward launch ~10 % otherwise).  A positive NaN counts as positive here; a float
// compare would say no — ReLU outputs are never NaN.
template <bool F16>
__device__ __forceinline__ unsigned pos_bits8(bf16x8_t v) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4v_t __attribute__((ext_vector_type(4)));
  const u32x4v_t w = __builtin_bit_cast(u32x4v_t, v);
  unsigned f[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned wi = w[i];     // (a bit_cast of the vector ELEMENT expression itself miscompiles: clang 19, host and device)
    const s16x2_t one = {1, 1}, zero = {0, 0};
    s16x2_t h = __builtin_bit_cast(s16x2_t, wi);
    h = __builtin_elementwise_max(__builtin_elementwise_min(h, one), zero);
    f[i] = __builtin_bit_cast(unsigned, h);          // bit 0: element 2i, bit 16: element 2i + 1
  }
  const unsigned r = f[0] | (f[1] << 2) | (f[2] << 4) | (f[3] << 6);   // element 2i at bit 2i, element 2i + 1 at bit 16 + 2i
  return (r | (r >> 15)) & 0xffu;
}
// the four lanes fr, fr + 16, fr + 32, fr + 48 (fq = 0..3) each hold one byte of a pixel's 32-channel word: every lane
// gets the whole word
__device__ __forceinline__ unsigned gather_word4(unsigned byte, int fq) {
  unsigned w = byte << (8 * fq);
  w |= __shfl_xor(w, 16);
  w |= __shfl_xor(w, 32);
  return w;
}

// Workgroup barrier that is also a compiler barrier for memory operations and retires this wave's LDS reads first.
// The raw __builtin_amdgcn_s_barrier() orders nothing for the compiler: a ds_write into a region other waves were
// still reading before the barrier (H1 over ring slots 2 / 3, H2 over H1) may be hoisted above it — seen as a
// timing-dependent mismatch at 525 co-resident workgroups, never at small grids.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// HEAD (the stage's first block, resnet.py:130-136 with stride 1: layer1.0): the block input has C channels and the
// residual branch is a 1x1 conv + BN of it.  Forward: phase 1 runs over K = C (two K-steps), the addend of phase 3 is
// the downsample branch — read from p.ad (computed by a conv launch of its own), or computed here when p.wd is given.
// Backward: phases 1 / 2 as usual (g has 4C channels); phase 3 produces C channels, dx = conv1^T(g1) + p.ad where p.ad
// is the downsample conv's input gradient; no mask (the producer of x is the max pool).
// HEAD == 2 (forward only): the downsample branch is computed HERE — phase 3 becomes, per pass of 128 output
// channels, a GEMM of the block input's tile with the downsample weights (result affine-mapped and rounded to 16 bit:
// the value the separate launch would have stored) followed by the conv3 GEMM whose epilogue adds it.  LDS from b3 on:
//   [0, 16384)        X2: the tile's 128 pixels x C input channels, rows in H2's layout      (issued at b3)
//   [16384, 32768)    downsample rows 0..127 (b3); after the first pass's downsample GEMM: conv3 rows 128..255
//   [32768, 40960)    conv3 rows 0..63 (b3)          [57344, 65536)  conv3 rows 64..127 (b4)
//   [40960, 57344)    H2                             [65536, 81920)  downsample rows 128..255 (b4)
template <bool BWD, bool F16, bool MB = false, int HEAD = 0>
__global__ __launch_bounds__(256, 2) void bottleneck64_kernel(const BlockParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 64, C4 = 256, TH = 8, TW = 16, HWD = TW + 2, PH = (TH + 2) * HWD /* 180 */, PHP = 192;
  constexpr int CA = (HEAD && !BWD) ? C : C4;   // channels of p.a = K of phase 1
  constexpr int KS1 = CA / 32;
  constexpr bool DSK = HEAD == 2 && !BWD;       // downsample branch computed in phase 3
  // HEAD == 2, backward: the downsample conv's input gradient t = Wd^T . g is accumulated in phase 1 beside conv3's
  // (same operand stream g, a second weight chunk per K-step and a second accumulator set, over the haloed patch like
  // the first), rounded to 16 bit as its own launch would store it, parked in registers through phase 2, handed to
  // phase 3's pixel layout through a [128 pixels][C] LDS tile at [8192, 24576), and added there.  The ring then has
  // three slots of 20 KB (X 12 KB + two weight chunks of 4 KB); taps 2, 3 are issued at the last K-step, 4..6 behind b0.
  constexpr bool DSB = HEAD == 2 && BWD;
  constexpr int ROWB = 128;                    // H1 / H2 / conv2 / conv3 weight rows: 64 channels
  // ---- LDS map (80 KB; two workgroups per CU) ----
  // phase 1: ring of four 32-channel K-steps, S(s) = s * 16384: X[192 rows][64 B] (12 KB) + W1[64 rows][64 B] (4 KB)
  constexpr int SLOT = DSB ? 20480 : 16384, NSLOT = DSB ? 3 : 4, XB32 = PHP * 64;
  constexpr int T_OFF = 65536;                 // conv2 taps 0, 1 (issued at kernel start), later taps 7, 8
  constexpr int TAP_BYTES = C * ROWB;          // 8192
  // conv2 taps 2..6 at [0, 40960): issued into the ring slots as phase 1 releases them (they take the place of the
  // K-steps 8, 9, 10 the ring would prefetch next, so the counted vmcnt stays uniform)
  constexpr int H1_OFF = 40960;                // [40960, 65536): written once the ring is drained
  constexpr int W3_OFF = 0;                    // [0, 32768): issued once taps 2..6 are consumed
  constexpr int H2_OFF = 40960;                // [40960, 57344): written once H1 is dead

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.nwg_pad >> 3) + (bid >> 3);   // XCD x owns a contiguous run of tiles (shared halos)
  if (tile >= p.ntiles) return;
  BLK_STAMP(0);
  BLK_STAMP_RT(14);
  const int H = p.H, W = p.W;
  const int tpi = p.tiles_x * p.tiles_y;
  const int img = tile / tpi;
  const int trem = tile - img * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t img_pix0 = (int64_t)img * H * W;

  auto swz_w8 = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };    // 128-B rows, 8 channels per lane
  auto swz_w16 = [](int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); };   // 128-B rows, 16 channels per lane

  // ---- conv2 weight taps: 64 rows x 128 B each, two LDS-DMA instructions per wave ----
  const int lrow8 = lane >> 3, lchunk8 = lane & 7;
  auto tap_off = [](int t) { return t < 2 ? T_OFF + t * TAP_BYTES : (t < 7 ? (t - 2) * TAP_BYTES : T_OFF + (t - 7) * TAP_BYTES); };
  auto load_tap = [&](int t) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g8 = it * 4 + wave;
      const int n = g8 * 8 + lrow8;
      const char* src = (const char*)p.w2 + ((int64_t)n * (9 * C) + t * C) * 2 + ((lchunk8 ^ swz_w8(n)) * 16);
      glds16_async(src, smem + tap_off(t) + g8 * 8 * ROWB);
    }
  };
  load_tap(0);
  load_tap(1);

  // ---- phase 1 loader state: 32-channel K-steps, 64-byte rows, 16 rows per LDS-DMA instruction ----
  // 64-byte rows: four rows per 256-byte bank line; chunk swizzle f(row) = (4 - ((row >> 2) & 3)) & 3 keeps the
  // ds_read_b128 lane groups (rows {0-3, 12-15} with k-chunk kq, rows {4-11} with kq + 1) on 16 distinct slots
  const int lrow16 = lane >> 2, lchunk4 = lane & 3;
  const char* zero = (const char*)g_zero_page + lchunk4 * 16;
  const char* xsrc[3];
  unsigned xok = 0;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int R = (it * 4 + wave) * 16 + lrow16;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    const int swz = (4 - ((R >> 2) & 3)) & 3;
    xsrc[it] = (const char*)p.a + ((img_pix0 + (int64_t)y * W + x) * CA + ((lchunk4 ^ swz) * 8)) * 2;
    xok |= ok ? (1u << it) : 0u;
  }
  const char* w1src;
  {
    const int n = wave * 16 + lrow16;
    w1src = (const char*)p.w1 + ((int64_t)n * CA) * 2 + ((lchunk4 ^ ((4 - ((n >> 3) & 3)) & 3)) * 16);
  }
  const char* wdsrc = nullptr;
  if constexpr (DSB) {
    const int n = wave * 16 + lrow16;
    wdsrc = (const char*)p.wd + ((int64_t)n * CA) * 2 + ((lchunk4 ^ ((4 - ((n >> 3) & 3)) & 3)) * 16);
  }
  auto load_step = [&](int kc) {   // K-step kc (32 channels) into ring slot kc % NSLOT
    char* sX = smem + (kc % NSLOT) * SLOT;
#pragma unroll
    for (int it = 0; it < 3; ++it)
      glds16_async((xok >> it) & 1u ? xsrc[it] + kc * 64 : zero, sX + (it * 4 + wave) * 1024);
    glds16_async(w1src + kc * 64, sX + XB32 + wave * 1024);
    if constexpr (DSB) glds16_async(wdsrc + kc * 64, sX + XB32 + 4096 + wave * 1024);
  };

  // ---- fragment read constants ----
  const int f_rd = (fr >> 1) & 7;                               // 128-B rows: swizzle of a pixel row congruent to fr (mod 16)
  const int f_rd_w = ((fr & 3) >> 1) | ((fr >> 2) << 1);        // 128-B rows: swizzle of this lane's weight rows
  const int f_rd32 = (4 - (fr >> 2)) & 3;                       // 64-B rows: pixel rows and weight rows alike
  const int wrow8 = wn * 32 + (fr >> 2) * 8 + (fr & 3);         // + 4i: this lane's weight row of fragment i (8 ch / lane)
  // even / odd deal of the 3x3 phase: MFMA column r <-> pixel pi(r) of the 16-pixel tile row
  const int pi = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * (fr - 8) : 2 * (fr - 4) + 1);
  const int cb8 = wn * 32 + fq * 8;     // this lane's 8 consecutive channels of the C-wide outputs

  // pick element fq of a 4-entry register array (every lane holds all entries after the gathers): lets the four lanes
  // of a pixel column store four different fragments' words with ONE instruction instead of one lane storing four times
  auto sel4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3) { return fq == 0 ? a0 : (fq == 1 ? a1 : (fq == 2 ? a2 : a3)); };
  // per-channel affine of an epilogue: folded BN in the forward pass (NULL: 1 / 0); the backward pass has none — the
  // same fma(acc, 1, 0) as conv_gemm_kernel computes for a NULL scale / shift, with constant operands
  auto load_affine = [&](const float* scp, const float* shp, int ch, f32x4_t& sc, f32x4_t& sh) {
    sc = (!BWD && scp) ? *(const f32x4_t*)(scp + ch) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
    sh = (!BWD && shp) ? *(const f32x4_t*)(shp + ch) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  f32x4_t sc1v[2], sh1v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) load_affine(p.sc1, p.sh1, cb8 + 4 * i, sc1v[i], sh1v[i]);
  // backward: ReLU-mask operands of the phase 1 epilogue, requested before the K loop
  bf16x8_t mk1[MB ? 1 : 6];
  unsigned mw1[MB ? 6 : 1];      // MB: the pixel's 32-channel word of the h2 > 0 bit plane (this lane's byte: fq)
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int R = wm * 96 + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw1[j] = *(ok ? p.b2 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk1[j] = (ok && p.m1) ? *(const bf16x8_t*)(p.m1 + pix * C + cb8) : bf16x8_t{};
    }
  }

  // backward: mask operands of the phase 2 epilogue (requested behind b1; before the K loop when the ring is followed
  // by tap transfers whose counted waits they must not disturb: DSB)
  bf16x8_t mk2[MB ? 1 : 4];
  unsigned mw2[MB ? 4 : 1];
  auto load_mask2 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + wm * 4 + j, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw2[j] = *(ok ? p.b1 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk2[j] = (p.m2 && ok) ? *(const bf16x8_t*)(p.m2 + pix * C + cb8) : bf16x8_t{};
    }
  };
  if constexpr (DSB) load_mask2();

  // ================= phase 1: H1[halo][C], eight 32-deep K-steps through a four-slot ring =================
  f32x4_t acc1[2][6];
  f32x4_t acct[DSB ? 2 : 1][DSB ? 6 : 1];
  if constexpr (DSB) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acct[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // head, forward: the downsample branch (both passes' addends) is requested first of all — it returns with the first
  // K-step; anywhere later it would sit behind LDS-DMA batches whose counted waits it must not disturb
  bf16x8_t ad0[4][2], ad1[4][2];
  if constexpr (HEAD && !BWD && !DSK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + wm * 4 + j, x = x0 + fr;
      const bf16_t* src = ((y < H) && (x < W)) ? p.ad + (img_pix0 + (int64_t)y * W + x) * C4 + wn * 64 + fq * 16
                                              : (const bf16_t*)g_blk_zero;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ad0[j][h] = *(const bf16x8_t*)(src + h * 8);
        ad1[j][h] = *(const bf16x8_t*)(src + 128 + h * 8);
      }
    }
  }
  load_step(0);
  load_step(1);
  if constexpr (DSB) {}
  else if constexpr (KS1 == 8) load_step(2);
  else load_tap(6);       // [32768, 40960): the third ring slot, which two K-steps never use
#pragma unroll
  for (int kc = 0; kc < KS1; ++kc) {
    // every issue slot below is 4 LDS-DMA instructions per wave (a K-step, or two conv2 taps): two of them younger
    // than K-step kc are in flight here
    // lgkmcnt(0): with the loop unrolled the scheduler sinks the MFMAs — and the waits of their fragment reads — below
    // the barrier; a read still queued in the LDS pipe can then be overtaken by the LDS-DMA another wave issues into
    // the same ring slot right behind the barrier (seen: 1 KiB pieces of stale data in ~10 of 525 tiles per launch)
    if constexpr (DSB) {   // three slots: one K-step (5 instructions per wave) in flight behind K-step kc
      if (kc < 7) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if constexpr (KS1 == 8) {
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // K-step kc landed; slot (kc - 1) & 3 is free
    } else {      // two K-steps: behind K-step 0 are K-step 1 (4 instructions per wave) and tap 6 (2)
      if (kc == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (kc == 0) BLK_STAMP(1);
    if (kc == 4) BLK_STAMP(2);
    if constexpr (DSB) {
      if (kc + 2 < 8) load_step(kc + 2);
      else if (kc == 7) {
        // taps 2, 3 into slot 0 (read last at K-step 6).  The epilogues' mask operands are touched first: everything
        // issued so far has landed (vmcnt(0) above), so the compiler's own wait for them costs nothing here — behind
        // the tap transfers it would drain them
        if constexpr (MB) {
          asm volatile("" ::"v"(mw1[0]), "v"(mw1[MB ? 5 : 0]), "v"(mw2[0]), "v"(mw2[MB ? 3 : 0]));
        } else {
          asm volatile("" ::"v"(mk1[0]), "v"(mk1[MB ? 0 : 5]), "v"(mk2[0]), "v"(mk2[MB ? 0 : 3]));
        }
        load_tap(2); load_tap(3);
      }
    } else if constexpr (KS1 == 8) {
      if (kc + 3 < 8) load_step(kc + 3);
      else if (kc == 5) { load_tap(2); load_tap(3); }
      else if (kc == 6) { load_tap(4); load_tap(5); }
      else { load_tap(6); }
    }
    const char* sX = smem + (kc % NSLOT) * SLOT + (wm * 96 + fr) * 64 + ((fq ^ f_rd32) * 16);
    const char* sW = smem + (kc % NSLOT) * SLOT + XB32 + wrow8 * 64 + ((fq ^ f_rd32) * 16);
    bf16x8_t wf[2], xf[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * 64);
#pragma unroll
    for (int j = 0; j < 6; ++j) xf[j] = lds_read_b128(sX + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc1[i][j] = mfma16<F16>(wf[i], xf[j], acc1[i][j]);
    if constexpr (DSB) {
      bf16x8_t wt[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wt[i] = lds_read_b128(sW + 4096 + i * 4 * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acct[i][j] = mfma16<F16>(wt[i], xf[j], acct[i][j]);
    }
  }
  lds_barrier();   // b0: every wave is done reading the ring
  BLK_STAMP(3);
  bf16x8_t tpk[DSB ? 6 : 1];   // DSB: t of this lane's six patch pixels (its 8 channels), as the downsample dgrad launch stores it
  if constexpr (DSB) {
    load_tap(4); load_tap(5); load_tap(6);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tpk[j][e] = f32_to_elem<F16>(acct[0][j][e] * 1.f + 0.f);
        tpk[j][4 + e] = f32_to_elem<F16>(acct[1][j][e] * 1.f + 0.f);
      }
  }
  if constexpr (KS1 != 8) {
    // taps 2..5 go where the two K-steps were.  The epilogue's affine operands are touched first: the compiler waits
    // vmcnt(0) at the first use of a load it knows about, which must not fall behind these eight instructions
    asm volatile("" ::"v"(sc1v[0]), "v"(sc1v[1]), "v"(sh1v[0]), "v"(sh1v[1]));
    load_tap(2); load_tap(3); load_tap(4); load_tap(5);
  }

  bf16x8_t o1v[6];
  unsigned b1w[6];                // forward: the h1 > 0 word of the pixel (32 channels: this wave's wn)
  unsigned st1 = 0;               // bit j: fragment j of this lane is an interior pixel inside the image -> stored
  {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int R = wm * 96 + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc1[i][j] * sc1v[i] + sh1v[i];
      if constexpr (BWD) {
        if (MB || p.m1) {
          const unsigned m = MB ? (mw1[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk1[MB ? 0 : j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
            v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + H1_OFF + R * ROWB + (((wn * 4 + fq) ^ ((R >> 1) & 7)) * 16)) = o;
      o1v[j] = o;
      st1 |= (ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) ? (1u << j) : 0u;
      if constexpr (!BWD) {
        if (p.b1) b1w[j] = gather_word4(pos_bits8<F16>(o), fq);   // wave-uniform branch: every lane shuffles
      }
    }
  }
  // b1: H1 complete (LDS writes of every wave), taps 2..6 landed (taps 0 / 1 long ago); two K-steps: taps 2..5 (eight
  // instructions per wave) are still travelling and are waited for at b2
  if constexpr (DSB) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // taps 2..6 travel on
  else if constexpr (KS1 == 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  BLK_STAMP(4);
  // h1 / g2 to HBM: behind the barrier, so that nobody waits for the stores (DSB: behind b2, which drains the queue)
  auto store_o1 = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if ((st1 >> j) & 1u) {
        const int R = wm * 96 + j * 16 + fr;
        const int hy = R / HWD, hx = R - hy * HWD;
        const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
        *(bf16x8_t*)(p.o1 + pix * C + cb8) = o1v[j];
      }
    }
  };
  if constexpr (!DSB) store_o1();
  if constexpr (!BWD) {
    if (p.b1) {   // h1 > 0 words: lane fq stores fragment fq's (then fragment 4 + fq's) word of its pixel column
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int j = r * 4 + fq;
        const unsigned w = r == 0 ? sel4(b1w[0], b1w[1], b1w[2], b1w[3]) : sel4(b1w[4], b1w[5], 0u, 0u);
        const int R = wm * 96 + j * 16 + fr;
        const int hy = R / HWD, hx = R - hy * HWD;
        const bool st = j < 6 && ((st1 >> j) & 1u);
        const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
        *(st ? p.b1 + pix * (C / 32) + wn : (unsigned*)g_blk_sink) = w;
      }
    }
  }
  if constexpr (BWD && !DSB) load_mask2();

  // Phase 3's per-pixel operands are requested HERE, a whole phase ahead: loads return in issue order, so behind a
  // batch of LDS-DMA they would not be back before it has landed, and their first use sits behind b3 / b5, where the
  // queue is drained anyway.  Forward: both passes' addends; backward (which also carries mask operands): the first
  // pass's addend and mask, the second pass's after the first pass.
  const int chw = wn * 64 + fq * 16;       // + nc * 128: this lane's 16 consecutive output channels
  // Loads and stores of output pixels outside the image (ragged tiles) are redirected to the zero page / a sink line
  // instead of being branched around: see g_blk_sink.
  // Only the pixel index is kept per output pixel (-1: outside the image); operand / result addresses are formed where
  // they are used (one 64-bit multiply-add each).  Invalid pixels read the zero page and write a sink line.
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  int pixj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int y = y0 + wm * 4 + j, x = x0 + fr;
    pixj[j] = ((y < H) && (x < W)) ? (int)(img_pix0 + (int64_t)y * W + x) : -1;
  }
  auto adp = [&](int j) { return pixj[j] >= 0 ? (HEAD ? p.ad : p.a) + (int64_t)pixj[j] * C4 + chw : (const bf16_t*)g_blk_zero; };
  auto mkp = [&](int j) { return (pixj[j] >= 0 && p.m3 && !MB) ? p.m3 + (int64_t)pixj[j] * C4 + chw : (const bf16_t*)g_blk_zero; };
  auto o3p = [&](int j) { return pixj[j] >= 0 ? p.o3 + (int64_t)pixj[j] * C4 + chw : (bf16_t*)g_blk_sink; };
  // MB: this lane's pair of words (64 channels: wn) of the x > 0 plane, + nc * (C / 16); forward: lane fq == 0 stores it
  auto b3r = [&](int j) { return (pixj[j] >= 0 && MB) ? p.b3 + (int64_t)pixj[j] * (C4 / 32) + wn * 2 : (const unsigned*)g_blk_zero; };
  bf16x8_t mk3[MB ? 1 : 4][2];
  u32x2_t mw3[MB ? 2 : 1][4];    // MB: both passes' word pairs
  auto load_ad = [&](int nc, bf16x8_t (&dst)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) dst[j][h] = *(const bf16x8_t*)(adp(j) + nc * 128 + h * 8);
  };
  auto load_mask3 = [&](int nc) {
    if constexpr (!MB) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) mk3[j][h] = *(const bf16x8_t*)(mkp(j) + nc * 128 + h * 8);
    }
  };
  bf16x8_t adh[(BWD && HEAD) ? 4 : 1];   // head, backward: this lane's 8 channels of the downsample branch's input gradient
  if constexpr (BWD && HEAD) {
    if constexpr (!DSB) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        adh[j] = *(const bf16x8_t*)(pixj[j] >= 0 ? p.ad + (int64_t)pixj[j] * C + cb8 : (const bf16_t*)g_blk_zero);
    }
  } else if constexpr (!HEAD) {
    load_ad(0, ad0);
    if constexpr (BWD && !MB) load_mask3(0);
    else load_ad(1, ad1);      // forward, and backward with bit planes (8 B of mask per pixel instead of 32 B per pass)
    if constexpr (MB) {
#pragma unroll
      for (int nc = 0; nc < 2; ++nc)
#pragma unroll
        for (int j = 0; j < 4; ++j) mw3[nc][j] = *(const u32x2_t*)(b3r(j) + nc * (C / 16));
    }
  }

  // ================= phase 2: H2[8x16][C] =================
  f32x4_t acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto tap_compute = [&](int t) {
    const int ky = t / 3, kx = t - ky * 3;
    const int oy = BWD ? 2 - ky : ky, ox = BWD ? 2 - kx : kx;   // patch offset of the tap (build_fwd / build_dgrad order)
    const char* sW = smem + tap_off(t) + wrow8 * ROWB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int R = (wm * 4 + j + oy) * HWD + pi + ox;
        xf[j] = lds_read_b128(smem + H1_OFF + R * ROWB + (((kk * 4 + fq) ^ ((R >> 1) & 7)) * 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = mfma16<F16>(wf[i], xf[j], acc2[i][j]);
    }
  };
  tap_compute(0);
  tap_compute(1);
  if constexpr (KS1 == 8 && !DSB) lds_barrier();   // b2: [65536, 81920) is free
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... and taps 2..5 (DSB: 2..6) have landed
  BLK_STAMP(5);
  if constexpr (DSB) store_o1();
  f32x4_t sc2v[2], sh2v[2];
  if constexpr (DSK) {   // requested a barrier early and touched right behind b3's vmcnt(0): see below
#pragma unroll
    for (int i = 0; i < 2; ++i) load_affine(p.sc2, p.sh2, cb8 + 4 * i, sc2v[i], sh2v[i]);
  }
  load_tap(7);
  load_tap(8);
#pragma unroll
  for (int t = 2; t < 7; ++t) tap_compute(t);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // b3: taps 7, 8 landed; [0, 40960) is free
  BLK_STAMP(6);
  // (the compiler waits vmcnt(0) where a load it knows about is first used; with the operands of epilogue 2 touched
  // here that wait coincides with b3's instead of draining the transfers issued at b3 / b4)
  if constexpr (DSK) asm volatile("" ::"v"(sc2v[0]), "v"(sc2v[1]), "v"(sh2v[0]), "v"(sh2v[1]));
  // conv3 weights [4C][C] into [0, 32768) while taps 7 and 8 are multiplied
  // rows n0 .. n0 + 8 * ngroups of a [rows][C] weight matrix (16 channels per lane: swz_w16) to smem + base
  auto load_wrows = [&](const bf16_t* w, int n0, int ngroups, int base) {
#pragma unroll
    for (int g8 = wave; g8 < ngroups; g8 += 4) {
      const int n = n0 + g8 * 8 + lrow8;
      glds16_async((const char*)w + (int64_t)n * C * 2 + ((lchunk8 ^ swz_w16(n)) * 16), smem + base + g8 * 8 * ROWB);
    }
  };
  constexpr int TD_OFF = 8192;
  if constexpr (DSB) {   // t of the tile's interior pixels to [TD_OFF, TD_OFF + 16384), rows in H2's layout
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if ((st1 >> j) & 1u) {
        const int R = wm * 96 + j * 16 + fr;
        const int hy = R / HWD, hx = R - hy * HWD;
        const int pr = (hy - 1) * TW + (hx - 1);
        *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + TD_OFF + pr * ROWB + (((wn * 4 + fq) ^ ((pr >> 1) & 7)) * 16)) = tpk[j];
      }
    }
  }
  if constexpr (BWD && HEAD) {   // [C][C]: one tap's worth, in the taps' layout
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g8 = it * 4 + wave;
      const int n = g8 * 8 + lrow8;
      glds16_async((const char*)p.w3 + (int64_t)n * C * 2 + ((lchunk8 ^ swz_w8(n)) * 16), smem + W3_OFF + g8 * 8 * ROWB);
    }
  } else if constexpr (DSK) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {      // X2: pixel row pr = 16 * tile row + tile column, chunk swizzle (pr >> 1) & 7
      const int g8 = it * 4 + wave;
      const int pr = g8 * 8 + lrow8;
      const int y = y0 + (pr >> 4), x = x0 + (pr & 15);
      const char* src = ((y < H) && (x < W))
                            ? (const char*)p.a + ((img_pix0 + (int64_t)y * W + x) * C + ((lchunk8 ^ ((pr >> 1) & 7)) * 8)) * 2
                            : (const char*)g_zero_page + lchunk8 * 16;
      glds16_async(src, smem + g8 * 8 * ROWB);
    }
    load_wrows(p.wd, 0, 16, 16384);
    load_wrows(p.w3, 0, 8, 32768);
  } else {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int g8 = it * 4 + wave;
      const int n = g8 * 8 + lrow8;
      glds16_async((const char*)p.w3 + (int64_t)n * C * 2 + ((lchunk8 ^ swz_w16(n)) * 16), smem + W3_OFF + g8 * 8 * ROWB);
    }
  }
  if constexpr (!DSK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) load_affine(p.sc2, p.sh2, cb8 + 4 * i, sc2v[i], sh2v[i]);
  }
  tap_compute(7);
  tap_compute(8);
  lds_barrier();   // b4: every wave is done with H1
  BLK_STAMP(7);
  if constexpr (DSK) {
    load_wrows(p.w3, 64, 8, 57344);
    load_wrows(p.wd, 128, 16, 65536);
  }
  bf16x8_t o2v[4];
  unsigned b2w[4];
  f32x4_t sc3v[4], sh3v[4];
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc2[i][j] * sc2v[i] + sh2v[i];
      if constexpr (BWD) {
        if (MB || p.m2) {
          const unsigned m = MB ? (mw2[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk2[MB ? 0 : j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
            v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      const int pr = (wm * 4 + j) * TW + pi;
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + H2_OFF + pr * ROWB + (((wn * 4 + fq) ^ ((pr >> 1) & 7)) * 16)) = o;
      o2v[j] = o;
      if constexpr (!BWD) {
        if (p.b2) b2w[j] = gather_word4(pos_bits8<F16>(o), fq);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) load_affine(p.sc3, p.sh3, chw + 4 * i, sc3v[i], sh3v[i]);
  f32x4_t scdv[DSK ? 4 : 1], shdv[DSK ? 4 : 1];
  if constexpr (DSK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_affine(p.scd, p.shd, chw + 4 * i, scdv[i], shdv[i]);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // b5: H2 complete, conv3 weights landed

  BLK_STAMP(8);
  auto store_o2 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + wm * 4 + j, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      *(bf16x8_t*)(ok ? p.o2 + pix * C + cb8 : (bf16_t*)g_blk_sink) = o2v[j];
    }
    if constexpr (!BWD) {
      if (p.b2) {   // h2 > 0 words: lane fq stores tile row wm * 4 + fq's word of its pixel column
        const int y = y0 + wm * 4 + fq, x = x0 + pi;
        *((y < H && x < W) ? p.b2 + (img_pix0 + (int64_t)y * W + x) * (C / 32) + wn : (unsigned*)g_blk_sink) =
            sel4(b2w[0], b2w[1], b2w[2], b2w[3]);
      }
    }
  };
  // downsample branch computed here: no load is in flight behind b5, so h2 leaves at once (and frees its registers)
  if constexpr (DSK) store_o2();
  // ================= phase 3: OUT[8x16][4C], two passes of 128 channels =================
  const int wrow16 = (wn * 64 + (fr >> 2) * 16 + (fr & 3)) * ROWB;   // + 4i rows: 16 consecutive channels per lane
  if constexpr (BWD && HEAD) {
    // dx[8x16][C] = W1d[C][C] . G1 + (downsample branch's input gradient): one pass in the shape of a conv2 tap
    if constexpr (DSB) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pr = (wm * 4 + j) * TW + fr;
        adh[j] = lds_read_b128(smem + TD_OFF + pr * ROWB + (((wn * 4 + fq) ^ ((pr >> 1) & 7)) * 16));
      }
    }
    f32x4_t acc3[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* sW = smem + W3_OFF + wrow8 * ROWB;
    const char* sX = smem + H2_OFF + (wm * 64 + fr) * ROWB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = lds_read_b128(sX + j * 16 * ROWB + (((kk * 4 + fq) ^ f_rd) * 16));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = mfma16<F16>(wf[i], xf[j], acc3[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // g1 to HBM
      const int y = y0 + wm * 4 + j, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      *(bf16x8_t*)(ok ? p.o2 + pix * C + cb8 : (bf16_t*)g_blk_sink) = o2v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // fma(acc, 1, 0) + addend: the arithmetic of conv_gemm_kernel's epilogue without scale / shift
        o[e] = f32_to_elem<F16>((acc3[0][j][e] * 1.f + 0.f) + elem_to_f32<F16>(adh[j][e]));
        o[4 + e] = f32_to_elem<F16>((acc3[1][j][e] * 1.f + 0.f) + elem_to_f32<F16>(adh[j][4 + e]));
      }
      *(bf16x8_t*)(pixj[j] >= 0 ? p.o3 + (int64_t)pixj[j] * C + cb8 : (bf16_t*)g_blk_sink) = o;
    }
    BLK_STAMP(9);
    BLK_STAMP(10);
  } else {
#pragma unroll
    for (int nc = 0; nc < 2; ++nc) {
      __builtin_amdgcn_sched_barrier(0);   // keep the passes apart: hoisting the second one's loads and MFMAs spills
      bf16x8_t rd[DSK ? 4 : 1][2];     // the downsample branch of this pass, as the separate launch would have stored it
      if constexpr (DSK) {
        f32x4_t accd[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) accd[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const char* sWd = smem + (nc == 0 ? 16384 : 65536) + wrow16;
        const char* sXd = smem + (wm * 64 + fr) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8_t wf[4], xf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = lds_read_b128(sWd + i * 4 * ROWB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
          for (int j = 0; j < 4; ++j) xf[j] = lds_read_b128(sXd + j * 16 * ROWB + (((kk * 4 + fq) ^ f_rd) * 16));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) accd[i][j] = mfma16<F16>(wf[i], xf[j], accd[i][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              rd[j][h][e] = f32_to_elem<F16>((accd[2 * h][j] * scdv[2 * h] + shdv[2 * h])[e]);
              rd[j][h][4 + e] = f32_to_elem<F16>((accd[2 * h + 1][j] * scdv[2 * h + 1] + shdv[2 * h + 1])[e]);
            }
        if (nc == 0) {
          lds_barrier();                       // b6: the first pass's downsample rows are consumed
          load_wrows(p.w3, 128, 16, 16384);    // conv3 rows 128..255 take their place
        } else {
          // b7: those rows have landed — behind them this wave has issued the first pass's 8 stores
          asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
      f32x4_t acc3[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const char* sW = !DSK ? smem + W3_OFF + nc * 128 * ROWB + wrow16
                            : (nc == 0 ? smem + (wn ? 57344 : 32768) + ((fr >> 2) * 16 + (fr & 3)) * ROWB : smem + 16384 + wrow16);
      const char* sX = smem + H2_OFF + (wm * 64 + fr) * ROWB;
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
      bf16x8_t ov[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc3[i][j] * sc3v[i] + sh3v[i];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * h][e] += elem_to_f32<F16>((DSK ? rd[DSK ? j : 0] : (nc == 0 ? ad0 : ad1)[j])[h][e]);
            v[2 * h + 1][e] += elem_to_f32<F16>((DSK ? rd[DSK ? j : 0] : (nc == 0 ? ad0 : ad1)[j])[h][4 + e]);
          }
        if constexpr (BWD) {
          // this lane's 16 mask bits of the pixel (bit 8h + e: element e of half h); no mask at all: all ones
          unsigned m;
          if constexpr (MB) m = ((fq & 2) ? mw3[MB ? nc : 0][j][1] : mw3[MB ? nc : 0][j][0]) >> (16 * (fq & 1));
          else m = p.m3 ? (pos_bits8<F16>(mk3[MB ? 0 : j][0]) | (pos_bits8<F16>(mk3[MB ? 0 : j][1]) << 8)) : 0xffffu;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * h][e] = ((m >> (8 * h + e)) & 1u) ? v[2 * h][e] : 0.f;
              v[2 * h + 1][e] = ((m >> (8 * h + 4 + e)) & 1u) ? v[2 * h + 1][e] : 0.f;
            }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ov[j][h][e] = f32_to_elem<F16>(v[2 * h][e]);
            ov[j][h][4 + e] = f32_to_elem<F16>(v[2 * h + 1][e]);
          }
      }
      BLK_STAMP(9 + nc);
      __builtin_amdgcn_sched_barrier(0);
      if (nc == 0) {   // the second pass's operands travel while the first pass's results are stored
        if constexpr (BWD && !MB) { load_ad(1, ad1); load_mask3(1); }
        int ch1 = 128 + chw;
        // (16 loads = 64 registers when the downsample branch is computed here.  The opaque definition keeps them from
        // being hoisted into the GEMM above, the opaque uses keep the epilogue from being sunk below them: either way
        // they would be live together with the accumulators, and spill)
        if constexpr (DSK) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(ov[j][h]));
          asm volatile("" : "+v"(ch1));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) load_affine(p.sc3, p.sh3, ch1 + 4 * i, sc3v[i], sh3v[i]);
        if constexpr (DSK) {
#pragma unroll
          for (int i = 0; i < 4; ++i) load_affine(p.scd, p.shd, ch1 + 4 * i, scdv[i], shdv[i]);
        }
        if constexpr (!DSK) store_o2();   // h2 / g1 to HBM: behind every load a later wait counts
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) *(bf16x8_t*)(o3p(j) + nc * 128 + h * 8) = ov[j][h];
      if constexpr (!BWD && !HEAD) {
        if (p.b3) {   // x > 0: the pixel's 64-channel word pair from the four lanes' 16 bits each; lane fq stores row fq's
          const bf16x8_t (&ad)[4][2] = nc == 0 ? ad0 : ad1;
          unsigned lo[4], hi[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned v16 = (pos_bits8<F16>(ad[j][0]) | (pos_bits8<F16>(ad[j][1]) << 8)) << (16 * (fq & 1));
            v16 |= __shfl_xor(v16, 16);
            const unsigned other = __shfl_xor(v16, 32);
            lo[j] = (fq & 2) ? other : v16;
            hi[j] = (fq & 2) ? v16 : other;
          }
          const int y = y0 + wm * 4 + fq, x = x0 + fr;
          unsigned* dst = (y < H && x < W) ? p.b3 + (img_pix0 + (int64_t)y * W + x) * (C4 / 32) + wn * 2 + nc * (C / 16)
                                           : (unsigned*)g_blk_sink;
          *(u32x2_t*)dst = (u32x2_t){sel4(lo[0], lo[1], lo[2], lo[3]), sel4(hi[0], hi[1], hi[2], hi[3])};
        }
      }
    }
  }
#ifdef TDN_TRACE_BUILD
  BLK_STAMP(11);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BLK_STAMP(12);
  BLK_STAMP_RT(15);
  if (p.trace && tid == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.trace[(size_t)blockIdx.x * 16 + 13] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// C = 128 (layer2): the same three phases with 8 waves (2 pixel halves x 4 channel quarters, the per-wave tiles of
// the C = 64 kernel), one workgroup per CU and all 160 KB of LDS:
//   [0, 49152)        H1[192 rows][256 B]   (later H2[128 rows][256 B])
//   [49152, 163840)   seven 16 KB slots.  Phase 1 lays its four 20 KB K-step slots (X 12 KB + W1 8 KB per 32 channels)
//                     over the first five; everything else is a stream of 26 weight UNITS of 128 rows x 64 K-values
//                     (16 KB, two LDS-DMA instructions per wave) through the seven slots, unit u in slot (u + 5) % 7:
//                       u = 0..17   conv2, half h = u / 9 of the input channels, tap t = u % 9
//                                   (channel chunk outer, taps inner: the K order of conv_gemm_kernel / conv_halo_kernel)
//                       u = 18..25  conv3, pass nc = (u - 18) / 4, row half (u - 18) / 2 % 2, K half (u - 18) % 2
//                     Units 0, 1 are issued at kernel start, 2..4 into the K-step slots phase 1 releases, then one unit
//                     per consumed unit: six units stay in flight (counted vmcnt).
// 256-byte rows: chunk swizzle f(R) = 2 * ((R >> 1) & 7); with the even / odd pixel deal (also used by phase 3 here)
// the eight lanes of a ds_read_b128 lane group that share a k-chunk see eight consecutive values of R >> 1, and the
// two k-chunks of a group differ in bit 0, which f leaves alone.
// ---------------------------------------------------------------------------------------------
template <bool BWD, bool F16, bool MB = false>
__global__ __launch_bounds__(512, 1) void bottleneck128_kernel(const BlockParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 128, C4 = 512, TH = 8, TW = 16, HWD = TW + 2, PH = (TH + 2) * HWD /* 180 */, PHP = 192;
  constexpr int RB = 256;                      // H1 / H2 row bytes
  constexpr int WRB = 128;                     // weight unit row bytes
  constexpr int RING = 49152, UB = 16384;      // unit slot s: RING + s * UB
  constexpr int P1SLOT = 20480, XB32 = PHP * 64;   // phase 1 K-step slot s: RING + s * P1SLOT
  constexpr int KS1 = C4 / 32;                 // 16 K-steps of 32 channels

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.nwg_pad >> 3) + (bid >> 3);
  if (tile >= p.ntiles) return;
  BLK_STAMP(0);
  BLK_STAMP_RT(14);
  const int H = p.H, W = p.W;
  const int tpi = p.tiles_x * p.tiles_y;
  const int img = tile / tpi;
  const int trem = tile - img * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t img_pix0 = (int64_t)img * H * W;

  auto swz_w8 = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  auto swz_w16 = [](int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); };
  auto f256 = [](int R) { return ((R >> 1) & 7) << 1; };

  // ---- weight units ----
  const int lrow8 = lane >> 3, lchunk8 = lane & 7;
  auto unit_off = [](int u) { return RING + ((u + 5) % 7) * UB; };
  auto load_unit = [&](int u) {
    char* dst = smem + unit_off(u);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g8 = it * 8 + wave;
      const int r = g8 * 8 + lrow8;                  // row of the unit
      const char* src;
      if (u < 18) {
        const int h = u / 9, t = u - h * 9;
        src = (const char*)p.w2 + ((int64_t)r * (9 * C) + t * C + h * 64) * 2 + ((lchunk8 ^ swz_w8(r)) * 16);
      } else {
        const int v = u - 18, nc = v >> 2, ru = (v >> 1) & 1, h = v & 1;
        src = (const char*)p.w3 + ((int64_t)(nc * 256 + ru * 128 + r) * C + h * 64) * 2 + ((lchunk8 ^ swz_w16(r)) * 16);
      }
      glds16_async(src, dst + g8 * 8 * WRB);
    }
  };
  load_unit(0);
  load_unit(1);

  // ---- phase 1 loader: per K-step 12 X pieces (16 rows x 64 B) + 8 W1 pieces over 8 waves: every wave issues three
  // LDS-DMA instructions (waves 4..7 have no second X piece: a dummy copy of the zero page into the idle H1 region
  // keeps the per-wave vmcnt arithmetic uniform) ----
  const int lrow16 = lane >> 2, lchunk4 = lane & 3;
  const char* zero = (const char*)g_zero_page + lchunk4 * 16;
  const char* xsrc[2];
  unsigned xok = 0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int R = (it * 8 + wave) * 16 + lrow16;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    const int swz = (4 - ((R >> 2) & 3)) & 3;
    xsrc[it] = (const char*)p.a + ((img_pix0 + (int64_t)y * W + x) * C4 + ((lchunk4 ^ swz) * 8)) * 2;
    xok |= ok ? (1u << it) : 0u;
  }
  const char* w1src;
  {
    const int n = wave * 16 + lrow16;
    w1src = (const char*)p.w1 + ((int64_t)n * C4) * 2 + ((lchunk4 ^ ((4 - ((n >> 3) & 3)) & 3)) * 16);
  }
  auto load_step = [&](int kc) {
    char* sX = smem + RING + (kc & 3) * P1SLOT;
    glds16_async(xok & 1u ? xsrc[0] + kc * 64 : zero, sX + wave * 1024);
    if (wave < 4) glds16_async(xok & 2u ? xsrc[1] + kc * 64 : zero, sX + (8 + wave) * 1024);
    else glds16_async(zero, smem + wave * 1024);            // dummy, into the idle H1 region
    glds16_async(w1src + kc * 64, sX + XB32 + wave * 1024);
  };

  const int f_rd_w = ((fr & 3) >> 1) | ((fr >> 2) << 1);
  const int f_rd32 = (4 - (fr >> 2)) & 3;
  const int wrow8 = wn * 32 + (fr >> 2) * 8 + (fr & 3);
  const int pi = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * (fr - 8) : 2 * (fr - 4) + 1);
  const int cb8 = wn * 32 + fq * 8;

  // pick element fq of a 4-entry register array (every lane holds all entries after the gathers): lets the four lanes
  // of a pixel column store four different fragments' words with ONE instruction instead of one lane storing four times
  auto sel4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3) { return fq == 0 ? a0 : (fq == 1 ? a1 : (fq == 2 ? a2 : a3)); };
  auto load_affine = [&](const float* scp, const float* shp, int ch, f32x4_t& sc, f32x4_t& sh) {
    sc = (!BWD && scp) ? *(const f32x4_t*)(scp + ch) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
    sh = (!BWD && shp) ? *(const f32x4_t*)(shp + ch) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  f32x4_t sc1v[2], sh1v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) load_affine(p.sc1, p.sh1, cb8 + 4 * i, sc1v[i], sh1v[i]);
  bf16x8_t mk1[MB ? 1 : 6];
  unsigned mw1[MB ? 6 : 1];      // MB: the pixel's 32-channel word of the h2 > 0 bit plane (this lane's byte: fq)
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int R = wm * 96 + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw1[j] = *(ok ? p.b2 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk1[j] = *(const bf16x8_t*)((ok && p.m1) ? p.m1 + pix * C + cb8 : (const bf16_t*)g_blk_zero);
    }
  }

  // ================= phase 1 =================
  f32x4_t acc1[2][6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  load_step(0);
  load_step(1);
  load_step(2);
#pragma unroll
  for (int kc = 0; kc < KS1; ++kc) {
    // younger than K-step kc at this point: two K-steps (3 LDS-DMA each), or the units that take their place at the tail
    if (kc <= KS1 - 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (kc == KS1 - 2) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kc == 0) BLK_STAMP(1);
    if (kc == 8) BLK_STAMP(2);
    if (kc + 3 < KS1) load_step(kc + 3);
    else load_unit(kc + 3 - KS1 + 2);            // kc = 13, 14, 15 -> units 2, 3, 4 into the slots just released
    const char* sX = smem + RING + (kc & 3) * P1SLOT + (wm * 96 + fr) * 64 + ((fq ^ f_rd32) * 16);
    const char* sW = smem + RING + (kc & 3) * P1SLOT + XB32 + wrow8 * 64 + ((fq ^ f_rd32) * 16);
    bf16x8_t wf[2], xf[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * 64);
#pragma unroll
    for (int j = 0; j < 6; ++j) xf[j] = lds_read_b128(sX + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc1[i][j] = mfma16<F16>(wf[i], xf[j], acc1[i][j]);
  }
  lds_barrier();   // b0: the K-step slots are free
  BLK_STAMP(3);
  // Compiler-visible loads / stores share the vmcnt queue with the LDS-DMA stream.  They are placed where the counted
  // waits of the unit stream only ever find them OLDER than the units they may leave in flight (a count that is too
  // small over-waits; one that is too large would let a unit be read before it has landed): the phase 2 mask loads and
  // the h1 / g2 stores go in front of units 5 and 6.
  bf16x8_t mk2[MB ? 1 : 4];
  unsigned mw2[MB ? 4 : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = y0 + wm * 4 + j, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw2[j] = *(ok ? p.b1 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk2[j] = *(const bf16x8_t*)((p.m2 && ok) ? p.m2 + pix * C + cb8 : (const bf16_t*)g_blk_zero);
    }
  }

  bf16x8_t o1v[6];
  unsigned b1w[6];
  unsigned st1 = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int R = wm * 96 + j * 16 + fr;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    f32x4_t v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = acc1[i][j] * sc1v[i] + sh1v[i];
    if constexpr (BWD) {
      if (MB || p.m1) {
        const unsigned m = MB ? (mw1[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk1[MB ? 0 : j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
          v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      o[e] = f32_to_elem<F16>(ok ? v[0][e] : 0.f);
      o[4 + e] = f32_to_elem<F16>(ok ? v[1][e] : 0.f);
    }
    *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + R * RB + (((wn * 4 + fq) ^ f256(R)) * 16)) = o;
    o1v[j] = o;
    st1 |= (ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) ? (1u << j) : 0u;
    if constexpr (!BWD) {
      if (p.b1) b1w[j] = gather_word4(pos_bits8<F16>(o), fq);   // wave-uniform branch: every lane shuffles
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {   // h1 / g2 to HBM
    if ((st1 >> j) & 1u) {
      const int R = wm * 96 + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
      *(bf16x8_t*)(p.o1 + pix * C + cb8) = o1v[j];
    }
  }
  if constexpr (!BWD) {
    if (p.b1) {   // h1 > 0 words: lane fq stores fragment fq's (then fragment 4 + fq's) word of its pixel column
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int j = r * 4 + fq;
        const unsigned w = r == 0 ? sel4(b1w[0], b1w[1], b1w[2], b1w[3]) : sel4(b1w[4], b1w[5], 0u, 0u);
        const int R = wm * 96 + j * 16 + fr;
        const int hy = R / HWD, hx = R - hy * HWD;
        const bool st = j < 6 && ((st1 >> j) & 1u);
        const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
        *(st ? p.b1 + pix * (C / 32) + wn : (unsigned*)g_blk_sink) = w;
      }
    }
  }
  load_unit(5);
  load_unit(6);

  // ================= phase 2: 18 units =================
  f32x4_t acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // phase 3 operand addresses (dealt pixel pi of tile row wm * 4 + j)
  const int chw = wn * 64 + fq * 16;       // + nc * 256
  // Only the pixel index is kept per output pixel (-1: outside the image); the operand / result addresses are formed
  // where they are used (one 64-bit multiply-add each) — seven pointer sets cost 56 registers this kernel does not have.
  // Invalid pixels read the zero page and write a sink line (no branches: see g_blk_sink).
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  int pixj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int y = y0 + wm * 4 + j, x = x0 + pi;
    pixj[j] = ((y < H) && (x < W)) ? (int)(img_pix0 + (int64_t)y * W + x) : -1;
  }
  auto adp = [&](int j) { return pixj[j] >= 0 ? p.a + (int64_t)pixj[j] * C4 + chw : (const bf16_t*)g_blk_zero; };
  auto mkp = [&](int j) { return (pixj[j] >= 0 && p.m3 && !MB) ? p.m3 + (int64_t)pixj[j] * C4 + chw : (const bf16_t*)g_blk_zero; };
  auto o3p = [&](int j) { return pixj[j] >= 0 ? p.o3 + (int64_t)pixj[j] * C4 + chw : (bf16_t*)g_blk_sink; };
  auto o2p = [&](int j) { return pixj[j] >= 0 ? p.o2 + (int64_t)pixj[j] * C + cb8 : (bf16_t*)g_blk_sink; };
  auto b3r = [&](int j) { return (pixj[j] >= 0 && MB) ? p.b3 + (int64_t)pixj[j] * (C4 / 32) + wn * 2 : (const unsigned*)g_blk_zero; };
  // Phase 3's per-pixel operands (addend, ReLU-mask source).  Loads return in issue order, so a load issued behind a
  // batch of LDS-DMA units is not back before those have landed: the first pass's operands are requested in the middle
  // of phase 2 (8 or 16 unconditional loads, accounted for in the counted waits of the six iterations in which they
  // are younger than the awaited unit), the second pass's addend right behind phase 2.
  bf16x8_t ad0[4][2], ad1[4][2], mk3[MB ? 1 : 4][2];
  u32x2_t mw3[MB ? 2 : 1][4];    // MB: both passes' word pairs
  constexpr int NADD = BWD ? 16 : 8;   // loads issued at u == 9: 8 addend + (backward) 8 mask values or 8 word pairs
  auto load_mask3 = [&](int nc) {
    if constexpr (!MB) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) mk3[j][h] = *(const bf16x8_t*)(mkp(j) + nc * 256 + h * 8);
    }
  };
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    // unit u landed for every wave; the slot of unit u - 1 is free.  In flight behind it: units u + 1 .. u + 5
    // (u = 0: .. u + 6), two LDS-DMA instructions each
    if (u == 0) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // + H1 complete
    else if (u >= 10 && u <= 15) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(10 + NADD) : "memory");
    else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (u >= 1) load_unit(u + 6);
    if (u == 9) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) ad0[j][h] = *(const bf16x8_t*)(adp(j) + h * 8);
      if constexpr (BWD && !MB) load_mask3(0);
      if constexpr (MB) {
#pragma unroll
        for (int nc = 0; nc < 2; ++nc)
#pragma unroll
          for (int j = 0; j < 4; ++j) mw3[nc][j] = *(const u32x2_t*)(b3r(j) + nc * (C / 16));
      }
    }
    if (u == 0) BLK_STAMP(4);
    if (u == 9) BLK_STAMP(5);
    const int h = u / 9, t = u - h * 9;
    const int ky = t / 3, kx = t - ky * 3;
    const int oy = BWD ? 2 - ky : ky, ox = BWD ? 2 - kx : kx;
    const char* sW = smem + unit_off(u) + wrow8 * WRB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * WRB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int R = (wm * 4 + j + oy) * HWD + pi + ox;
        xf[j] = lds_read_b128(smem + R * RB + (((h * 8 + kk * 4 + fq) ^ f256(R)) * 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = mfma16<F16>(wf[i], xf[j], acc2[i][j]);
    }
  }
  lds_barrier();   // b4: H1 is dead; the slot of unit 17 is free
  BLK_STAMP(6);
  // second pass's addend (forward; the backward pass, which also carries mask operands, has no registers to spare and
  // requests it with the second mask after the first pass), first pass's affine: in front of unit 24, so that b5's
  // counted wait covers them
  auto load_ad1 = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) ad1[j][h] = *(const bf16x8_t*)(adp(j) + 256 + h * 8);
  };
  if constexpr (!BWD) load_ad1();
  f32x4_t sc3v[4], sh3v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) load_affine(p.sc3, p.sh3, chw + 4 * i, sc3v[i], sh3v[i]);
  unsigned b2w[4];
  {
    f32x4_t sc2v[2], sh2v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) load_affine(p.sc2, p.sh2, cb8 + 4 * i, sc2v[i], sh2v[i]);
    asm volatile("" ::: "memory");   // the loads above stay in front of the unit below
    load_unit(24);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc2[i][j] * sc2v[i] + sh2v[i];
      if constexpr (BWD) {
        if (MB || p.m2) {
          const unsigned m = MB ? (mw2[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk2[MB ? 0 : j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
            v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      const int pr = (wm * 4 + j) * TW + pi;
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + pr * RB + (((wn * 4 + fq) ^ f256(pr)) * 16)) = o;
      *(bf16x8_t*)o2p(j) = o;           // h2 / g1 to HBM (sink line for pixels outside the image)
      if constexpr (!BWD) {
        if (p.b2) b2w[j] = gather_word4(pos_bits8<F16>(o), fq);
      }
    }
    if constexpr (!BWD) {
      if (p.b2) {   // h2 > 0 words: lane fq stores tile row wm * 4 + fq's word of its pixel column
        const int y = y0 + wm * 4 + fq, x = x0 + pi;
        *((y < H && x < W) ? p.b2 + (img_pix0 + (int64_t)y * W + x) * (C / 32) + wn : (unsigned*)g_blk_sink) =
            sel4(b2w[0], b2w[1], b2w[2], b2w[3]);
      }
    }
  }
  // b5: H2 complete; conv3 units 18..21 landed.  Younger than unit 21 in this wave's queue: units 22..24 (6 LDS-DMA)
  // and the 4 unconditional h2 / g1 stores above (the forward's bit-plane words, if any, only make the wait stricter)
  asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  BLK_STAMP(8);

  // ================= phase 3: two passes of 256 channels =================
  const int lrow3 = ((wn & 1) * 64 + (fr >> 2) * 16 + (fr & 3)) * WRB;   // + 4i rows, inside row-half unit wn >> 1
#pragma unroll
  for (int nc = 0; nc < 2; ++nc) {
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc3[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* sW = smem + unit_off(18 + nc * 4 + (wn >> 1) * 2 + h) + lrow3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8_t wf[4], xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = lds_read_b128(sW + i * 4 * WRB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pr = (wm * 4 + j) * TW + pi;
          xf[j] = lds_read_b128(smem + pr * RB + (((h * 8 + kk * 4 + fq) ^ f256(pr)) * 16));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc3[i][j] = mfma16<F16>(wf[i], xf[j], acc3[i][j]);
      }
    }
    if (nc == 0) {
      lds_barrier();     // b6: every wave is done with units 18..21
      load_unit(25);     // into the slot of unit 18
    }
    bf16x8_t ov[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc3[i][j] * sc3v[i] + sh3v[i];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * h][e] += elem_to_f32<F16>((nc == 0 ? ad0 : ad1)[j][h][e]);
          v[2 * h + 1][e] += elem_to_f32<F16>((nc == 0 ? ad0 : ad1)[j][h][4 + e]);
        }
      if constexpr (BWD) {
        unsigned m;
        if constexpr (MB) m = ((fq & 2) ? mw3[MB ? nc : 0][j][1] : mw3[MB ? nc : 0][j][0]) >> (16 * (fq & 1));
        else m = p.m3 ? (pos_bits8<F16>(mk3[MB ? 0 : j][0]) | (pos_bits8<F16>(mk3[MB ? 0 : j][1]) << 8)) : 0xffffu;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * h][e] = ((m >> (8 * h + e)) & 1u) ? v[2 * h][e] : 0.f;
            v[2 * h + 1][e] = ((m >> (8 * h + 4 + e)) & 1u) ? v[2 * h + 1][e] : 0.f;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ov[j][h][e] = f32_to_elem<F16>(v[2 * h][e]);
          ov[j][h][4 + e] = f32_to_elem<F16>(v[2 * h + 1][e]);
        }
      __builtin_amdgcn_sched_barrier(0);   // one pixel at a time: interleaving the four keeps 4 x 16 fp32 temporaries live
    }
    BLK_STAMP(9 + nc);
    __builtin_amdgcn_sched_barrier(0);
    if (nc == 0) {
      if constexpr (BWD) { load_ad1(); load_mask3(1); }
#pragma unroll
      for (int i = 0; i < 4; ++i) load_affine(p.sc3, p.sh3, 256 + chw + 4 * i, sc3v[i], sh3v[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) *(bf16x8_t*)(o3p(j) + nc * 256 + h * 8) = ov[j][h];
    if constexpr (!BWD) {
      if (p.b3) {   // x > 0: the pixel's 64-channel word pair from the four lanes' 16 bits each; lane fq stores row fq's
        const bf16x8_t (&ad)[4][2] = nc == 0 ? ad0 : ad1;
        unsigned lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned v16 = (pos_bits8<F16>(ad[j][0]) | (pos_bits8<F16>(ad[j][1]) << 8)) << (16 * (fq & 1));
          v16 |= __shfl_xor(v16, 16);
          const unsigned other = __shfl_xor(v16, 32);
          lo[j] = (fq & 2) ? other : v16;
          hi[j] = (fq & 2) ? v16 : other;
        }
        const int y = y0 + wm * 4 + fq, x = x0 + pi;
        unsigned* dst = (y < H && x < W) ? p.b3 + (img_pix0 + (int64_t)y * W + x) * (C4 / 32) + wn * 2 + nc * (C / 16)
                                         : (unsigned*)g_blk_sink;
        *(u32x2_t*)dst = (u32x2_t){sel4(lo[0], lo[1], lo[2], lo[3]), sel4(hi[0], hi[1], hi[2], hi[3])};
      }
    }
    // b7: units 22..25 landed.  Behind unit 25 this wave has issued 8 + 8 addend and mask loads (backward without bit
    // planes) or up to 8 affine loads (forward: none if there is no BN), then the pass's 8 stores — all unconditional
    // (sink / zero-page redirect); bit-plane words, if any, only make the wait stricter
    if (nc == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(BWD ? (MB ? 16 : 24) : 8) : "memory");
  }
#ifdef TDN_TRACE_BUILD
  BLK_STAMP(11);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BLK_STAMP(12);
  BLK_STAMP_RT(15);
#endif
}

// ---------------------------------------------------------------------------------------------
// C = 128 with 10 x 16 tiles ("tall").  Why: layer2's 100 x 168 map makes 13 x 11 = 143 tiles of 8 x 16 per image; the
// two images' launches of a step run side by side, 286 workgroups on 256 CUs — two rounds for 1.12 rounds of work
// (stand-alone, both images in one launch: 57.6 us; 242 tiles of the same kernel: 35.2 us).  10 x 16 tiles are
// 10 x 11 = 110 per image, 220 for the pair: one round of 1.25x longer workgroups, and H = 100 has no ragged band.
// Same three phases, 8 waves = 2 pixel halves (7 + 7 patch fragments, 5 + 5 tile rows) x 4 channel quarters:
//   [0, 57344)          H1[224 rows][256 B]   (later H2[160 rows][256 B])
//   [57344, 163840)     phase 1: four K-step slots of 22528 B (X 224 rows x 64 B + W1 128 rows x 64 B); then six 16 KB
//                       unit slots — five over the first 80 KB, the sixth behind the K-step slots (free from the
//                       start: unit 0 is issued there at kernel start), unit u in slot (u + 5) % 6
// Units: u = 0..17 conv2 as in the 8 x 16 kernel; u = 18..25 conv3 in FOUR passes of 128 output channels (pass q: units
// 18 + 2q, 19 + 2q = the two K halves), 8 channels per lane like the other phases — 40 accumulator registers per pass
// instead of 64, which is what makes room for the fifth tile row.  Five units stay in flight (counted vmcnt).
// ---------------------------------------------------------------------------------------------
template <bool BWD, bool F16, bool MB = false>
__global__ __launch_bounds__(512, 1) void bottleneck128t_kernel(const BlockParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 128, C4 = 512, TH = 10, TW = 16, HWD = TW + 2, PH = (TH + 2) * HWD /* 216 */, PHP = 224;
  constexpr int NF1 = PHP / 32;                // 7 patch fragments per wave
  constexpr int NF2 = TH / 2;                  // 5 tile rows per wave
  constexpr int RB = 256;                      // H1 / H2 row bytes
  constexpr int WRB = 128;                     // weight unit row bytes
  constexpr int RING = PHP * RB, UB = 16384;   // 57344
  constexpr int XB32 = PHP * 64, P1SLOT = XB32 + 8192;   // 14336, 22528
  constexpr int KS1 = C4 / 32;                 // 16 K-steps of 32 channels
  constexpr int NOP = BWD ? 2 * NF2 : NF2;     // unconditional per-pixel operand loads of a phase 3 pass (addend [+ mask])
  static_assert(RING + 4 * P1SLOT + UB == 163840, "LDS map");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.nwg_pad >> 3) + (bid >> 3);
  if (tile >= p.ntiles) return;
  const int H = p.H, W = p.W;
  const int tpi = p.tiles_x * p.tiles_y;
  const int img = tile / tpi;
  const int trem = tile - img * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int64_t img_pix0 = (int64_t)img * H * W;

  auto swz_w8 = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  auto f256 = [](int R) { return ((R >> 1) & 7) << 1; };

  // ---- weight units ----
  const int lrow8 = lane >> 3, lchunk8 = lane & 7;
  auto unit_off = [](int u) { return (u + 5) % 6 < 5 ? RING + ((u + 5) % 6) * UB : RING + 4 * P1SLOT; };
  auto load_unit = [&](int u) {
    char* dst = smem + unit_off(u);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int g8 = it * 8 + wave;
      const int r = g8 * 8 + lrow8;                  // row of the unit
      const char* src;
      if (u < 18) {
        const int h = u / 9, t = u - h * 9;
        src = (const char*)p.w2 + ((int64_t)r * (9 * C) + t * C + h * 64) * 2 + ((lchunk8 ^ swz_w8(r)) * 16);
      } else {
        const int v = u - 18, q = v >> 1, h = v & 1;
        src = (const char*)p.w3 + ((int64_t)(q * 128 + r) * C + h * 64) * 2 + ((lchunk8 ^ swz_w8(r)) * 16);
      }
      glds16_async(src, dst + g8 * 8 * WRB);
    }
  };
  load_unit(0);

  // ---- phase 1 loader: per K-step 14 X pieces (16 rows x 64 B) + 8 W1 pieces over 8 waves: every wave issues three
  // LDS-DMA instructions (waves 6, 7 have no second X piece: a dummy copy of the zero page into the idle H1 region
  // keeps the per-wave vmcnt arithmetic uniform) ----
  const int lrow16 = lane >> 2, lchunk4 = lane & 3;
  const char* zero = (const char*)g_zero_page + lchunk4 * 16;
  const char* xsrc[2];
  unsigned xok = 0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int R = (it * 8 + wave) * 16 + lrow16;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    const int swz = (4 - ((R >> 2) & 3)) & 3;
    xsrc[it] = (const char*)p.a + ((img_pix0 + (int64_t)y * W + x) * C4 + ((lchunk4 ^ swz) * 8)) * 2;
    xok |= ok ? (1u << it) : 0u;
  }
  const char* w1src;
  {
    const int n = wave * 16 + lrow16;
    w1src = (const char*)p.w1 + ((int64_t)n * C4) * 2 + ((lchunk4 ^ ((4 - ((n >> 3) & 3)) & 3)) * 16);
  }
  auto load_step = [&](int kc) {
    char* sX = smem + RING + (kc & 3) * P1SLOT;
    glds16_async(xok & 1u ? xsrc[0] + kc * 64 : zero, sX + wave * 1024);
    if (wave < 6) glds16_async(xok & 2u ? xsrc[1] + kc * 64 : zero, sX + (8 + wave) * 1024);
    else glds16_async(zero, smem + wave * 1024);            // dummy, into the idle H1 region
    glds16_async(w1src + kc * 64, sX + XB32 + wave * 1024);
  };

  const int f_rd_w = ((fr & 3) >> 1) | ((fr >> 2) << 1);
  const int f_rd32 = (4 - (fr >> 2)) & 3;
  const int wrow8 = wn * 32 + (fr >> 2) * 8 + (fr & 3);
  const int pi = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * (fr - 8) : 2 * (fr - 4) + 1);
  const int cb8 = wn * 32 + fq * 8;

  auto sel4 = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3) { return fq == 0 ? a0 : (fq == 1 ? a1 : (fq == 2 ? a2 : a3)); };
  auto load_affine = [&](const float* scp, const float* shp, int ch, f32x4_t& sc, f32x4_t& sh) {
    sc = (!BWD && scp) ? *(const f32x4_t*)(scp + ch) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
    sh = (!BWD && shp) ? *(const f32x4_t*)(shp + ch) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  f32x4_t sc1v[2], sh1v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) load_affine(p.sc1, p.sh1, cb8 + 4 * i, sc1v[i], sh1v[i]);
  bf16x8_t mk1[MB ? 1 : NF1];
  unsigned mw1[MB ? NF1 : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < NF1; ++j) {
      const int R = wm * (NF1 * 16) + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw1[j] = *(ok ? p.b2 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk1[j] = *(const bf16x8_t*)((ok && p.m1) ? p.m1 + pix * C + cb8 : (const bf16_t*)g_blk_zero);
    }
  }

  // ================= phase 1 =================
  f32x4_t acc1[2][NF1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF1; ++j) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  load_step(0);
  load_step(1);
  load_step(2);
#pragma unroll
  for (int kc = 0; kc < KS1; ++kc) {
    // younger than K-step kc at this point: two K-steps (3 LDS-DMA each), or the units that take their place at the tail
    if (kc <= KS1 - 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (kc == KS1 - 2) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kc + 3 < KS1) load_step(kc + 3);
    else load_unit(kc + 3 - KS1 + 1);            // kc = 13, 14, 15 -> units 1, 2, 3: unit slots 0, 1, 2 lie inside the
                                                 // K-step slots released by then (0; 0 + 1; 1 + 2)
    const char* sX = smem + RING + (kc & 3) * P1SLOT + (wm * (NF1 * 16) + fr) * 64 + ((fq ^ f_rd32) * 16);
    const char* sW = smem + RING + (kc & 3) * P1SLOT + XB32 + wrow8 * 64 + ((fq ^ f_rd32) * 16);
    bf16x8_t wf[2], xf[NF1];
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * 64);
#pragma unroll
    for (int j = 0; j < NF1; ++j) xf[j] = lds_read_b128(sX + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NF1; ++j) acc1[i][j] = mfma16<F16>(wf[i], xf[j], acc1[i][j]);
  }
  lds_barrier();   // b0: the K-step slots are free
  // compiler-visible loads / stores of this stretch go in front of units 4 and 5 (see the 8 x 16 kernel)
  bf16x8_t mk2[MB ? 1 : NF2];
  unsigned mw2[MB ? NF2 : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < NF2; ++j) {
      const int y = y0 + wm * NF2 + j, x = x0 + pi;
      const bool ok = (y < H) && (x < W);
      const int64_t pix = img_pix0 + (int64_t)y * W + x;
      if constexpr (MB) mw2[j] = *(ok ? p.b1 + pix * (C / 32) + wn : (const unsigned*)g_blk_zero);
      else mk2[j] = *(const bf16x8_t*)((p.m2 && ok) ? p.m2 + pix * C + cb8 : (const bf16_t*)g_blk_zero);
    }
  }

  bf16x8_t o1v[NF1];
  unsigned b1w[8];
  unsigned st1 = 0;
  b1w[7] = 0u;
#pragma unroll
  for (int j = 0; j < NF1; ++j) {
    const int R = wm * (NF1 * 16) + j * 16 + fr;
    const int hy = R / HWD, hx = R - hy * HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = (R < PH) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);
    f32x4_t v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = acc1[i][j] * sc1v[i] + sh1v[i];
    if constexpr (BWD) {
      if (MB || p.m1) {
        const unsigned m = MB ? (mw1[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk1[MB ? 0 : j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
          v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      o[e] = f32_to_elem<F16>(ok ? v[0][e] : 0.f);
      o[4 + e] = f32_to_elem<F16>(ok ? v[1][e] : 0.f);
    }
    *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + R * RB + (((wn * 4 + fq) ^ f256(R)) * 16)) = o;
    o1v[j] = o;
    st1 |= (ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW) ? (1u << j) : 0u;
    b1w[j] = 0u;
    if constexpr (!BWD) {
      if (p.b1) b1w[j] = gather_word4(pos_bits8<F16>(o), fq);   // wave-uniform branch: every lane shuffles
    }
  }
#pragma unroll
  for (int j = 0; j < NF1; ++j) {   // h1 / g2 to HBM
    if ((st1 >> j) & 1u) {
      const int R = wm * (NF1 * 16) + j * 16 + fr;
      const int hy = R / HWD, hx = R - hy * HWD;
      const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
      *(bf16x8_t*)(p.o1 + pix * C + cb8) = o1v[j];
    }
  }
  if constexpr (!BWD) {
    if (p.b1) {   // h1 > 0 words: lane fq stores fragment fq's (then fragment 4 + fq's) word of its pixel column
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int j = r * 4 + fq;
        const unsigned w = r == 0 ? sel4(b1w[0], b1w[1], b1w[2], b1w[3]) : sel4(b1w[4], b1w[5], b1w[6], 0u);
        const int R = wm * (NF1 * 16) + j * 16 + fr;
        const int hy = R / HWD, hx = R - hy * HWD;
        const bool st = j < NF1 && ((st1 >> j) & 1u);
        const int64_t pix = img_pix0 + (int64_t)(y0 - 1 + hy) * W + (x0 - 1 + hx);
        *(st ? p.b1 + pix * (C / 32) + wn : (unsigned*)g_blk_sink) = w;
      }
    }
  }
  load_unit(4);
  load_unit(5);

  // ================= phase 2: 18 units =================
  f32x4_t acc2[2][NF2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF2; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // per output pixel (dealt pixel pi of tile row wm * 5 + j) only the pixel index is kept (-1: outside the image)
  int pixj[NF2];
#pragma unroll
  for (int j = 0; j < NF2; ++j) {
    const int y = y0 + wm * NF2 + j, x = x0 + pi;
    pixj[j] = ((y < H) && (x < W)) ? (int)(img_pix0 + (int64_t)y * W + x) : -1;
  }
  // phase 3 operands of pass q: this lane's 8 channels q * 128 + cb8 .. + 7 of its five pixels
  bf16x8_t ad[NF2], mk3[MB ? 1 : NF2];
  unsigned mw3[MB ? NF2 : 1];
  auto load_ops3 = [&](int q) {
#pragma unroll
    for (int j = 0; j < NF2; ++j)
      ad[j] = *(const bf16x8_t*)(pixj[j] >= 0 ? p.a + (int64_t)pixj[j] * C4 + q * 128 + cb8 : (const bf16_t*)g_blk_zero);
    if constexpr (BWD && !MB) {
#pragma unroll
      for (int j = 0; j < NF2; ++j)
        mk3[j] = *(const bf16x8_t*)((pixj[j] >= 0 && p.m3) ? p.m3 + (int64_t)pixj[j] * C4 + q * 128 + cb8 : (const bf16_t*)g_blk_zero);
    }
    if constexpr (MB) {
#pragma unroll
      for (int j = 0; j < NF2; ++j)
        mw3[j] = *(pixj[j] >= 0 ? p.b3 + (int64_t)pixj[j] * (C4 / 32) + q * 4 + wn : (const unsigned*)g_blk_zero);
    }
  };
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    // unit u landed for every wave; the slot of unit u - 1 is free.  In flight behind it: units u + 1 .. u + 4
    // (u = 0: .. u + 5), two LDS-DMA instructions each; u = 10 .. 14: also the NOP operand loads issued at u == 9
    if (u == 0) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // + H1 complete
    else if (u >= 10 && u <= 14) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(8 + NOP) : "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (u >= 1) load_unit(u + 5);
    if (u == 9) load_ops3(0);
    const int h = u / 9, t = u - h * 9;
    const int ky = t / 3, kx = t - ky * 3;
    const int oy = BWD ? 2 - ky : ky, ox = BWD ? 2 - kx : kx;
    const char* sW = smem + unit_off(u) + wrow8 * WRB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t wf[2], xf[NF2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * WRB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
      for (int j = 0; j < NF2; ++j) {
        const int R = (wm * NF2 + j + oy) * HWD + pi + ox;
        xf[j] = lds_read_b128(smem + R * RB + (((h * 8 + kk * 4 + fq) ^ f256(R)) * 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF2; ++j) acc2[i][j] = mfma16<F16>(wf[i], xf[j], acc2[i][j]);
    }
  }
  lds_barrier();   // b4: H1 is dead; the slot of unit 17 is free
  f32x4_t sc3v[2], sh3v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) load_affine(p.sc3, p.sh3, cb8 + 4 * i, sc3v[i], sh3v[i]);
  unsigned b2w[8];
  {
    f32x4_t sc2v[2], sh2v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) load_affine(p.sc2, p.sh2, cb8 + 4 * i, sc2v[i], sh2v[i]);
    asm volatile("" ::: "memory");   // the loads above stay in front of the unit below
    load_unit(23);
#pragma unroll
    for (int j = 0; j < NF2; ++j) {
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc2[i][j] * sc2v[i] + sh2v[i];
      if constexpr (BWD) {
        if (MB || p.m2) {
          const unsigned m = MB ? (mw2[MB ? j : 0] >> (8 * fq)) : pos_bits8<F16>(mk2[MB ? 0 : j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
            v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
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
      const int pr = (wm * NF2 + j) * TW + pi;
      *(TDN_LDS bf16x8_t*)(TDN_LDS char*)(smem + pr * RB + (((wn * 4 + fq) ^ f256(pr)) * 16)) = o;
      *(bf16x8_t*)(pixj[j] >= 0 ? p.o2 + (int64_t)pixj[j] * C + cb8 : (bf16_t*)g_blk_sink) = o;   // h2 / g1 to HBM
      b2w[j] = 0u;
      if constexpr (!BWD) {
        if (p.b2) b2w[j] = gather_word4(pos_bits8<F16>(o), fq);
      }
    }
    if constexpr (!BWD) {
      if (p.b2) {   // h2 > 0 words: lane fq stores tile row wm * 5 + fq's word of its pixel column, lane 0's group row 4's
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const unsigned w = r == 0 ? sel4(b2w[0], b2w[1], b2w[2], b2w[3]) : b2w[4];
          const int pj = r == 0 ? (int)sel4((unsigned)pixj[0], (unsigned)pixj[1], (unsigned)pixj[2], (unsigned)pixj[3]) : pixj[4];
          const bool st = pj >= 0 && (r == 0 || fq == 0);
          *(st ? p.b2 + (int64_t)pj * (C / 32) + wn : (unsigned*)g_blk_sink) = w;
        }
      }
    }
  }
  // b5: H2 complete; conv3 units 18, 19 landed.  Younger than unit 19 in this wave's queue: units 20 .. 23 (8 LDS-DMA)
  // and the 5 unconditional h2 / g1 stores above
  asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // ================= phase 3: four passes of 128 channels =================
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __builtin_amdgcn_sched_barrier(0);
    // pass q > 0: its units 18 + 2q, 19 + 2q have landed for every wave.  Younger than unit 19 + 2q in this wave's
    // queue (all unconditional): see the table in the comment of each case
    if (q == 1)        // unit 22, 23 (4), h2 stores (5), units 24, 25 (4), pass 1 operands (NOP), pass 0 stores (5)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(18 + NOP) : "memory");
    else if (q == 2)   // h2 stores (5), units 24, 25 (4), operands of passes 1, 2 (2 NOP), stores of passes 0, 1 (10)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(19 + 2 * NOP) : "memory");
    else if (q == 3)   // operands of passes 1, 2, 3 (3 NOP), stores of passes 0, 1, 2 (15)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(15 + 3 * NOP) : "memory");
    f32x4_t acc3[2][NF2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NF2; ++j) acc3[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* sW = smem + unit_off(18 + 2 * q + h) + wrow8 * WRB;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8_t wf[2], xf[NF2];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = lds_read_b128(sW + i * 4 * WRB + (((kk * 4 + fq) ^ f_rd_w) * 16));
#pragma unroll
        for (int j = 0; j < NF2; ++j) {
          const int pr = (wm * NF2 + j) * TW + pi;
          xf[j] = lds_read_b128(smem + pr * RB + (((h * 8 + kk * 4 + fq) ^ f256(pr)) * 16));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NF2; ++j) acc3[i][j] = mfma16<F16>(wf[i], xf[j], acc3[i][j]);
      }
    }
    if (q == 0) {
      lds_barrier();     // b6: every wave is done with units 18, 19
      load_unit(24);     // into their slots
      load_unit(25);
    }
    // this pass's epilogue values, then (opaque uses / definition: see bottleneck64_kernel) the next pass's operands
    bf16x8_t ov[NF2];
    unsigned b3w[8];
#pragma unroll
    for (int j = 0; j < NF2; ++j) {
      f32x4_t v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = acc3[i][j] * sc3v[i] + sh3v[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[0][e] += elem_to_f32<F16>(ad[j][e]);
        v[1][e] += elem_to_f32<F16>(ad[j][4 + e]);
      }
      if constexpr (BWD) {
        unsigned m;
        if constexpr (MB) m = mw3[MB ? j : 0] >> (8 * fq);
        else m = p.m3 ? pos_bits8<F16>(mk3[MB ? 0 : j]) : 0xffu;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[0][e] = ((m >> e) & 1u) ? v[0][e] : 0.f;
          v[1][e] = ((m >> (4 + e)) & 1u) ? v[1][e] : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ov[j][e] = f32_to_elem<F16>(v[0][e]);
        ov[j][4 + e] = f32_to_elem<F16>(v[1][e]);
      }
      b3w[j] = 0u;
      if constexpr (!BWD) {
        if (p.b3) b3w[j] = gather_word4(pos_bits8<F16>(ad[j]), fq);
      }
    }
#pragma unroll
    for (int j = 0; j < NF2; ++j) asm volatile("" ::"v"(ov[j]));
    __builtin_amdgcn_sched_barrier(0);
    if (q < 3) {
      int qn = q + 1;
      asm volatile("" : "+s"(qn));
      load_ops3(qn);
#pragma unroll
      for (int i = 0; i < 2; ++i) load_affine(p.sc3, p.sh3, qn * 128 + cb8 + 4 * i, sc3v[i], sh3v[i]);
    }
#pragma unroll
    for (int j = 0; j < NF2; ++j)
      *(bf16x8_t*)(pixj[j] >= 0 ? p.o3 + (int64_t)pixj[j] * C4 + q * 128 + cb8 : (bf16_t*)g_blk_sink) = ov[j];
    if constexpr (!BWD) {
      if (p.b3) {   // x > 0 words of this pass's 128 channels: word q * 4 + wn of the pixel
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const unsigned w = r == 0 ? sel4(b3w[0], b3w[1], b3w[2], b3w[3]) : b3w[4];
          const int pj = r == 0 ? (int)sel4((unsigned)pixj[0], (unsigned)pixj[1], (unsigned)pixj[2], (unsigned)pixj[3]) : pixj[4];
          const bool st = pj >= 0 && (r == 0 || fq == 0);
          *(st ? p.b3 + (int64_t)pj * (C4 / 32) + q * 4 + wn : (unsigned*)g_blk_sink) = w;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <bool BWD, bool F16, bool MB = false, int HEAD = 0>
static int launch_block64(BlockParams& p, hipStream_t stream) {
  constexpr int lds = 81920;
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)bottleneck64_kernel<BWD, F16, MB, HEAD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
    attr_once.mark();
    if (getenv("TDN_DEBUG_OCC")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)bottleneck64_kernel<BWD, F16, MB, HEAD>, 256, lds);
      fprintf(stderr, "[tdn] bottleneck64<%d,%d,%d,%d>: %d B LDS, %d workgroups/CU\n", (int)BWD, (int)F16, (int)MB, (int)HEAD, lds, nb);
    }
  }
  TDN_LAUNCH((bottleneck64_kernel<BWD, F16, MB, HEAD>), dim3(p.nwg_pad), dim3(256), lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

#ifdef TDN_TRACE_BUILD
static unsigned long long* g_blk_trace = nullptr;
extern "C" int tdn_debug_block_trace(void* buf) { g_blk_trace = (unsigned long long*)buf; return 0; }   // >= 128 B per workgroup
#endif

template <bool BWD, bool F16, bool MB = false>
static int launch_block128(BlockParams& p, hipStream_t stream) {
  constexpr int lds = 163840;
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)bottleneck128_kernel<BWD, F16, MB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
    attr_once.mark();
  }
  TDN_LAUNCH((bottleneck128_kernel<BWD, F16, MB>), dim3(p.nwg_pad), dim3(512), lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

// Tile height of the C = 128 kernels for this launch: 8 or 10 rows, whichever takes fewer workgroup-rounds x rows on
// the chip's 256 CUs (one workgroup per CU).  Only the launch's own tiles count: the two single-image launches of a
// batch-2 step (one chain per image) do not run in lock-step — measured in the step, 8 and 10 rows are equal there
// (525.7 vs 526.5 img/s) while a single image alone is 10 % faster with 8 rows; launches of two or more images
// (R101 at 4 images per GPU: two chains of two; TDN_IMG_SPLIT_M below the layer2 size) take 10 rows and gain a third
// (both images of 100 x 168: 57.4 -> 38.7 us).  TDN_BLOCK128_TH=8 / 10 forces one.
static int block128_th(const BlockParams& p) {
  const char* e = getenv("TDN_BLOCK128_TH");
  if (e && *e) return atoi(e) == 10 ? 10 : 8;
  const int tx = ceil_div(p.W, 16);
  const int r8 = ceil_div(p.N * tx * ceil_div(p.H, 8), 256) * 8;
  const int r10 = ceil_div(p.N * tx * ceil_div(p.H, 10), 256) * 10;
  return r10 < r8 ? 10 : 8;
}

template <bool BWD, bool F16, bool MB = false>
static int launch_block128t(BlockParams& p, hipStream_t stream) {
  constexpr int lds = 163840;
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)bottleneck128t_kernel<BWD, F16, MB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
    attr_once.mark();
  }
  p.tiles_y = ceil_div(p.H, 10);
  p.ntiles = p.N * p.tiles_x * p.tiles_y;
  p.nwg_pad = (p.ntiles + 7) & ~7;
  TDN_LAUNCH((bottleneck128t_kernel<BWD, F16, MB>), dim3(p.nwg_pad), dim3(512), lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_bottleneck_supported(int H, int W, int C, int stride, int dilation) {
  return ((C == 64 || C == 128) && stride == 1 && dilation == 1 && H > 0 && W > 0) ? 1 : 0;
}

static int block_common(BlockParams& p, const tdn_bottleneck_args* a, int dtype) {
  TDN_CHECK_DTYPE(dtype);
  TDN_CHECK(a != nullptr, "bottleneck: NULL argument block");
  TDN_CHECK(a->N > 0 && a->H > 0 && a->W > 0, "bottleneck: bad tensor shape N=%d H=%d W=%d", a->N, a->H, a->W);
  TDN_CHECK(tdn_bottleneck_supported(a->H, a->W, a->C, 1, 1), "bottleneck: C=%d is not built (64, 128)", a->C);
  TDN_CHECK(a->in && a->w1 && a->w2 && a->w3 && a->out1 && a->out2 && a->out3, "bottleneck: NULL tensor pointer");
  TDN_CHECK((int64_t)a->N * a->H * a->W < (1ll << 31) / 4, "tensor too large for 32-bit pixel indexing");
  memset(&p, 0, sizeof(p));
  p.a = (const bf16_t*)a->in; p.w1 = (const bf16_t*)a->w1; p.w2 = (const bf16_t*)a->w2; p.w3 = (const bf16_t*)a->w3;
  p.o1 = (bf16_t*)a->out1; p.o2 = (bf16_t*)a->out2; p.o3 = (bf16_t*)a->out3;
  p.N = a->N; p.H = a->H; p.W = a->W;
  p.tiles_x = ceil_div(a->W, 16); p.tiles_y = ceil_div(a->H, 8);
  p.ntiles = a->N * p.tiles_x * p.tiles_y;
  p.nwg_pad = (p.ntiles + 7) & ~7;
#ifdef TDN_TRACE_BUILD
  p.trace = g_blk_trace;
#endif
  return 0;
}

extern "C" int tdn_bottleneck_fwd(const tdn_bottleneck_args* a, int dtype, void* stream) {
  BlockParams p;
  if (block_common(p, a, dtype)) return -1;
  p.sc1 = a->scale1; p.sh1 = a->shift1; p.sc2 = a->scale2; p.sh2 = a->shift2; p.sc3 = a->scale3; p.sh3 = a->shift3;
  p.b1 = (unsigned*)a->bits1; p.b2 = (unsigned*)a->bits2; p.b3 = (unsigned*)a->bits3;   // optional outputs
  if (a->C == 128) {
    if (block128_th(p) == 10) {
      if (dtype == TDN_F16) return launch_block128t<false, true>(p, (hipStream_t)stream);
      return launch_block128t<false, false>(p, (hipStream_t)stream);
    }
    if (dtype == TDN_F16) return launch_block128<false, true>(p, (hipStream_t)stream);
    return launch_block128<false, false>(p, (hipStream_t)stream);
  }
  if (dtype == TDN_F16) return launch_block64<false, true>(p, (hipStream_t)stream);
  return launch_block64<false, false>(p, (hipStream_t)stream);
}

extern "C" int tdn_bottleneck_dgrad(const tdn_bottleneck_args* a, int dtype, void* stream) {
  BlockParams p;
  if (block_common(p, a, dtype)) return -1;
  p.m1 = (const bf16_t*)a->mask1; p.m2 = (const bf16_t*)a->mask2; p.m3 = (const bf16_t*)a->mask3;
  p.b1 = (unsigned*)a->bits1; p.b2 = (unsigned*)a->bits2; p.b3 = (unsigned*)a->bits3;
  const bool mb = a->bits1 || a->bits2 || a->bits3;
  if (mb) {
    TDN_CHECK(a->bits1 && a->bits2 && a->bits3, "bottleneck dgrad: all three bit planes or none");
    if (a->C == 128) {
      if (block128_th(p) == 10) {
        if (dtype == TDN_F16) return launch_block128t<true, true, true>(p, (hipStream_t)stream);
        return launch_block128t<true, false, true>(p, (hipStream_t)stream);
      }
      if (dtype == TDN_F16) return launch_block128<true, true, true>(p, (hipStream_t)stream);
      return launch_block128<true, false, true>(p, (hipStream_t)stream);
    }
    if (dtype == TDN_F16) return launch_block64<true, true, true>(p, (hipStream_t)stream);
    return launch_block64<true, false, true>(p, (hipStream_t)stream);
  }
  if (a->C == 128) {
    if (block128_th(p) == 10) {
      if (dtype == TDN_F16) return launch_block128t<true, true>(p, (hipStream_t)stream);
      return launch_block128t<true, false>(p, (hipStream_t)stream);
    }
    if (dtype == TDN_F16) return launch_block128<true, true>(p, (hipStream_t)stream);
    return launch_block128<true, false>(p, (hipStream_t)stream);
  }
  if (dtype == TDN_F16) return launch_block64<true, true>(p, (hipStream_t)stream);
  return launch_block64<true, false>(p, (hipStream_t)stream);
}

// ---- head block (layer1.0): C input channels, 1x1 downsample on the residual branch ----
extern "C" int tdn_bottleneck_head_supported(int H, int W, int Cin, int C, int stride, int dilation) {
  return (Cin == 64 && C == 64 && stride == 1 && dilation == 1 && H > 0 && W > 0) ? 1 : 0;
}

extern "C" int tdn_bottleneck_head_fwd(const tdn_bottleneck_head_args* a, int dtype, void* stream) {
  BlockParams p;
  TDN_CHECK(a != nullptr, "bottleneck head: NULL argument block");
  if (block_common(p, &a->b, dtype)) return -1;
  TDN_CHECK(a->b.C == 64, "bottleneck head: C=%d is not built (64)", a->b.C);
  TDN_CHECK((a->addend != nullptr) != (a->wd != nullptr),
            "bottleneck head fwd: give either the downsample branch (addend) or its weights (wd), not both / neither");
  TDN_CHECK(a->b.bits3 == nullptr, "bottleneck head fwd: there is no bits3 plane (the block input is not masked)");
  p.sc1 = a->b.scale1; p.sh1 = a->b.shift1; p.sc2 = a->b.scale2; p.sh2 = a->b.shift2; p.sc3 = a->b.scale3; p.sh3 = a->b.shift3;
  p.b1 = (unsigned*)a->b.bits1; p.b2 = (unsigned*)a->b.bits2;
  p.ad = (const bf16_t*)a->addend;
  if (a->wd) {
    p.wd = (const bf16_t*)a->wd; p.scd = a->scale_d; p.shd = a->shift_d;
    if (dtype == TDN_F16) return launch_block64<false, true, false, 2>(p, (hipStream_t)stream);
    return launch_block64<false, false, false, 2>(p, (hipStream_t)stream);
  }
  if (dtype == TDN_F16) return launch_block64<false, true, false, 1>(p, (hipStream_t)stream);
  return launch_block64<false, false, false, 1>(p, (hipStream_t)stream);
}

extern "C" int tdn_bottleneck_head_dgrad(const tdn_bottleneck_head_args* a, int dtype, void* stream) {
  BlockParams p;
  TDN_CHECK(a != nullptr, "bottleneck head: NULL argument block");
  if (block_common(p, &a->b, dtype)) return -1;
  TDN_CHECK(a->b.C == 64, "bottleneck head: C=%d is not built (64)", a->b.C);
  TDN_CHECK((a->addend != nullptr) != (a->wd != nullptr),
            "bottleneck head dgrad: give either the downsample conv's input gradient (addend) or its w_dgrad (wd)");
  TDN_CHECK(a->b.mask3 == nullptr && a->b.bits3 == nullptr, "bottleneck head dgrad: the block input gradient takes no mask");
  p.m1 = (const bf16_t*)a->b.mask1; p.m2 = (const bf16_t*)a->b.mask2;
  p.b1 = (unsigned*)a->b.bits1; p.b2 = (unsigned*)a->b.bits2;
  p.ad = (const bf16_t*)a->addend;
  p.wd = (const bf16_t*)a->wd;
  if (a->b.bits1 || a->b.bits2) {
    TDN_CHECK(a->b.bits1 && a->b.bits2, "bottleneck head dgrad: both bit planes or none");
    if (a->wd) {
      if (dtype == TDN_F16) return launch_block64<true, true, true, 2>(p, (hipStream_t)stream);
      return launch_block64<true, false, true, 2>(p, (hipStream_t)stream);
    }
    if (dtype == TDN_F16) return launch_block64<true, true, true, 1>(p, (hipStream_t)stream);
    return launch_block64<true, false, true, 1>(p, (hipStream_t)stream);
  }
  if (a->wd) {
    if (dtype == TDN_F16) return launch_block64<true, true, false, 2>(p, (hipStream_t)stream);
    return launch_block64<true, false, false, 2>(p, (hipStream_t)stream);
  }
  if (dtype == TDN_F16) return launch_block64<true, true, false, 1>(p, (hipStream_t)stream);
  return launch_block64<true, false, false, 1>(p, (hipStream_t)stream);
}
