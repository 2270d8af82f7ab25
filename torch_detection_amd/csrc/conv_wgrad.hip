// Weight-gradient GEMM on MFMA (gfx950) + finalize (BN-affine / bias grads, split-K reduction).
//
// Autograd of nn.Conv2d weights / BatchNorm2d affine / conv bias for the layers built at
// models/backbone/resnet.py:74-91,214-216 and models/necks/fpn.py:44-58 (the reference never calls
// backward itself; the oracle is torch autograd on the CPU restatement).
//
//   G[co][tap][ci] = sum_m g[m][co] * x[pix(m) + tap][ci]          (m = output pixel)
// Both operands are pixel-major (NHWC), i.e. the reduction index m is the SLOW index of both tiles, so
// fragments are fetched with ds_read_b64_tr_b16 (hardware transpose read) from row-major LDS tiles that
// are filled by 16-byte LDS-DMA.  D is kept as D[ci][co] so a lane owns 4 consecutive ci (float4 stores).
// sum_m g[m][co] (dbeta / dbias) comes from one extra MFMA against a ones fragment — no extra traffic.
// Split-K over pixels writes fp32 slabs; tdn finalize reduces them in a fixed order (deterministic).
#include "common.h"
#include <type_traits>

struct WgradParams {
  const bf16_t* x;
  const bf16_t* g;
  float* slab;      // [splitk][Cout][Ktot]
  float* colsum;    // [splitk][Cout]
  int Hin, Win, Cpix, Ktap;   // x geometry (pixel stride Cpix elements, Ktap elements consumed per tap)
  int Ho, Wo, Cout, sa;
  int M, Mchunk, splitk;
  int ntaps, Ktot;
  int tiles_co, tiles_k;     // tiles over Cout, tiles over (tap, ci)
  int taps[9];               // (dh+64) | (dw+64)<<8
  int grouped;               // block-diagonal grouped conv: the ci block of a tile is its co block (64 x 64 tiles)
};


// 32-byte-chunk XOR swizzle for tr-read tiles, by row bytes.
template <int RB>
__device__ __forceinline__ int tr_swz(int row) {
  if constexpr (RB >= 256) return (row & 3) | (((row >> 3) & 1) << 2);
  else if constexpr (RB == 128) return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
  else return (row >> 3) & 1;
}

template <int BMW /*co*/, int BNW /*ci*/, int WM = 2, int WN = 2, bool F16 = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKW = 64;                       // pixels per stage
  constexpr int RBG = BMW * 2, RBX = BNW * 2;   // row bytes
  constexpr int G_BYTES = BKW * RBG, X_BYTES = BKW * RBX, STAGE = G_BYTES + X_BYTES;
  constexpr int RPIG = 1024 / RBG, RPIX = 1024 / RBX;   // rows per wave-instruction
  constexpr int NW = WM * WN;
  constexpr int G_IT = BKW / (RPIG * NW) > 0 ? BKW / (RPIG * NW) : 1;
  constexpr int X_IT = BKW / (RPIX * NW) > 0 ? BKW / (RPIX * NW) : 1;
  constexpr bool G_PART = (RPIG * NW > BKW), X_PART = (RPIX * NW > BKW);  // fewer than NW waves needed
  constexpr int WTM = BMW / WM, WTN = BNW / WN, FM = WTM / 16, FN = WTN / 16;  // per-wave co / ci frags
  static_assert(FM >= 1 && FN >= 1, "tile too small");
  // the per-lane swizzle constants assume the row offset between a lane's loads keeps row bits 0..3
  static_assert((RPIG * NW) % 16 == 0 && (RPIX * NW) % 16 == 0, "loader round must be a multiple of 16 rows");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware work map: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch; speed only).  XCD x owns
  // the pixel splits s = x (mod 8) and walks all tiles of one split before the next, so the ~ntiles workgroups
  // that stream the same g / x pixel range run together on ONE L2 instead of being dealt over all eight.
  const int ntiles = p.tiles_co * p.tiles_k;
  int split, tile;
  if (p.splitk >= 8) {
    const int xj = blockIdx.x >> 3;
    split = (blockIdx.x & 7) + 8 * (xj / ntiles);
    tile = xj - (xj / ntiles) * ntiles;
  } else {   // too few splits to feed 8 XCDs that way: plain order, tile fastest
    split = blockIdx.x / ntiles;
    tile = blockIdx.x - split * ntiles;
  }
  if (split >= p.splitk) return;
  const int tile_co = tile % p.tiles_co, tile_k = tile / p.tiles_co;
  const int co0 = tile_co * BMW;
  const int kt_per_tap = p.Ktap / BNW;
  const int tap_i = tile_k / kt_per_tap;
  const int ci_k = (tile_k - tap_i * kt_per_tap) * BNW;   // column offset inside the tap's K range (slab index)
  const int ci0 = p.grouped ? co0 : ci_k;                 // channel offset in x
  const int tp = p.taps[tap_i];
  const int dh = (tp & 0xff) - 64, dw = ((tp >> 8) & 0xff) - 64;
  const int m_begin = split * p.Mchunk;
  const int m_end = min(p.M, m_begin + p.Mchunk);

  // ---- loader constants ----
  constexpr int CPRG = RBG / 16, CPRX = RBX / 16;  // 16B chunks per row
  const int g_lrow = lane / CPRG, g_pc = lane % CPRG;
  const int x_lrow = lane / CPRX, x_pc = lane % CPRX;
  const int g_row0 = wave * RPIG + g_lrow;   // + it*RPIG*NW
  const int x_row0 = wave * RPIX + x_lrow;
  const int g_src_el = ((((g_pc >> 1) ^ tr_swz<RBG>(g_row0)) << 1) | (g_pc & 1)) * 8;
  const int x_src_el = ((((x_pc >> 1) ^ tr_swz<RBX>(x_row0)) << 1) | (x_pc & 1)) * 8;
  const bf16_t* zero = (const bf16_t*)g_zero_page;

  // pixel decode state for this thread's x rows (advanced incrementally by BKW per stage)
  int xa[X_IT], xb[X_IT], ximg[X_IT];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    const int m = m_begin + it * (RPIX * NW) + x_row0;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    ximg[it] = img;
    xa[it] = rem / p.Wo;
    xb[it] = rem - xa[it] * p.Wo;
  }

  auto stage_load = [&](int mt, int s) {
    char* sG = smem + s * STAGE;
    char* sX = sG + G_BYTES;
    if (!G_PART || g_row0 < BKW) {
#pragma unroll
      for (int it = 0; it < G_IT; ++it) {
        const int r = it * (RPIG * NW) + g_row0;
        const int m = mt + r;
        const bf16_t* src = (m < m_end) ? p.g + ((int64_t)m * p.Cout + co0 + g_src_el) : zero + (g_src_el & 127);
        glds16_async(src, sG + (it * (RPIG * NW) + wave * RPIG) * RBG);
      }
    }
    if (!X_PART || x_row0 < BKW) {
#pragma unroll
      for (int it = 0; it < X_IT; ++it) {
        const int r = it * (RPIX * NW) + x_row0;
        const int m = mt + r;
        const int h = xa[it] * p.sa + dh, w = xb[it] * p.sa + dw;
        const bool ok = (m < m_end) && ((unsigned)h < (unsigned)p.Hin) && ((unsigned)w < (unsigned)p.Win);
        const bf16_t* src = ok ? p.x + (((int64_t)(ximg[it] * p.Hin + h) * p.Win + w) * p.Cpix + ci0 + x_src_el)
                               : zero + (x_src_el & 127);
        glds16_async(src, sX + (it * (RPIX * NW) + wave * RPIX) * RBX);
        // advance to the next stage's pixel
        xb[it] += BKW;
        while (xb[it] >= p.Wo) { xb[it] -= p.Wo; xa[it] += 1; }
        while (xa[it] >= p.Ho) { xa[it] -= p.Ho; ximg[it] += 1; }
      }
    }
  };

  // ---- fragment reader constants (ds_read_b64_tr_b16) ----
  const int wm = wave / WN, wn = wave % WN;
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int rrow = 8 * grp + q;                 // + kk*32 + 4*half
  const int fG = tr_swz<RBG>(rrow), fX = tr_swz<RBX>(rrow);
  int g_off[FM], x_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int c5 = (wm * WTM + i * 16) >> 4;    // 32-byte chunk index of this fragment's 16 channels
    g_off[i] = rrow * RBG + ((c5 ^ fG) << 5) + pp * 8;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c5 = (wn * WTN + j * 16) >> 4;
    x_off[j] = rrow * RBX + ((c5 ^ fX) << 5) + pp * 8;
  }

  f32x4_t acc[FM][FN];
  f32x4_t acc1[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    acc1[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_colsum = (tile_k == 0) && (wn == 0);
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = f32_to_elem<F16>(1.0f);

  const int T = (m_end > m_begin) ? ceil_div(m_end - m_begin, BKW) : 0;
  if (T > 0) {
    stage_load(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) stage_load(m_begin + (t + 1) * BKW, (t + 1) & 1);
      const char* sG = smem + (t & 1) * STAGE;
      const char* sX = sG + G_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8_t gf[FM], xf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const s16x4_t lo = lds_read_tr16(sG + g_off[i] + kk * 32 * RBG);
          const s16x4_t hi = lds_read_tr16(sG + g_off[i] + (kk * 32 + 4) * RBG);
          s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          gf[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const s16x4_t lo = lds_read_tr16(sX + x_off[j] + kk * 32 * RBX);
          const s16x4_t hi = lds_read_tr16(sX + x_off[j] + (kk * 32 + 4) * RBX);
          s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          xf[j] = __builtin_bit_cast(bf16x8_t, v);
        }
        // D[row = ci][col = co] += sum_m X[m][ci] * G[m][co]
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = mfma16<F16>(xf[j], gf[i], acc[i][j]);
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc1[i] = mfma16<F16>(ones, gf[i], acc1[i]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- store: lane holds ci = ci_base + 4*grp .. +3 for co = co_base + (lane&15) ----
  const int fr = lane & 15;
  float* slab = p.slab + (int64_t)split * p.Cout * p.Ktot;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int co = co0 + wm * WTM + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int kidx = tap_i * p.Ktap + ci_k + wn * WTN + j * 16 + grp * 4;
      *(f32x4_t*)(slab + (int64_t)co * p.Ktot + kidx) = acc[i][j];
    }
    if (do_colsum && grp == 0) p.colsum[(int64_t)split * p.Cout + co] = acc1[i][0];
  }
}

// ---------------------------------------------------------------------------------------------
// All nine taps of a 3x3 / stride 1 / pad 1 weight gradient in ONE workgroup ("T9").
//
// The tap-per-tile kernel above re-reads the g tile once per (tap, ci block) and the x tile once per tap.  Here a
// workgroup owns a 128 (co) x 64 (ci) block for all nine taps and streams, per 64-pixel K-step,
//   * the g tile once.  The horizontal taps dw = -1 / +1 need it with the rows of the pixels in image column 0 / W-1
//     zeroed (their neighbour in that direction is off the line); a lane's fragment holds 8 consecutive pixels of one
//     channel, so that is one 16-bit field cleared in registers — and only in the K-steps whose 64 pixels touch a line
//     end at all (a workgroup-uniform test);
//   * three x windows, one per vertical tap offset dh: the 66 consecutive input pixels  m0 + dh*W - 1 ... m0 + dh*W + 64
//     (in NHWC the neighbour (h+dh, w+dw) of flattened pixel m is pixel m + dh*W + dw), rows above / below the image
//     read from the zero page; tap (dh, dw) reads its fragments from window dh at a row offset of 1 + dw.
// 43 KB of LDS-DMA per K-step (two stages: 86 KB of LDS, one workgroup per CU) feed 9 x 128 x 64 x 64 MACs: 219 flop/B
// against 51 for the 256x64 single-tap tile.
// Needs W >= 8 (at most one line end per 8 consecutive pixels).  Slab layout and finalize pass are unchanged:
// slab[split][co][tap*Cin + ci].
template <bool F16>
__global__ __launch_bounds__(512) void conv_wgrad9_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKW = 64, BMW = 128, BNW = 64;
  constexpr int RBG = BMW * 2, RBX = BNW * 2;                 // 256 / 128 bytes per row
  constexpr int G_TILE = BKW * RBG;                           // 16 KB
  constexpr int XROWS = 72;                                   // 66 used
  constexpr int X_TILE = XROWS * RBX;                         // 9 KB per dh window
  constexpr int STAGE = G_TILE + 3 * X_TILE;                  // 43 KB
  // Measured on the 200x336 256->256 layer (profiles/r01_wgrad9_bench.log): with the MFMAs fed from registers instead
  // of LDS the kernel takes the same time, and without the DMA 10% less: the loop is MFMA-paced at the clock the chip
  // holds under that load.  A 64 x 16 wave tile (fewer LDS reads) and a three-stage ring changed nothing.
  constexpr int FM = 2, FN = 2, WN = 2, WTM = 16 * FM, WTN = 16 * FN;   // 4 x 2 waves, wave tile 32 co x 32 ci
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ntiles = p.tiles_co * p.tiles_k;
  int split, tile;
  if (p.splitk >= 8) {
    const int xj = blockIdx.x >> 3;
    split = (blockIdx.x & 7) + 8 * (xj / ntiles);
    tile = xj - (xj / ntiles) * ntiles;
  } else {
    split = blockIdx.x / ntiles;
    tile = blockIdx.x - split * ntiles;
  }
  if (split >= p.splitk) return;
  const int tile_co = tile % p.tiles_co, tile_ci = tile / p.tiles_co;
  const int co0 = tile_co * BMW, ci0 = tile_ci * BNW;
  const int m_begin = split * p.Mchunk;
  const int m_end = min(p.M, m_begin + p.Mchunk);
  const int W = p.Wo, H = p.Ho;            // stride 1, pad 1: input and output grids coincide
  const bf16_t* zero = (const bf16_t*)g_zero_page;

  // ---- loader state: 2 g rows per lane; per dh window 1 x row per lane (+ rows 64..71 from wave 0) ----
  const int g_lrow = lane >> 4, g_pc = lane & 15;             // 4 rows x 16 chunks per piece
  const int x_lrow = lane >> 3, x_pc = lane & 7;              // 8 rows x 8 chunks per piece
  const int g_row0 = wave * 4 + g_lrow;                       // + it*32
  const int x_row0 = wave * 8 + x_lrow;                       // + it*64
  const int g_src_el = ((((g_pc >> 1) ^ tr_swz<RBG>(g_row0)) << 1) | (g_pc & 1)) * 8;
  const int x_src_el = ((((x_pc >> 1) ^ tr_swz<RBX>(x_row0)) << 1) | (x_pc & 1)) * 8;
  int xh[2], xw[2];                                           // (row, column) of pixel q0 = m0 + i - 1 of the x rows
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = m_begin + it * 64 + x_row0 - 1;             // may be -1
    const int qq = q < 0 ? q + W * H : q;                     // keep the decode non-negative; q = -1 is masked below
    xw[it] = qq % W;
    xh[it] = (qq / W) % H;
  }

  auto stage_load = [&](int mt, int s) {
    char* sG = smem + s * STAGE;
    char* sX = sG + G_TILE;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int m = mt + it * 32 + g_row0;
      const bf16_t* src = (m < m_end) ? p.g + ((int64_t)m * p.Cout + co0 + g_src_el) : zero + (g_src_el & 127);
      glds16_async(src, sG + (it * 32 + wave * 4) * RBG);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = it * 64 + x_row0;                 // window row
      if (it == 1 && wave != 0) break;                // rows 64..71: one piece, wave 0 only (wave-uniform)
      const int q0 = mt + i - 1;
      const bool in = (i < 66) && q0 >= 0 && q0 < p.M;
      const bf16_t* src = p.x + ((int64_t)q0 * p.Cpix + ci0 + x_src_el);
      const bf16_t* z = zero + (x_src_el & 127);
      char* dst = sX + (it * 64 + wave * 8) * RBX;
      glds16_async((in && xh[it] != 0) ? src - (int64_t)W * p.Cpix : z, dst);               // dh = -1: row above
      glds16_async(in ? src : z, dst + X_TILE);                                             // dh =  0
      glds16_async((in && xh[it] != H - 1) ? src + (int64_t)W * p.Cpix : z, dst + 2 * X_TILE);  // dh = +1: row below
      xw[it] += BKW;
      while (xw[it] >= W) { xw[it] -= W; xh[it] += 1; }
      while (xh[it] >= H) xh[it] -= H;
    }
  };

  // ---- fragment read offsets (ds_read_b64_tr_b16) ----
  const int wm = wave / WN, wn = wave % WN;
  const int grp = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int rrow = 8 * grp + q4;
  int g_off[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int c5 = (wm * WTM + i * 16) >> 4;
    g_off[i] = rrow * RBG + ((c5 ^ tr_swz<RBG>(rrow)) << 5) + pp * 8;
  }
  int x_lo[3][FN], x_hi[3][FN];                               // per row shift 1 + dw = 0, 1, 2
#pragma unroll
  for (int sft = 0; sft < 3; ++sft)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c5 = (wn * WTN + j * 16) >> 4;
      const int rl = rrow + sft, rh = rrow + sft + 4;
      x_lo[sft][j] = rl * RBX + ((c5 ^ tr_swz<RBX>(rl)) << 5) + pp * 8;
      x_hi[sft][j] = rh * RBX + ((c5 ^ tr_swz<RBX>(rh)) << 5) + pp * 8;
    }

  f32x4_t acc[9][FM][FN];
  f32x4_t acc1[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    acc1[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_colsum = (tile_ci == 0) && (wn == 0);
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = f32_to_elem<F16>(1.0f);

  // image column of the first of this lane's 8 fragment pixels (rows 8*grp .. 8*grp+7 of each 32-pixel half step)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int step64 = BKW % W;
  int wb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wb[kk] = (m_begin + kk * 32 + 8 * grp) % W;
  int mw = m_begin % W;                                       // column of the K-step's first pixel (uniform)
  auto keep_mask = [](int e) -> u32x4_t {                     // all ones except the 16 bits of element e (e >= 8: all ones)
    const unsigned keep = (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
    u32x4_t u;
#pragma unroll
    for (int d = 0; d < 4; ++d) u[d] = ((e >> 1) == d) ? keep : 0xFFFFFFFFu;
    return u;
  };

  const int T = (m_end > m_begin) ? ceil_div(m_end - m_begin, BKW) : 0;
  if (T > 0) {
    stage_load(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) stage_load(m_begin + (t + 1) * BKW, (t + 1) & 1);
      const char* sG = smem + (t & 1) * STAGE;
      const char* sX = sG + G_TILE;
      const bool line_end = (mw == 0) || (mw + BKW >= W);     // some pixel of this K-step sits in column 0 or W-1
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8_t gf[3][FM];                                   // [dw + 1]
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const s16x4_t lo = lds_read_tr16(sG + g_off[i] + kk * 32 * RBG);
          const s16x4_t hi = lds_read_tr16(sG + g_off[i] + (kk * 32 + 4) * RBG);
          gf[1][i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          gf[0][i] = gf[1][i];
          gf[2][i] = gf[1][i];
        }
        if (line_end) {   // workgroup-uniform: only these steps need masked copies for the dw = -1 / +1 taps
          const u32x4_t keep_l = keep_mask(wb[kk] == 0 ? 0 : W - wb[kk]);   // column 0 has no left neighbour
          const u32x4_t keep_r = keep_mask(W - 1 - wb[kk]);                 // column W-1 has no right neighbour
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            gf[0][i] = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_t, gf[1][i]) & keep_l);
            gf[2][i] = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_t, gf[1][i]) & keep_r);
          }
        }
        // nine taps, the x fragments of tap t+1 in flight while tap t's MFMAs run
        bf16x8_t xf[2][FN];
        auto read_x = [&](int tap, bf16x8_t (&dst)[FN]) {
          const int dh = tap / 3, sft = tap % 3;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const s16x4_t lo = lds_read_tr16(sX + dh * X_TILE + x_lo[sft][j] + kk * 32 * RBX);
            const s16x4_t hi = lds_read_tr16(sX + dh * X_TILE + x_hi[sft][j] + kk * 32 * RBX);
            dst[j] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        };
        read_x(0, xf[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * FM + 2 * FN, 0);   // the g fragments and tap 0
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) {
            read_x(tap + 1, xf[(tap + 1) & 1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * FN, 0);
          }
          // tap (kh, kw) = (tap / 3, tap % 3): D[ci][co] += sum_m X[m + (kh-1)*W + (kw-1)][ci] * G_kw[m][co]
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[tap][i][j] = mfma16<F16>(xf[tap & 1][j], gf[tap % 3][i], acc[tap][i][j]);
          __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0);
        }
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < FM; ++i) acc1[i] = mfma16<F16>(ones, gf[1][i], acc1[i]);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        wb[kk] += step64;
        if (wb[kk] >= W) wb[kk] -= W;
      }
      mw += step64;
      if (mw >= W) mw -= W;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  const int fr = lane & 15;
  float* slab = p.slab + (int64_t)split * p.Cout * p.Ktot;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int co = co0 + wm * WTM + i * 16 + fr;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int kidx = t * p.Ktap + ci0 + wn * WTN + j * 16 + grp * 4;
        *(f32x4_t*)(slab + (int64_t)co * p.Ktot + kidx) = acc[t][i][j];
      }
    if (do_colsum && grp == 0) p.colsum[(int64_t)split * p.Cout + co] = acc1[i][0];
  }
}

// One block per output channel: reduce the split-K slabs in a fixed order, scale, and produce affine grads.
// The 256 threads are arranged as K4P k-lanes (float4 each) x SL split-lanes so that short rows (Ktot = 64)
// still use the whole block; split-lane partial sums are combined through LDS in lane order (deterministic).
//   map_mode 0: dw index = co*Ktot + k  ([Cout][kh][kw][Cin] = channels_last view of the OIHW grad)
//   map_mode 1: stem, k = (kh*8 + kw)*4 + c  ->  dw[co][c][kh][kw] contiguous (pads dropped)
//   map_mode 4*cpg: grouped conv in block-diagonal form -> dw[co][kh][kw][cpg]
template <bool F16>
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ slab,
                                                             const float* __restrict__ colsum, int splitk,
                                                             int Cout, int Ktot, const bf16_t* __restrict__ w_fwd,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* dw,
                                                             float* dgamma, float* dbeta, float beta,
                                                             int map_mode, int K4P) {
  __shared__ f32x4_t part[256];
  __shared__ float red[4];
  const int co = blockIdx.x;
  const int tid = threadIdx.x;
  const float sc = scale ? scale[co] : 1.f;
  const int K4 = Ktot >> 2;
  const int SL = 256 / K4P;
  const int k4l = tid % K4P, spl = tid / K4P;
  const f32x4_t* slab4 = (const f32x4_t*)slab;
  float dot = 0.f;
  for (int base = 0; base < K4; base += K4P) {
    const int k4 = base + k4l;
    const bool active = k4 < K4;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      int sp = spl;
      for (; sp + 3 * SL < splitk; sp += 4 * SL) {   // 4 independent loads in flight
        const f32x4_t a0 = slab4[((int64_t)sp * Cout + co) * K4 + k4];
        const f32x4_t a1 = slab4[((int64_t)(sp + SL) * Cout + co) * K4 + k4];
        const f32x4_t a2 = slab4[((int64_t)(sp + 2 * SL) * Cout + co) * K4 + k4];
        const f32x4_t a3 = slab4[((int64_t)(sp + 3 * SL) * Cout + co) * K4 + k4];
        s += (a0 + a1) + (a2 + a3);
      }
      for (; sp < splitk; sp += SL) s += slab4[((int64_t)sp * Cout + co) * K4 + k4];
    }
    if (SL > 1) {
      part[tid] = s;
      __syncthreads();
      if (spl == 0) {
        for (int j = 1; j < SL; ++j) s += part[j * K4P + k4l];
      }
      __syncthreads();
    }
    if (active && spl == 0) {
      const int k = k4 * 4;
      const f32x4_t wv = load4_f32<F16>(w_fwd + (int64_t)co * Ktot + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) dot += wv[e] * s[e];
      if (map_mode == 0) {
        f32x4_t* o = (f32x4_t*)(dw + (int64_t)co * Ktot + k);
        f32x4_t v = s * sc;
        if (beta != 0.f) v += *o * beta;
        *o = v;
      } else if (map_mode >= 4) {
        // grouped conv (map_mode = 4 * channels-per-group): k = tap*64 + j is column j of the 64-channel block;
        // only the columns of co's own group are weight entries: dw[co][tap][j % cpg]
        const int cpg = map_mode >> 2;
        const int tap = k >> 6, j0 = k & 63;
        const int gsel = (co & 63) / cpg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + e;
          if (j / cpg != gsel) continue;
          const int64_t oidx = ((int64_t)co * (Ktot >> 6) + tap) * cpg + (j % cpg);
          const float v = sc * s[e];
          dw[oidx] = (beta != 0.f) ? beta * dw[oidx] + v : v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = k + e;
          const int c = kk & 3, kw = (kk >> 2) & 7, kh = kk >> 5;
          if (c == 3 || kw == 7) continue;
          const int64_t oidx = (int64_t)co * 147 + c * 49 + kh * 7 + kw;
          const float v = sc * s[e];
          dw[oidx] = (beta != 0.f) ? beta * dw[oidx] + v : v;
        }
      }
    }
  }
  // block reduce dot; colsum over splits by wave 1
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  if (tid < 64) {
    float cs = 0.f;
    for (int sp = tid; sp < splitk; sp += 64) cs += colsum[(int64_t)sp * Cout + co];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_down(cs, o, 64);
    if (tid == 0) {
      const float d = (red[0] + red[1]) + (red[2] + red[3]);
      if (dbeta) dbeta[co] = (beta != 0.f) ? beta * dbeta[co] + cs : cs;
      if (mean && invstd && dgamma) {
        const float dg = (d - mean[co] * cs) * invstd[co];
        dgamma[co] = (beta != 0.f) ? beta * dgamma[co] + dg : dg;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct WgradPlan { int bmw, bnw, tiles_co, tiles_k, splitk, mchunk, M, Ktot, t9; };

// t9_ok: the layer is a dense 3x3 / stride 1 / pad 1 conv at least 8 pixels wide (the caller checks the geometry)
static WgradPlan plan_wgrad(int M, int Cout, int Ktap, int ntaps, bool grouped = false, bool t9_ok = false) {
  WgradPlan w;
  w.t9 = 0;
  if (t9_ok && !grouped && ntaps == 9 && Cout % 128 == 0 && Ktap % 64 == 0) {
    // measured (profiles/r01_wgrad9_bench.log): alone, 293 -> 187 us on the 200x336 FPN 3x3 and 90 -> 72 us on the
    // 100x168 one, but ~20% slower on the small-M 3x3 layers (few one-per-CU workgroups).  Inside the captured step,
    // where the weight gradients run on side streams next to the dgrad chain, taking it for every eligible layer was
    // still the fastest of three settings in each of three back-to-back rounds (380.9 off / 383.6 large layers only /
    // 386.3 img/s always), so eligibility alone decides.  TDN_WGRAD9=0 turns the kernel off.
    const char* env = getenv("TDN_WGRAD9");   // 0: off, 1: every eligible layer, 2: only layers with >= 256 units of work
    const int mode = env ? atoi(env) : 1;
    const int64_t units = (int64_t)ceil_div(M, 512) * (Cout / 128) * (Ktap / 64);
    w.t9 = mode == 2 ? (units >= 256) : (mode != 0);
  }
  // measured (scripts/wgrad_bench.py): 64-wide ci tiles beat 128; 256-wide co tiles (8 waves) win when Cout allows,
  // except for the small-M 3x3 layers where the extra workgroups of the 128-wide tile matter more
  w.bmw = (Cout % 256 == 0 && !(ntaps > 1 && M < 20000)) ? 256 : ((Cout % 128 == 0) ? 128 : 64);
  w.bnw = (Ktap % 64 == 0) ? 64 : 32;
  if (const char* env = getenv("TDN_WGRAD_TILE")) {   // tuning override: "BMWxBNW" with 64/128 entries
    int a = 0, b = 0;
    if (sscanf(env, "%dx%d", &a, &b) == 2 && (a == 64 || a == 128 || a == 256) && (b == 64 || b == 128 || b == 256) &&
        Cout % a == 0 && Ktap % b == 0) {
      w.bmw = a;
      w.bnw = b;
    }
  }
  if (grouped) { w.bmw = 64; w.bnw = 64; }   // one 64 x 64 diagonal block per tile
  w.tiles_co = Cout / w.bmw;
  w.tiles_k = ntaps * (Ktap / w.bnw);
  if (w.t9) {   // conv_wgrad9_kernel: 128 co x 64 ci x all nine taps per workgroup
    w.bmw = 128; w.bnw = 64;
    w.tiles_co = Cout / 128;
    w.tiles_k = Ktap / 64;
  }
  w.M = M;
  w.Ktot = ntaps * Ktap;
  const int tiles = w.tiles_co * w.tiles_k;
  // Aim for ~512 workgroups (two 64 KB-LDS workgroups fit a CU) but keep each split >= 1024 pixels deep: every
  // workgroup writes a full fp32 tile slab, so short splits turn the kernel (and the finalize pass that re-reads
  // the slabs) into an HBM-bound slab copy.
  int target = w.t9 ? 256 : 512, min_chunk = 512;   // the nine-tap kernel holds 150 KB of LDS: one per CU
  if (const char* env = getenv(w.t9 ? "TDN_WGRAD9_WGS" : "TDN_WGRAD_WGS")) target = atoi(env) > 0 ? atoi(env) : target;
  if (const char* env = getenv("TDN_WGRAD_MINCHUNK")) min_chunk = atoi(env) > 0 ? atoi(env) : min_chunk;
  int splitk = ceil_div(target, tiles);
  const int max_split = ceil_div(M, min_chunk) > 0 ? ceil_div(M, min_chunk) : 1;
  if (splitk > max_split) splitk = max_split;
  if (splitk > 256) splitk = 256;
  if (splitk < 1) splitk = 1;
  // splits are dealt round-robin to the 8 XCD labels (kernel's work map): keep the per-XCD load even
  if (splitk >= 8) splitk = (splitk + 4) / 8 * 8;
  int mchunk = ceil_div(ceil_div(M, splitk), 64) * 64;
  splitk = ceil_div(M, mchunk);
  w.splitk = splitk;
  w.mchunk = mchunk;
  return w;
}

static int64_t wgrad_ws_bytes(const WgradPlan& w, int Cout) {
  return ((int64_t)w.splitk * Cout * w.Ktot + (int64_t)w.splitk * Cout) * 4 + 256;
}

template <int BMW, int BNW, int WM, int WN, bool F16>
static int launch_wgrad_t(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)64 * (BMW + BNW) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_kernel<BMW, BNW, WM, WN, F16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", (int)lds, hipGetErrorString(e));
    attr_set = true;
  }
  const int slots = p.splitk >= 8 ? 8 * ((p.splitk + 7) / 8) : p.splitk;   // see the kernel's work map
  dim3 grid(slots * p.tiles_co * p.tiles_k, 1, 1), block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL((conv_wgrad_kernel<BMW, BNW, WM, WN, F16>), grid, block, lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <int BMW, int BNW, int WM = 2, int WN = 2>
static int launch_wgrad(const WgradParams& p, hipStream_t stream, int dtype) {
  return dtype == TDN_F16 ? launch_wgrad_t<BMW, BNW, WM, WN, true>(p, stream)
                          : launch_wgrad_t<BMW, BNW, WM, WN, false>(p, stream);
}

template <bool F16>
static int launch_wgrad9_t(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(64 * 256 + 3 * 72 * 128);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad9_kernel<F16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", (int)lds, hipGetErrorString(e));
    attr_set = true;
  }
  const int slots = p.splitk >= 8 ? 8 * ((p.splitk + 7) / 8) : p.splitk;
  dim3 grid(slots * p.tiles_co * p.tiles_k, 1, 1), block(512, 1, 1);
  hipLaunchKernelGGL((conv_wgrad9_kernel<F16>), grid, block, lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

static int launch_wgrad9(const WgradParams& p, hipStream_t stream, int dtype) {
  return dtype == TDN_F16 ? launch_wgrad9_t<true>(p, stream) : launch_wgrad9_t<false>(p, stream);
}

static int run_wgrad(WgradParams& p, const WgradPlan& w, const void* w_fwd, const float* scale,
                     const float* mean, const float* invstd, float* dw, float* dgamma, float* dbeta,
                     float beta, void* workspace, int64_t workspace_bytes, int map_mode, int dtype,
                     hipStream_t stream) {
  TDN_CHECK(workspace_bytes >= wgrad_ws_bytes(w, p.Cout), "wgrad workspace too small: %lld < %lld",
            (long long)workspace_bytes, (long long)wgrad_ws_bytes(w, p.Cout));
  TDN_CHECK(((uintptr_t)workspace & 15) == 0, "wgrad workspace must be 16-byte aligned");
  p.slab = (float*)workspace;
  p.colsum = p.slab + (int64_t)w.splitk * p.Cout * w.Ktot;
  p.M = w.M; p.Mchunk = w.mchunk; p.splitk = w.splitk; p.Ktot = w.Ktot;
  p.tiles_co = w.tiles_co; p.tiles_k = w.tiles_k;
  int rc;
  if (w.t9) rc = launch_wgrad9(p, stream, dtype);
  else if (w.bmw == 256 && w.bnw == 256) rc = launch_wgrad<256, 256, 4, 4>(p, stream, dtype);
  else if (w.bmw == 256 && w.bnw == 128) rc = launch_wgrad<256, 128, 4, 4>(p, stream, dtype);
  else if (w.bmw == 128 && w.bnw == 256) rc = launch_wgrad<128, 256, 4, 4>(p, stream, dtype);
  else if (w.bmw == 256 && w.bnw == 64) rc = launch_wgrad<256, 64, 4, 2>(p, stream, dtype);
  else if (w.bmw == 128 && w.bnw == 128) rc = launch_wgrad<128, 128>(p, stream, dtype);
  else if (w.bmw == 128 && w.bnw == 64) rc = launch_wgrad<128, 64>(p, stream, dtype);
  else if (w.bmw == 64 && w.bnw == 128) rc = launch_wgrad<64, 128>(p, stream, dtype);
  else if (w.bmw == 64 && w.bnw == 64) rc = launch_wgrad<64, 64>(p, stream, dtype);
  else if (w.bmw == 64 && w.bnw == 32) rc = launch_wgrad<64, 32>(p, stream, dtype);
  else { tdn_set_error("wgrad: no kernel for tile %dx%d", w.bmw, w.bnw); return -1; }
  if (rc) return rc;
  int k4p = 1;
  while (k4p < (w.Ktot >> 2) && k4p < 256) k4p <<= 1;
  if (dtype == TDN_F16)
    hipLaunchKernelGGL(wgrad_finalize_kernel<true>, dim3(p.Cout), dim3(256), 0, stream, p.slab, p.colsum, w.splitk,
                       p.Cout, w.Ktot, (const bf16_t*)w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, map_mode,
                       k4p);
  else
    hipLaunchKernelGGL(wgrad_finalize_kernel<false>, dim3(p.Cout), dim3(256), 0, stream, p.slab, p.colsum, w.splitk,
                       p.Cout, w.Ktot, (const bf16_t*)w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, map_mode,
                       k4p);
  TDN_LAUNCH_CHECK();
  return 0;
}

// "same" convs: for k = 3 the padding is the dilation (conv3x3_group: padding = dilation, layers.py:20-32)
static int conv_dil(int k, int pad) { return k == 3 ? pad : 1; }
static int conv_out(int H, int k, int stride, int pad) {
  return (H + 2 * pad - (conv_dil(k, pad) * (k - 1) + 1)) / stride + 1;
}

extern "C" int64_t tdn_conv2d_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int k, int stride,
                                               int pad) {
  const int Ho = conv_out(H, k, stride, pad), Wo = conv_out(W, k, stride, pad);
  const WgradPlan w = plan_wgrad(N * Ho * Wo, Cout, Cin, k * k, false, k == 3 && stride == 1 && pad == 1 && W >= 8);
  return wgrad_ws_bytes(w, Cout);
}

int tdn_wgrad_plan(int N, int H, int W, int Cin, int Cout, int k, int stride, int pad, int32_t* o) {
  const int Ho = conv_out(H, k, stride, pad), Wo = conv_out(W, k, stride, pad);
  const WgradPlan w = plan_wgrad(N * Ho * Wo, Cout, Cin, k * k, false, k == 3 && stride == 1 && pad == 1 && W >= 8);
  o[0] = Cout; o[1] = w.Ktot; o[2] = w.M; o[3] = w.bmw; o[4] = w.bnw; o[5] = 64;
  o[6] = w.tiles_co * w.tiles_k; o[7] = w.splitk; o[8] = 1; o[9] = 1; o[10] = k * k; o[11] = w.splitk;
  o[12] = w.mchunk; o[13] = Ho; o[14] = Wo; o[15] = w.M;
  return 0;
}

extern "C" int tdn_conv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale,
                                const float* mean, const float* invstd, float* dw, float* dgamma,
                                float* dbeta, float beta, int N, int H, int W, int Cin, int Cout, int k,
                                int stride, int pad, void* workspace, int64_t workspace_bytes, int dtype,
                                void* stream) {
  TDN_CHECK(dtype == TDN_BF16 || dtype == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", dtype);
  TDN_CHECK(x && g && w_fwd && dw && workspace, "tdn_conv2d_wgrad: NULL pointer");
  TDN_CHECK(k == 1 || k == 3, "kernel size %d not supported", k);
  TDN_CHECK(stride == 1 || stride == 2, "stride %d not supported", stride);
  TDN_CHECK((k == 1 && pad == 0) || (k == 3 && pad >= 1 && pad <= 32), "pad %d not supported for k=%d", pad, k);
  TDN_CHECK(Cin % 64 == 0 && Cout % 64 == 0, "channels must be multiples of 64 (Cin=%d Cout=%d)", Cin, Cout);
  const int Ho = conv_out(H, k, stride, pad), Wo = conv_out(W, k, stride, pad);
  const WgradPlan w = plan_wgrad(N * Ho * Wo, Cout, Cin, k * k, false, k == 3 && stride == 1 && pad == 1 && W >= 8);
  WgradParams p;
  p.x = (const bf16_t*)x; p.g = (const bf16_t*)g; p.grouped = 0;
  p.Hin = H; p.Win = W; p.Cpix = Cin; p.Ktap = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.sa = stride;
  p.ntaps = k * k;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) p.taps[kh * k + kw] = (kh * conv_dil(k, pad) - pad + 64) | ((kw * conv_dil(k, pad) - pad + 64) << 8);
  return run_wgrad(p, w, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, workspace, workspace_bytes, 0,
                   dtype, (hipStream_t)stream);
}

// Grouped conv weight gradient (see tdn_gconv2d_fwd): per 64-channel block a dense 64 x (taps * 64) product, of which
// the finalize pass keeps each output channel's own group: dw fp32 [C][k][k][cpg].
extern "C" int64_t tdn_gconv2d_wgrad_workspace(int N, int H, int W, int C, int groups, int k, int stride, int pad) {
  const int Ho = conv_out(H, k, stride, pad), Wo = conv_out(W, k, stride, pad);
  const WgradPlan w = plan_wgrad(N * Ho * Wo, C, 64, k * k, true);
  return wgrad_ws_bytes(w, C);
}

extern "C" int tdn_gconv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale,
                                 const float* mean, const float* invstd, float* dw, float* dgamma, float* dbeta,
                                 float beta, int N, int H, int W, int C, int groups, int k, int stride, int pad,
                                 void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK(dtype == TDN_BF16 || dtype == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", dtype);
  TDN_CHECK(x && g && w_fwd && dw && workspace, "tdn_gconv2d_wgrad: NULL pointer");
  TDN_CHECK(k == 1 || k == 3, "kernel size %d not supported", k);
  TDN_CHECK(stride == 1 || stride == 2, "stride %d not supported", stride);
  TDN_CHECK((k == 1 && pad == 0) || (k == 3 && pad >= 1 && pad <= 32), "pad %d not supported for k=%d", pad, k);
  TDN_CHECK(groups > 0 && C % groups == 0 && C % 64 == 0 && (C / groups) <= 64 && 64 % (C / groups) == 0,
            "grouped conv: need C %% 64 == 0 and channels per group dividing 64 (C=%d, groups=%d)", C, groups);
  const int Ho = conv_out(H, k, stride, pad), Wo = conv_out(W, k, stride, pad);
  const WgradPlan w = plan_wgrad(N * Ho * Wo, C, 64, k * k, true);
  WgradParams p;
  p.x = (const bf16_t*)x; p.g = (const bf16_t*)g; p.grouped = 1;
  p.Hin = H; p.Win = W; p.Cpix = C; p.Ktap = 64; p.Ho = Ho; p.Wo = Wo; p.Cout = C; p.sa = stride;
  p.ntaps = k * k;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) p.taps[kh * k + kw] = (kh * conv_dil(k, pad) - pad + 64) | ((kw * conv_dil(k, pad) - pad + 64) << 8);
  return run_wgrad(p, w, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, workspace, workspace_bytes,
                   4 * (C / groups), dtype, (hipStream_t)stream);
}

extern "C" int64_t tdn_stem_conv_wgrad_workspace(int N, int H, int W, int Cout) {
  const WgradPlan w = plan_wgrad(N * (H / 2) * (W / 2), Cout, 32, 7);
  return wgrad_ws_bytes(w, Cout);
}

extern "C" int tdn_stem_conv_wgrad(const void* xp, const void* g, const void* w_stem, const float* scale,
                                   const float* mean, const float* invstd, float* dw, float* dgamma,
                                   float* dbeta, float beta, int N, int H, int W, int Cout, void* workspace,
                                   int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK(dtype == TDN_BF16 || dtype == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", dtype);
  TDN_CHECK(xp && g && w_stem && dw && workspace, "tdn_stem_conv_wgrad: NULL pointer");
  TDN_CHECK(H % 2 == 0 && W % 2 == 0 && Cout % 64 == 0, "stem wgrad: bad shape");
  const int Ho = H / 2, Wo = W / 2;
  const WgradPlan w = plan_wgrad(N * Ho * Wo, Cout, 32, 7);
  WgradParams p;
  p.x = (const bf16_t*)xp; p.g = (const bf16_t*)g; p.grouped = 0;
  p.Hin = H + 6; p.Win = W + 8; p.Cpix = 4; p.Ktap = 32; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.sa = 2;
  p.ntaps = 7;
  for (int kh = 0; kh < 7; ++kh) p.taps[kh] = (kh + 64) | ((0 + 64) << 8);
  return run_wgrad(p, w, w_stem, scale, mean, invstd, dw, dgamma, dbeta, beta, workspace, workspace_bytes, 1,
                   dtype, (hipStream_t)stream);
}
