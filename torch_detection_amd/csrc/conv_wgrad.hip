// Weight-gradient GEMMs on MFMA (gfx950), launched per GROUP of layers, + one grouped finalize pass.
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
//
// Grouping.  The weight gradients of one layer are a small GEMM output (64x64 ... 2048x512) under a very long
// reduction (2,100 ... 268,800 pixels): alone, a layer can only fill 256 CUs by cutting the pixel range into many
// splits, each of which writes a full fp32 copy of its tile that a second kernel has to read back.  A GROUP of layers
// (a ResNet stage, the FPN) fills the chip together, so the splits are sized for the group (tdn_wgrad_group): the
// work list of one launch spans all members (descriptors travel in the kernel-argument block), members whose
// reduction fits one workgroup per tile write their gradient directly, and the remaining slabs, the BN gamma / beta
// and bias gradients of EVERY member are reduced by one finalize launch, in a fixed order (deterministic).
#include "common.h"
#include <limits.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include <math.h>
#include <mutex>
#include <unordered_map>
#include <vector>

constexpr int WG_MAXI = 26;    // members per gradient launch: the kernel-argument block must stay under 4 KB
constexpr int FIN_MAXI = 30;   // members per finalize launch

// One member of a grouped launch, as the kernels see it.
struct WgItem {
  const bf16_t* x;
  const bf16_t* g;
  const bf16_t* w_fwd;   // direct members with BN: read for the partial sum_k w * G
  const float* scale;    // direct members: BN scale folded into the stored gradient (NULL = 1)
  float* out;            // slab [splitk][Cout][Ktot]; direct members: dw itself ([Cout][Ktot])
  float* colsum;         // [splitk][Cout]
  float* dotpart;        // direct members with BN: [tiles_k][Cout] partial sum_k w * G per K tile, else NULL
  int Hin, Win, Cpix, Ktap;   // x geometry (pixel stride Cpix elements, Ktap elements consumed per tap)
  int Ho, Wo, Cout, sa;
  int M, Mchunk, splitk;
  int ntaps, Ktot;
  int tiles_co, tiles_k;      // tiles over Cout, tiles over (tap, ci)
  int tapgen;                 // k | pad << 8 | dilation << 16 | stem << 24: tap t -> (dh, dw)
  int grouped;                // block-diagonal grouped conv: the ci block of a tile is its co block (64 x 64 tiles)
  int direct;
  float beta;
  int reserved;
};

struct WgGroup {
  int nitems, reserved;
  int blk_start[WG_MAXI + 2];   // first workgroup of member i (a multiple of 8); INT_MAX past the last member
  WgItem it[WG_MAXI];
};
static_assert(sizeof(WgGroup) <= 4096, "kernel-argument block too large");

// tap t of a member -> displacement of the input pixel: (kh * dil - pad, kw * dil - pad); stem: one "tap" per kernel
// row (8 pixels x 4 channels contiguous)
__device__ __forceinline__ void wg_tap(int tapgen, int t, int& dh, int& dw) {
  const int k = tapgen & 0xff, pad = (tapgen >> 8) & 0xff, dil = (tapgen >> 16) & 0xff;
  if (tapgen >> 24) { dh = t; dw = 0; return; }
  const int kh = t / k, kw = t - kh * k;
  dh = kh * dil - pad;
  dw = kw * dil - pad;
}

// XCD-aware work map inside a member: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch; speed only; the
// member's first workgroup is a multiple of 8).  XCD x owns the pixel splits s = x (mod 8) and walks all tiles of one
// split before the next, so the ~ntiles workgroups that stream the same g / x pixel range run together on ONE L2
// instead of being dealt over all eight.
__device__ __forceinline__ bool wg_work(int bid, int splitk, int ntiles, int& split, int& tile) {
  if (splitk >= 8) {
    const int xj = bid >> 3;
    split = (bid & 7) + 8 * (xj / ntiles);
    tile = xj - (xj / ntiles) * ntiles;
  } else {   // too few splits to feed 8 XCDs that way: plain order, tile fastest
    split = bid / ntiles;
    tile = bid - split * ntiles;
  }
  return split < splitk;
}
static int wg_blocks(int splitk, int ntiles) {
  const int slots = splitk >= 8 ? 8 * ((splitk + 7) / 8) : splitk;
  return (slots * ntiles + 7) & ~7;
}

// 32-byte-chunk XOR swizzle for tr-read tiles, by row bytes.
template <int RB>
__device__ __forceinline__ int tr_swz(int row) {
  if constexpr (RB >= 256) return (row & 3) | (((row >> 3) & 1) << 2);
  else if constexpr (RB == 128) return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
  else return (row >> 3) & 1;
}

template <int N>
__device__ __forceinline__ void wg_wait_vm_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

#ifdef TDN_TRACE_BUILD
// libtdn_trace.so only (scripts/wgrad_stamps.py): 8 x 8-byte stamps per workgroup of the tap-per-tile kernel — cycle
// counter at entry / first data landed / loop end / stores issued, the 100 MHz wall clock at entry and exit, the number
// of K-steps, and the hardware id (XCC, SE, CU) the workgroup ran on.
__device__ unsigned long long* g_wg_trace = nullptr;
__device__ int g_wg_trace_cap = 0;
#define WG_STAMP(slot, val)                                                                        \
  do {                                                                                             \
    if (g_wg_trace && tid == 0 && (int)blockIdx.x < g_wg_trace_cap)                                \
      g_wg_trace[(size_t)blockIdx.x * 8 + (slot)] = (unsigned long long)(val);                     \
  } while (0)
extern "C" int tdn_debug_wgrad_trace(void* buf, int nwg_cap) {
  unsigned long long* b = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &b, sizeof(b));
  TDN_CHECK(e == hipSuccess, "tdn_debug_wgrad_trace: %s", hipGetErrorString(e));
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_cap), &nwg_cap, sizeof(nwg_cap));
  TDN_CHECK(e == hipSuccess, "tdn_debug_wgrad_trace: %s", hipGetErrorString(e));
  return 0;
}
#else
#define WG_STAMP(slot, val) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// One tap per tile: a workgroup owns BMW (co) x BNW (ci of one tap) of one member for one pixel split.
// NST-deep LDS ring filled by LDS-DMA: while K-step t is multiplied the loads of steps t+1 .. t+NST-2 stay in
// flight (counted vmcnt, one s_barrier per K-step).
// ---------------------------------------------------------------------------------------------
template <int BMW /*co*/, int BNW /*ci*/, int WM, int WN, int NST, bool F16>
__global__ __launch_bounds__(WM * WN * 64) void conv_wgrad_group_kernel(const WgGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKW = 64;                       // pixels per stage
  constexpr int RBG = BMW * 2, RBX = BNW * 2;   // row bytes
  constexpr int G_BYTES = BKW * RBG, X_BYTES = BKW * RBX, STAGE = G_BYTES + X_BYTES;
  constexpr int RPIG = 1024 / RBG, RPIX = 1024 / RBX;   // rows per wave-instruction
  constexpr int NW = WM * WN;
  constexpr int G_IT = BKW / (RPIG * NW), X_IT = BKW / (RPIX * NW);
  static_assert(G_IT >= 1 && X_IT >= 1, "every wave must issue the same number of loads per stage (counted vmcnt)");
  constexpr int LOADS = G_IT + X_IT;
  static_assert(LOADS * (NST - 2) < 64, "vmcnt immediate out of range");
  constexpr int WTM = BMW / WM, WTN = BNW / WN, FM = WTM / 16, FN = WTN / 16;  // per-wave co / ci frags
  static_assert(FM >= 1 && FN >= 1, "tile too small");
  // the per-lane swizzle constants assume the row offset between a lane's loads keeps row bits 0..3
  static_assert((RPIG * NW) % 16 == 0 && (RPIX * NW) % 16 == 0, "loader round must be a multiple of 16 rows");
  static_assert(WN * BMW * 4 <= NST * STAGE, "cross-wave reduction buffer must fit the ring");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which member, which (split, tile) ----
  int idx = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXI; ++i) idx += ((int)blockIdx.x >= grp.blk_start[i]) ? 1 : 0;
  const WgItem& p = grp.it[idx];
  const int ntiles = p.tiles_co * p.tiles_k;
  int split, tile;
  if (!wg_work((int)blockIdx.x - grp.blk_start[idx], p.splitk, ntiles, split, tile)) return;
  WG_STAMP(0, __builtin_readcyclecounter());
  WG_STAMP(4, __builtin_amdgcn_s_memrealtime());
  const int tile_co = tile % p.tiles_co, tile_k = tile / p.tiles_co;
  const int co0 = tile_co * BMW;
  const int kt_per_tap = p.Ktap / BNW;
  const int tap_i = tile_k / kt_per_tap;
  const int ci_k = (tile_k - tap_i * kt_per_tap) * BNW;   // column offset inside the tap's K range (slab index)
  const int ci0 = p.grouped ? co0 : ci_k;                 // channel offset in x
  int dh, dw;
  wg_tap(p.tapgen, tap_i, dh, dw);
  const int m_begin = split * p.Mchunk;
  const int m_end = min(p.M, m_begin + p.Mchunk);
  const int Cout = p.Cout, Cpix = p.Cpix, Hin = p.Hin, Win = p.Win, sa = p.sa, Wo = p.Wo, Ho = p.Ho;
  const bf16_t* const gx = p.x;
  const bf16_t* const gg = p.g;

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
  const int HoWo = Ho * Wo;
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    const int m = m_begin + it * (RPIX * NW) + x_row0;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    ximg[it] = img;
    xa[it] = rem / Wo;
    xb[it] = rem - xa[it] * Wo;
  }

  // LDS-DMA of the 64 pixels from mt into ring slot s; past the end of the split: the zero page (keeps the vmcnt
  // bookkeeping uniform)
  auto stage_load = [&](int mt, int s) {
    char* sG = smem + s * STAGE;
    char* sX = sG + G_BYTES;
#pragma unroll
    for (int it = 0; it < G_IT; ++it) {
      const int r = it * (RPIG * NW) + g_row0;
      const int m = mt + r;
      const bf16_t* src = (m < m_end) ? gg + ((int64_t)m * Cout + co0 + g_src_el) : zero + (g_src_el & 127);
      glds16_async(src, sG + (it * (RPIG * NW) + wave * RPIG) * RBG);
    }
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int r = it * (RPIX * NW) + x_row0;
      const int m = mt + r;
      const int h = xa[it] * sa + dh, w = xb[it] * sa + dw;
      const bool ok = (m < m_end) && ((unsigned)h < (unsigned)Hin) && ((unsigned)w < (unsigned)Win);
      const bf16_t* src = ok ? gx + (((int64_t)(ximg[it] * Hin + h) * Win + w) * Cpix + ci0 + x_src_el)
                             : zero + (x_src_el & 127);
      glds16_async(src, sX + (it * (RPIX * NW) + wave * RPIX) * RBX);
      // advance to the next stage's pixel
      xb[it] += BKW;
      while (xb[it] >= Wo) { xb[it] -= Wo; xa[it] += 1; }
      while (xa[it] >= Ho) { xa[it] -= Ho; ximg[it] += 1; }
    }
  };

  // ---- fragment reader constants (ds_read_b64_tr_b16) ----
  const int wm = wave / WN, wn = wave % WN;
  const int grp4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int rrow = 8 * grp4 + q;                 // + kk*32 + 4*half
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
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) stage_load(m_begin + s * BKW, s);
    int slot = 0, fill = NST - 1;
    for (int t = 0; t < T; ++t) {
      wg_wait_vm_and_barrier<LOADS * (NST - 2)>();   // K-step t has landed for every wave; slot (t-1) is free
      if (t == 0) WG_STAMP(1, __builtin_readcyclecounter());
      stage_load(m_begin + (t + NST - 1) * BKW, fill);
      const char* sG = smem + slot * STAGE;
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
      slot = (slot + 1 == NST) ? 0 : slot + 1;
      fill = (fill + 1 == NST) ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy tail loads still target the ring
  }
  WG_STAMP(2, __builtin_readcyclecounter());
  WG_STAMP(6, ((unsigned long long)T << 32) | (unsigned)idx);

  // ---- store: lane holds ci = ci_base + 4*grp4 .. +3 for co = co_base + (lane&15) ----
  const int fr = lane & 15;
  const int Ktot = p.Ktot;
  if (!p.direct) {
    float* slab = p.out + (int64_t)split * Cout * Ktot;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int co = co0 + wm * WTM + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int kidx = tap_i * p.Ktap + ci_k + wn * WTN + j * 16 + grp4 * 4;
        *(f32x4_t*)(slab + (int64_t)co * Ktot + kidx) = acc[i][j];
      }
      if (do_colsum && grp4 == 0) p.colsum[(int64_t)split * Cout + co] = acc1[i][0];
    }
#ifdef TDN_TRACE_BUILD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(3, __builtin_readcyclecounter());
    WG_STAMP(5, __builtin_amdgcn_s_memrealtime());
    WG_STAMP(7, (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |        // HW_ID
                    ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32));  // XCC_ID
#endif
    return;
  }
  // ---- direct member (one split): the gradient itself, BN scale folded, + this tile's share of sum_k w * G ----
  const float beta = p.beta;
  const bool want_dot = p.dotpart != nullptr;
  float dpart[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int co = co0 + wm * WTM + i * 16 + fr;
    const float sc = p.scale ? p.scale[co] : 1.f;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int kidx = tap_i * p.Ktap + ci_k + wn * WTN + j * 16 + grp4 * 4;
      const f32x4_t a = acc[i][j];
      if (want_dot) {
        const f32x4_t wv = load4_f32<F16>(p.w_fwd + (int64_t)co * Ktot + kidx);
        d += ((wv[0] * a[0] + wv[1] * a[1]) + wv[2] * a[2]) + wv[3] * a[3];
      }
      f32x4_t* o = (f32x4_t*)(p.out + (int64_t)co * Ktot + kidx);
      f32x4_t v = a * sc;
      if (beta != 0.f) v += *o * beta;
      *o = v;
    }
    dpart[i] = d;
    if (do_colsum && grp4 == 0) p.colsum[co] = acc1[i][0];
  }
  if (want_dot) {
    float* red = (float*)smem;
    __syncthreads();   // every wave is done reading the ring
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float d = dpart[i];
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      if (grp4 == 0) red[wn * BMW + wm * WTM + i * 16 + fr] = d;
    }
    __syncthreads();
    for (int c = tid; c < BMW; c += NW * 64) {
      float s = red[c];
#pragma unroll
      for (int w = 1; w < WN; ++w) s += red[w * BMW + c];
      p.dotpart[(int64_t)tile_k * Cout + co0 + c] = s;
    }
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
// Needs W >= 8 (at most one line end per 8 consecutive pixels).  Slab layout: slab[split][co][tap*Cin + ci].
// WMR = wave rows: 4 -> 128 output channels per workgroup (8 waves), 2 -> 64 (4 waves: the 64-channel 3x3 convs of
// layer1, whose tap-per-tile launches were most of the step's last stretch).
template <bool F16, int WMR = 4>
__global__ __launch_bounds__(WMR * 128) void conv_wgrad9_group_kernel(const WgGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKW = 64, BMW = 32 * WMR, BNW = 64;
  constexpr int NWV = WMR * 2;                                // waves
  constexpr int RBG = BMW * 2, RBX = BNW * 2;                 // 256 (128) / 128 bytes per row
  constexpr int CPRG = RBG / 16, RPIG = 64 / CPRG;            // 16-byte chunks per g row, g rows per wave piece
  constexpr int XIT = 64 / (NWV * 8);                         // passes over the 64 window rows (+ one for rows 64..71)
  static_assert(RPIG * NWV == 32 && XIT >= 1, "loader layout");
  constexpr int G_TILE = BKW * RBG;                           // 16 (8) KB
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

  int idx = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXI; ++i) idx += ((int)blockIdx.x >= grp.blk_start[i]) ? 1 : 0;
  const WgItem& p = grp.it[idx];
  const int ntiles = p.tiles_co * p.tiles_k;
  int split, tile;
  if (!wg_work((int)blockIdx.x - grp.blk_start[idx], p.splitk, ntiles, split, tile)) return;
  const int tile_co = tile % p.tiles_co, tile_ci = tile / p.tiles_co;
  const int co0 = tile_co * BMW, ci0 = tile_ci * BNW;
  const int m_begin = split * p.Mchunk;
  const int pM = p.M;
  const int m_end = min(pM, m_begin + p.Mchunk);
  const int W = p.Wo, H = p.Ho;            // stride 1, pad 1: input and output grids coincide
  const int Cout = p.Cout, Cpix = p.Cpix;
  const bf16_t* const gx = p.x;
  const bf16_t* const gg = p.g;
  const bf16_t* zero = (const bf16_t*)g_zero_page;

  // ---- loader state: 2 g rows per lane; per dh window 1 x row per lane (+ rows 64..71 from wave 0) ----
  const int g_lrow = lane / CPRG, g_pc = lane % CPRG;         // RPIG rows x CPRG chunks per piece
  const int x_lrow = lane >> 3, x_pc = lane & 7;              // 8 rows x 8 chunks per piece
  const int g_row0 = wave * RPIG + g_lrow;                    // + it*32
  const int x_row0 = wave * 8 + x_lrow;                       // + it*(NWV*8); the last pass (rows 64..71): wave 0 only
  const int g_src_el = ((((g_pc >> 1) ^ tr_swz<RBG>(g_row0)) << 1) | (g_pc & 1)) * 8;
  const int x_src_el = ((((x_pc >> 1) ^ tr_swz<RBX>(x_row0)) << 1) | (x_pc & 1)) * 8;
  int xh[XIT + 1], xw[XIT + 1];                               // (row, column) of pixel q0 = m0 + i - 1 of the x rows
#pragma unroll
  for (int it = 0; it <= XIT; ++it) {
    const int q = m_begin + it * (NWV * 8) + x_row0 - 1;      // may be -1
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
      const bf16_t* src = (m < m_end) ? gg + ((int64_t)m * Cout + co0 + g_src_el) : zero + (g_src_el & 127);
      glds16_async(src, sG + (it * 32 + wave * RPIG) * RBG);
    }
#pragma unroll
    for (int it = 0; it <= XIT; ++it) {
      const int i = it * (NWV * 8) + x_row0;          // window row
      if (it == XIT && wave != 0) break;              // rows 64..71: one piece, wave 0 only (wave-uniform)
      const int q0 = mt + i - 1;
      const bool in = (i < 66) && q0 >= 0 && q0 < pM;
      const bf16_t* src = gx + ((int64_t)q0 * Cpix + ci0 + x_src_el);
      const bf16_t* z = zero + (x_src_el & 127);
      char* dst = sX + (it * (NWV * 8) + wave * 8) * RBX;
      glds16_async((in && xh[it] != 0) ? src - (int64_t)W * Cpix : z, dst);               // dh = -1: row above
      glds16_async(in ? src : z, dst + X_TILE);                                           // dh =  0
      glds16_async((in && xh[it] != H - 1) ? src + (int64_t)W * Cpix : z, dst + 2 * X_TILE);  // dh = +1: row below
      xw[it] += BKW;
      while (xw[it] >= W) { xw[it] -= W; xh[it] += 1; }
      while (xh[it] >= H) xh[it] -= H;
    }
  };

  // ---- fragment read offsets (ds_read_b64_tr_b16) ----
  const int wm = wave / WN, wn = wave % WN;
  const int grp4 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int rrow = 8 * grp4 + q4;
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

  // image column of the first of this lane's 8 fragment pixels (rows 8*grp4 .. 8*grp4+7 of each 32-pixel half step)
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int step64 = BKW % W;
  int wb[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wb[kk] = (m_begin + kk * 32 + 8 * grp4) % W;
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
    // The two waves of a SIMD (w and w + NWV / 2) run the same program between the same barriers: left alone they
    // issue their LDS-DMA pieces (each holds the wave's issue slot for ~100-250 cycles) at the same time and their
    // MFMAs at the same time.  The second half of the waves issues the next K-step's pieces between the two 32-pixel
    // sub-steps instead, so that on every SIMD one wave's DMA issue runs beside the other's MFMAs (TDN_WGRAD9_STAGGER=0:
    // all waves at the top).
    const bool late_issue = grp.reserved != 0 && wave >= NWV / 2;
    for (int t = 0; t < T; ++t) {
      if (!late_issue && t + 1 < T) stage_load(m_begin + (t + 1) * BKW, (t + 1) & 1);
      const char* sG = smem + (t & 1) * STAGE;
      const char* sX = sG + G_TILE;
      const bool line_end = (mw == 0) || (mw + BKW >= W);     // some pixel of this K-step sits in column 0 or W-1
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk == 1 && late_issue && t + 1 < T) stage_load(m_begin + (t + 1) * BKW, (t + 1) & 1);
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
          const int dhi = tap / 3, sft = tap % 3;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const s16x4_t lo = lds_read_tr16(sX + dhi * X_TILE + x_lo[sft][j] + kk * 32 * RBX);
            const s16x4_t hi = lds_read_tr16(sX + dhi * X_TILE + x_hi[sft][j] + kk * 32 * RBX);
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
  const int Ktot = p.Ktot, Ktap = p.Ktap;
  if (!p.direct) {
    float* slab = p.out + (int64_t)split * Cout * Ktot;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int co = co0 + wm * WTM + i * 16 + fr;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int kidx = t * Ktap + ci0 + wn * WTN + j * 16 + grp4 * 4;
          *(f32x4_t*)(slab + (int64_t)co * Ktot + kidx) = acc[t][i][j];
        }
      if (do_colsum && grp4 == 0) p.colsum[(int64_t)split * Cout + co] = acc1[i][0];
    }
    return;
  }
  // ---- direct member: see conv_wgrad_group_kernel ----
  const float beta = p.beta;
  const bool want_dot = p.dotpart != nullptr;
  float dpart[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int co = co0 + wm * WTM + i * 16 + fr;
    const float sc = p.scale ? p.scale[co] : 1.f;
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int kidx = t * Ktap + ci0 + wn * WTN + j * 16 + grp4 * 4;
        const f32x4_t a = acc[t][i][j];
        if (want_dot) {
          const f32x4_t wv = load4_f32<F16>(p.w_fwd + (int64_t)co * Ktot + kidx);
          d += ((wv[0] * a[0] + wv[1] * a[1]) + wv[2] * a[2]) + wv[3] * a[3];
        }
        f32x4_t* o = (f32x4_t*)(p.out + (int64_t)co * Ktot + kidx);
        f32x4_t v = a * sc;
        if (beta != 0.f) v += *o * beta;
        *o = v;
      }
    dpart[i] = d;
    if (do_colsum && grp4 == 0) p.colsum[co] = acc1[i][0];
  }
  if (want_dot) {
    float* red = (float*)smem;   // the loop ended on a barrier: the ring is idle
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float d = dpart[i];
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      if (grp4 == 0) red[wn * BMW + wm * WTM + i * 16 + fr] = d;
    }
    __syncthreads();
    for (int c = tid; c < BMW; c += NWV * 64) {
      float s = red[c];
#pragma unroll
      for (int w = 1; w < WN; ++w) s += red[w * BMW + c];
      p.dotpart[(int64_t)tile_ci * Cout + co0 + c] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Grouped finalize: per member and output channel, reduce the split-K slabs in a fixed order, scale, and produce the
// affine gradients; direct members only have their per-tile partial sums and column sums combined.
// ---------------------------------------------------------------------------------------------
struct FinItem {
  const float* slab;      // [splitk][Cout][Ktot]  (NULL: direct member)
  const float* colsum;    // [splitk][Cout]
  const float* dotpart;   // direct member with BN: [ndot][Cout]
  const bf16_t* w_fwd;
  const float* scale;
  const float* mean;
  const float* invstd;
  float* dw;
  float* dgamma;
  float* dbeta;
  float beta;
  int splitk, Cout, Ktot, map_mode, K4P, ndot, reserved;
};
struct FinGroup {
  int nitems, reserved;
  int blk_start[FIN_MAXI + 2];
  FinItem it[FIN_MAXI];
};
static_assert(sizeof(FinGroup) <= 4096, "kernel-argument block too large");

// Slab members: one block per output channel.  The 256 threads are arranged as K4P k-lanes (float4 each) x SL
// split-lanes so that short rows (Ktot = 64) still use the whole block; split-lane partial sums are combined through
// LDS in lane order (deterministic).  Direct members: one thread per channel.
//   map_mode 0: dw index = co*Ktot + k  ([Cout][kh][kw][Cin] = channels_last view of the OIHW grad)
//   map_mode 1: stem, k = (kh*8 + kw)*4 + c  ->  dw[co][c][kh][kw] contiguous (pads dropped)
//   map_mode 4*cpg: grouped conv in block-diagonal form -> dw[co][kh][kw][cpg]
template <bool F16>
__global__ __launch_bounds__(256) void wgrad_finalize_group_kernel(const FinGroup grp) {
  __shared__ f32x4_t part[256];
  __shared__ float red[4];
  int idx = 0;
#pragma unroll
  for (int i = 1; i < FIN_MAXI; ++i) idx += ((int)blockIdx.x >= grp.blk_start[i]) ? 1 : 0;
  const FinItem& p = grp.it[idx];
  const int lb = (int)blockIdx.x - grp.blk_start[idx];
  const int tid = threadIdx.x;
  const int Cout = p.Cout;
  const float beta = p.beta;
  if (p.slab == nullptr) {
    const int co = lb * 256 + tid;
    if (co >= Cout) return;
    const float cs = p.colsum[co];
    if (p.dbeta) p.dbeta[co] = (beta != 0.f) ? beta * p.dbeta[co] + cs : cs;
    if (p.mean && p.invstd && p.dgamma) {
      float d = 0.f;
      for (int t = 0; t < p.ndot; ++t) d += p.dotpart[(int64_t)t * Cout + co];
      const float dg = (d - p.mean[co] * cs) * p.invstd[co];
      p.dgamma[co] = (beta != 0.f) ? beta * p.dgamma[co] + dg : dg;
    }
    return;
  }
  const int co = lb;
  const int splitk = p.splitk, Ktot = p.Ktot, K4P = p.K4P, map_mode = p.map_mode;
  const float sc = p.scale ? p.scale[co] : 1.f;
  const int K4 = Ktot >> 2;
  const int SL = 256 / K4P;
  const int k4l = tid % K4P, spl = tid / K4P;
  const f32x4_t* slab4 = (const f32x4_t*)p.slab;
  const bf16_t* w_fwd = p.w_fwd;
  float* dw = p.dw;
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
  // block reduce dot; colsum over splits by wave 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  if (tid < 64) {
    float cs = 0.f;
    for (int sp = tid; sp < splitk; sp += 64) cs += p.colsum[(int64_t)sp * Cout + co];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_down(cs, o, 64);
    if (tid == 0) {
      const float d = (red[0] + red[1]) + (red[2] + red[3]);
      if (p.dbeta) p.dbeta[co] = (beta != 0.f) ? beta * p.dbeta[co] + cs : cs;
      if (p.mean && p.invstd && p.dgamma) {
        const float dg = (d - p.mean[co] * cs) * p.invstd[co];
        p.dgamma[co] = (beta != 0.f) ? beta * p.dgamma[co] + dg : dg;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side: geometry, plan, launches
// ---------------------------------------------------------------------------------------------
// "same" convs: for k = 3 the padding is the dilation (conv3x3_group: padding = dilation, layers.py:20-32)
static int conv_dil(int k, int pad) { return k == 3 ? pad : 1; }
static int conv_out(int H, int k, int stride, int pad) {
  return (H + 2 * pad - (conv_dil(k, pad) * (k - 1) + 1)) / stride + 1;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

// Tile shapes of the tap-per-tile kernel: {co, ci, wave rows, wave cols, ring depth} and, for the planner's cost
// model, the workgroups of that shape a CU holds (LDS: NST * 64 * (co + ci) * 2 bytes of 160 KB; 32 waves) and the
// cycles one 64-pixel K-step takes with the CU that full.  The kernel is bound by its LDS-DMA stream, and a CU moves
// ~16 / 30 / 36 / 41 B/clk with 4 / 8 / 12 / >= 16 loading waves (scripts/trace_gemm.py), shared by its
// workgroups.
struct TapShape { int bmw, bnw, wm, wn, nst, per_cu, step_clk; };
static const TapShape kTapShapes[] = {
    {64, 32, 2, 2, 2, 6, 1800},     // 0  stem (Ktap = 32)
    {64, 64, 2, 2, 2, 5, 2000},     // 1
    {64, 64, 2, 2, 3, 3, 1365},     // 2
    {128, 64, 2, 2, 2, 3, 2050},    // 3
    {128, 64, 2, 2, 3, 2, 1640},    // 4
    {256, 64, 4, 2, 2, 2, 2000},    // 5
    {256, 64, 4, 2, 3, 1, 1460},    // 6
    {128, 128, 2, 2, 2, 2, 2180},   // 7
    {256, 128, 4, 2, 2, 1, 1750},   // 8
};
static const int kNumTapShapes = (int)(sizeof(kTapShapes) / sizeof(kTapShapes[0]));
static const int kT9StepClk = 2600;   // nine-tap kernel: 72 MFMAs per wave and K-step, one 8-wave workgroup per CU

struct ItemPlan {
  // geometry
  int Hin, Win, Cpix, Ktap, Ho, Wo, Cout, sa, M, ntaps, Ktot, tapgen, grouped, map_mode;
  bool t9_ok, bn;
  // decomposition
  int variant;   // nine-tap kernel: 1 = 128-channel tile, 2 = 64-channel tile; 0: tap-per-tile kernel with shape kTapShapes[shape]
  int shape;
  int tiles_co, tiles_k, splitk, mchunk, direct, blocks, ndot;
  int step_clk, tmin, slots;   // cost model: cycles per K-step, fewest K-steps per split, workgroup slots of the chip
  int64_t off_out, off_colsum, off_dot;   // workspace offsets in floats (off_out unused for direct members)
};

static int item_geometry(const tdn_wgrad_item& it, ItemPlan& g) {
  TDN_CHECK(it.x && it.g && it.w_fwd && it.dw, "wgrad item: NULL pointer");
  TDN_CHECK(it.N > 0 && it.H > 0 && it.W > 0, "wgrad item: bad tensor shape N=%d H=%d W=%d", it.N, it.H, it.W);
  g.grouped = 0;
  g.t9_ok = false;
  g.bn = it.mean && it.invstd && it.dgamma;
  if (it.kind == TDN_WGRAD_STEM) {
    TDN_CHECK(it.H % 2 == 0 && it.W % 2 == 0 && it.Cout % 64 == 0, "stem wgrad: bad shape");
    g.Hin = it.H + 6; g.Win = it.W + 8; g.Cpix = 4; g.Ktap = 32; g.Ho = it.H / 2; g.Wo = it.W / 2;
    g.Cout = it.Cout; g.sa = 2; g.ntaps = 7; g.tapgen = 1 << 24; g.map_mode = 1;
  } else {
    TDN_CHECK(it.kind == TDN_WGRAD_CONV || it.kind == TDN_WGRAD_GCONV, "wgrad item: bad kind %d", it.kind);
    const int k = it.k, stride = it.stride, pad = it.pad;
    TDN_CHECK(k == 1 || k == 3, "kernel size %d not supported", k);
    TDN_CHECK(stride == 1 || stride == 2, "stride %d not supported", stride);
    TDN_CHECK((k == 1 && pad == 0) || (k == 3 && pad >= 1 && pad <= 32), "pad %d not supported for k=%d", pad, k);
    TDN_CHECK(it.Cin % 64 == 0 && it.Cout % 64 == 0, "channels must be multiples of 64 (Cin=%d Cout=%d)", it.Cin,
              it.Cout);
    g.Hin = it.H; g.Win = it.W; g.Cpix = it.Cin; g.Ktap = it.Cin;
    g.Ho = conv_out(it.H, k, stride, pad); g.Wo = conv_out(it.W, k, stride, pad);
    g.Cout = it.Cout; g.sa = stride; g.ntaps = k * k;
    g.tapgen = k | (pad << 8) | (conv_dil(k, pad) << 16);
    g.map_mode = 0;
    if (it.kind == TDN_WGRAD_GCONV) {
      const int C = it.Cout, groups = it.groups;
      TDN_CHECK(it.Cin == it.Cout, "grouped conv: Cin must equal Cout");
      TDN_CHECK(groups > 0 && C % groups == 0 && C % 64 == 0 && (C / groups) <= 64 && 64 % (C / groups) == 0,
                "grouped conv: need C %% 64 == 0 and channels per group dividing 64 (C=%d, groups=%d)", C, groups);
      g.grouped = 1; g.Ktap = 64; g.map_mode = 4 * (C / groups);
    } else {
      g.t9_ok = (k == 3 && stride == 1 && pad == 1 && it.W >= 8 && it.Cout % 64 == 0 && it.Cin % 64 == 0);
    }
  }
  TDN_CHECK(g.Ho > 0 && g.Wo > 0, "wgrad item: empty output");
  TDN_CHECK((int64_t)it.N * g.Ho * g.Wo < (1ll << 31) / 4 && (int64_t)it.N * g.Hin * g.Win < (1ll << 31) / 4,
            "tensor too large for 32-bit pixel indexing");
  g.M = it.N * g.Ho * g.Wo;
  g.Ktot = g.ntaps * g.Ktap;
  return 0;
}

// launch key of a member: -1 nine-tap kernel, else its tile shape id
static std::mutex g_plan_mutex;
static std::unordered_map<uint64_t, double> g_plan_cache;   // immutable facts about shapes: split duration per group

static inline int launch_key(const ItemPlan& g) { return g.variant ? -g.variant : g.shape; }

// members of one launch: all nine-tap members / all members of one tile shape, at most WG_MAXI at a time; launch
// order: nine-tap members first (longest workgroups); members keep their order inside a launch
static void launch_lists(const std::vector<ItemPlan>& plans, std::vector<std::vector<int>>& lists) {
  const int n = (int)plans.size();
  for (int key = -2; key < kNumTapShapes; ++key) {
    std::vector<int> cur;
    for (int i = 0; i < n; ++i) {
      if (launch_key(plans[i]) != key) continue;
      cur.push_back(i);
      if ((int)cur.size() == WG_MAXI) { lists.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) lists.push_back(cur);
  }
}

// splits of one member for a target of T K-steps per split
static void split_member(ItemPlan& g, int T) {
  const int ksteps = ceil_div(g.M, 64);
  if (T < 1) T = 1;
  int splitk = (ksteps + T / 2) / T;   // nearest: 33 K-steps at T = 24 stay one split (and skip the slabs)
  if (splitk < 1) splitk = 1;
  if (splitk > 256) splitk = 256;
  // splits are dealt round-robin to the 8 XCD labels (kernel's work map): keep the per-XCD load even
  if (splitk >= 8) splitk = (splitk + 4) / 8 * 8;
  const int mchunk = ceil_div(ceil_div(g.M, splitk), 64) * 64;
  g.splitk = ceil_div(g.M, mchunk);
  g.mchunk = mchunk;
  g.blocks = wg_blocks(g.splitk, g.tiles_co * g.tiles_k);
}

// modelled cycles of the group's launches for the current splits: workgroups are handed to the chip's slots in order
// (what the dispatcher does); a workgroup costs its K-steps plus writing its fp32 tile at ~20 B/clk; the finalize
// pass re-reads every slab at ~4 KB/clk chip-wide
static double model_group(const std::vector<ItemPlan>& plans, const std::vector<std::vector<int>>& lists) {
  double total = 0.0, slab_bytes = 0.0;
  std::vector<double> heap;
  for (const std::vector<int>& L : lists) {
    const int slots = plans[L[0]].slots;
    heap.assign(slots, 0.0);   // min-heap of slot-free times (std::*_heap with greater)
    auto cmp = [](double a, double b) { return a > b; };
    double end = 0.0;
    for (int i : L) {
      const ItemPlan& g = plans[i];
      const int tile_floats = g.variant ? 9 * (g.variant == 2 ? 64 : 128) * 64 : kTapShapes[g.shape].bmw * kTapShapes[g.shape].bnw;
      const double c = (double)ceil_div(g.mchunk, 64) * g.step_clk + tile_floats * 4.0 / 20.0 + 3000.0;
      const int live = g.splitk * g.tiles_co * g.tiles_k;
      for (int b = 0; b < live; ++b) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const double t = heap.back() + c;
        heap.back() = t;
        std::push_heap(heap.begin(), heap.end(), cmp);
        if (t > end) end = t;
      }
      if (!g.direct) slab_bytes += (double)g.splitk * g.Cout * g.Ktot * 4.0;
    }
    total += end;
  }
  return total + slab_bytes / 4096.0;
}

// Decomposition of a group.  Every member is cut into splits of about the same modelled duration D instead of each
// layer filling the chip by itself; D is the candidate (a geometric grid between TDN_WGRAD_DMIN and _DMAX cycles) with
// the smallest modelled time, subject to a floor of K-steps per split below which a split's fp32 tile — written once,
// read once — rivals the operands it streams (nine-tap kernel: 288 KB per workgroup).
static int plan_group(const tdn_wgrad_item* items, int n, std::vector<ItemPlan>& plans, int64_t* ws_floats) {
  TDN_CHECK(items != nullptr && n > 0, "wgrad group: no items");
  plans.resize(n);
  const int t9_mode = env_int("TDN_WGRAD9", 1);   // 0: never use the nine-tap kernel
  const int t9_64 = env_int("TDN_WGRAD9_64", 1);
  const int shape_env = env_int("TDN_WGRAD_SHAPE", -1);   // force kTapShapes[id] where it divides the member
  const int s256 = env_int("TDN_WGRAD_S256", 8), s128 = env_int("TDN_WGRAD_S128", 4), s64 = env_int("TDN_WGRAD_S64", 2);
  const int s256f = env_int("TDN_WGRAD_S256F", 5);   // 256-channel members whose Cin does not divide the s256 tile
  const int uniform = env_int("TDN_WGRAD_UNIFORM", 0);   // 1: one tile shape per group (fewest launches)
  const int tmin_tap = env_int("TDN_WGRAD_TMIN", 24), tmin_t9 = env_int("TDN_WGRAD9_TMIN", 32);
  const int direct_ok = env_int("TDN_WGRAD_DIRECT", 1);
  int min_bmw = 256;
  for (int i = 0; i < n; ++i) {
    ItemPlan& g = plans[i];
    if (item_geometry(items[i], g)) return -1;
    // 64-channel tile only where the 128-channel one does not divide Cout (TDN_WGRAD9_64=0: such members stay with
    // the tap-per-tile kernel)
    g.variant = (g.t9_ok && t9_mode != 0) ? (g.Cout % 128 == 0 ? 1 : (t9_64 ? 2 : 0)) : 0;
    if (!g.variant && !g.grouped && g.Ktap != 32) {
      const int b = g.Cout % 256 == 0 ? 256 : (g.Cout % 128 == 0 ? 128 : 64);
      if (b < min_bmw) min_bmw = b;
    }
  }
  for (int i = 0; i < n; ++i) {
    ItemPlan& g = plans[i];
    if (g.variant) {
      g.shape = -1;
      g.tiles_co = g.Cout / (g.variant == 2 ? 64 : 128);
      g.tiles_k = g.Ktap / 64;
      g.step_clk = kT9StepClk;
      g.tmin = tmin_t9;
      g.slots = g.variant == 2 ? 512 : 256;   // 70 KB of LDS per 4-wave workgroup: two per CU
    } else {
      int bmw = g.Cout % 256 == 0 ? 256 : (g.Cout % 128 == 0 ? 128 : 64);
      if (uniform && bmw > min_bmw) bmw = min_bmw;
      int shape = g.Ktap == 32 ? 0 : ((g.grouped || bmw == 64) ? s64 : (bmw == 128 ? s128 : s256));
      auto fits = [&](int sh) {
        return sh >= 0 && sh < kNumTapShapes && g.Cout % kTapShapes[sh].bmw == 0 && g.Ktap % kTapShapes[sh].bnw == 0 &&
               !(g.grouped && (kTapShapes[sh].bmw != 64 || kTapShapes[sh].bnw != 64));
      };
      if (!fits(shape) && bmw == 256 && !g.grouped && fits(s256f)) shape = s256f;   // e.g. Cin = 64 under a 128-wide ci tile
      if (!fits(shape)) shape = g.Ktap == 32 ? 0 : (g.Ktap % 64 == 0 ? 2 : 0);
      if (shape_env >= 0 && shape_env < kNumTapShapes && !g.grouped && g.Cout % kTapShapes[shape_env].bmw == 0 &&
          g.Ktap % kTapShapes[shape_env].bnw == 0)
        shape = shape_env;
      g.shape = shape;
      const TapShape& s = kTapShapes[shape];
      g.tiles_co = g.Cout / s.bmw;
      g.tiles_k = g.ntaps * (g.Ktap / s.bnw);
      g.step_clk = s.step_clk;
      g.tmin = tmin_tap;
      g.slots = 256 * s.per_cu;
    }
  }
  std::vector<std::vector<int>> lists;
  launch_lists(plans, lists);
  const double dmin = env_int("TDN_WGRAD_DMIN", 40000), dmax = env_int("TDN_WGRAD_DMAX", 400000);
  const int fixed_t = env_int("TDN_WGRAD_T", 0);   // sweeps: the same K-steps per split for every member
  // the search below costs ~a millisecond per group: its result (D) is remembered per (member shapes, knobs)
  uint64_t key = 1469598103934665603ull;
  auto mix = [&key](int64_t v) { key = (key ^ (uint64_t)v) * 1099511628211ull; };
  mix(n); mix((int64_t)dmin); mix((int64_t)dmax); mix(fixed_t);
  for (const ItemPlan& g : plans) {
    mix(g.M); mix(g.Cout); mix(g.Ktap); mix(g.ntaps); mix(g.variant); mix(g.shape); mix(g.tmin); mix(g.map_mode);
    mix(g.step_clk);
  }
  double best = -1.0, bestD = dmin;
  bool hit = false;
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    auto itc = g_plan_cache.find(key);
    if (itc != g_plan_cache.end()) { bestD = itc->second; hit = true; }
  }
  const int ncand = (fixed_t > 0 || hit) ? 0 : 24;
  for (int c = 0; c < ncand; ++c) {
    const double D = dmin * pow(dmax / dmin, ncand > 1 ? (double)c / (ncand - 1) : 0.0);
    for (ItemPlan& g : plans) {
      int T = fixed_t > 0 ? fixed_t : (int)(D / g.step_clk + 0.5);
      if (T < g.tmin) T = g.tmin;
      split_member(g, T);
      g.direct = (g.splitk == 1 && g.map_mode == 0 && direct_ok) ? 1 : 0;
    }
    const double t = model_group(plans, lists);
    if (best < 0 || t < best) { best = t; bestD = D; }
  }
  if (ncand > 0) {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    g_plan_cache[key] = bestD;
  }
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    ItemPlan& g = plans[i];
    int T = fixed_t > 0 ? fixed_t : (int)(bestD / g.step_clk + 0.5);
    if (T < g.tmin) T = g.tmin;
    split_member(g, T);
    g.direct = (g.splitk == 1 && g.map_mode == 0 && direct_ok) ? 1 : 0;
    g.ndot = (g.direct && g.bn) ? g.tiles_k : 0;
    g.off_out = off;
    if (!g.direct) off += (int64_t)g.splitk * g.Cout * g.Ktot;
    g.off_colsum = off;
    off += (int64_t)g.splitk * g.Cout;
    g.off_dot = off;
    off += (int64_t)g.ndot * g.Cout;
    off = (off + 63) & ~(int64_t)63;   // 256-byte alignment of every member's region
  }
  *ws_floats = off;
  return 0;
}

template <int BMW, int BNW, int WM, int WN, int NST, bool F16>
static int launch_tap_t(const WgGroup& grp, int nblocks, hipStream_t stream) {
  constexpr size_t lds = (size_t)NST * 64 * (BMW + BNW) * 2;
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_group_kernel<BMW, BNW, WM, WN, NST, F16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", (int)lds, hipGetErrorString(e));
    attr_once.mark();
  }
  TDN_LAUNCH((conv_wgrad_group_kernel<BMW, BNW, WM, WN, NST, F16>), dim3(nblocks), dim3(WM * WN * 64), lds,
                     stream, grp);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <bool F16>
static int launch_tap(int shape, const WgGroup& grp, int nblocks, hipStream_t stream) {
  switch (shape) {
    case 0: return launch_tap_t<64, 32, 2, 2, 2, F16>(grp, nblocks, stream);
    case 1: return launch_tap_t<64, 64, 2, 2, 2, F16>(grp, nblocks, stream);
    case 2: return launch_tap_t<64, 64, 2, 2, 3, F16>(grp, nblocks, stream);
    case 3: return launch_tap_t<128, 64, 2, 2, 2, F16>(grp, nblocks, stream);
    case 4: return launch_tap_t<128, 64, 2, 2, 3, F16>(grp, nblocks, stream);
    case 5: return launch_tap_t<256, 64, 4, 2, 2, F16>(grp, nblocks, stream);
    case 6: return launch_tap_t<256, 64, 4, 2, 3, F16>(grp, nblocks, stream);
    case 7: return launch_tap_t<128, 128, 2, 2, 2, F16>(grp, nblocks, stream);
    case 8: return launch_tap_t<256, 128, 4, 2, 2, F16>(grp, nblocks, stream);
    default: tdn_set_error("wgrad: bad tile shape id %d", shape); return -1;
  }
}

template <bool F16, int WMR>
static int launch_t9(const WgGroup& grp, int nblocks, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(64 * (64 * WMR) + 3 * 72 * 128);
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad9_group_kernel<F16, WMR>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(%d B LDS) failed: %s", (int)lds, hipGetErrorString(e));
    attr_once.mark();
  }
  TDN_LAUNCH((conv_wgrad9_group_kernel<F16, WMR>), dim3(nblocks), dim3(WMR * 128), lds, stream, grp);
  TDN_LAUNCH_CHECK();
  return 0;
}

static void fill_item(WgItem& d, const tdn_wgrad_item& it, const ItemPlan& g, float* ws) {
  memset(&d, 0, sizeof(d));
  d.x = (const bf16_t*)it.x; d.g = (const bf16_t*)it.g; d.w_fwd = (const bf16_t*)it.w_fwd;
  d.scale = it.scale;
  d.out = g.direct ? it.dw : ws + g.off_out;
  d.colsum = ws + g.off_colsum;
  d.dotpart = g.ndot ? ws + g.off_dot : nullptr;
  d.Hin = g.Hin; d.Win = g.Win; d.Cpix = g.Cpix; d.Ktap = g.Ktap; d.Ho = g.Ho; d.Wo = g.Wo; d.Cout = g.Cout;
  d.sa = g.sa; d.M = g.M; d.Mchunk = g.mchunk; d.splitk = g.splitk; d.ntaps = g.ntaps; d.Ktot = g.Ktot;
  d.tiles_co = g.tiles_co; d.tiles_k = g.tiles_k; d.tapgen = g.tapgen; d.grouped = g.grouped; d.direct = g.direct;
  d.beta = it.beta;
}

extern "C" int64_t tdn_wgrad_group_workspace(const tdn_wgrad_item* items, int n, int dtype) {
  if (dtype != TDN_BF16 && dtype != TDN_F16) { tdn_set_error("dtype %d is neither TDN_BF16 nor TDN_F16", dtype); return -1; }
  std::vector<ItemPlan> plans;
  int64_t fl = 0;
  if (plan_group(items, n, plans, &fl)) return -1;
  return fl * 4 + 256;
}

extern "C" int tdn_wgrad_group_plan(const tdn_wgrad_item* items, int n, int dtype, int32_t* per_item,
                                    int32_t* totals) {
  TDN_CHECK(dtype == TDN_BF16 || dtype == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", dtype);
  std::vector<ItemPlan> plans;
  int64_t fl = 0;
  if (plan_group(items, n, plans, &fl)) return -1;
  std::vector<std::vector<int>> lists;
  launch_lists(plans, lists);
  int64_t blocks = 0, slab_kib = 0;
  for (int i = 0; i < n; ++i) {
    const ItemPlan& g = plans[i];
    const int64_t slab = g.direct ? 0 : (int64_t)g.splitk * g.Cout * g.Ktot * 4 / 1024;
    if (per_item) {
      int32_t* o = per_item + 8 * i;
      o[0] = g.variant; o[1] = g.variant ? (g.variant == 2 ? 64 : 128) : kTapShapes[g.shape].bmw; o[2] = g.variant ? 64 : kTapShapes[g.shape].bnw;
      o[3] = g.splitk; o[4] = g.mchunk; o[5] = g.direct; o[6] = g.blocks; o[7] = (int32_t)slab;
    }
    blocks += g.blocks;
    slab_kib += slab;
  }
  if (totals) {
    totals[0] = (int32_t)lists.size();
    totals[1] = ceil_div(n, FIN_MAXI);
    totals[2] = (int32_t)blocks;
    totals[3] = (int32_t)slab_kib;
  }
  return 0;
}

extern "C" int tdn_wgrad_group(const tdn_wgrad_item* items, int n, void* workspace, int64_t workspace_bytes,
                               int dtype, void* stream_) {
  TDN_CHECK(dtype == TDN_BF16 || dtype == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", dtype);
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<ItemPlan> plans;
  int64_t fl = 0;
  if (plan_group(items, n, plans, &fl)) return -1;
  TDN_CHECK(workspace != nullptr && workspace_bytes >= fl * 4, "wgrad workspace too small: %lld < %lld",
            (long long)workspace_bytes, (long long)(fl * 4));
  TDN_CHECK(((uintptr_t)workspace & 15) == 0, "wgrad workspace must be 16-byte aligned");
  float* ws = (float*)workspace;
  std::vector<std::vector<int>> lists;
  launch_lists(plans, lists);
  for (const std::vector<int>& L : lists) {
    WgGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.nitems = (int)L.size();
    grp.reserved = env_int("TDN_WGRAD9_STAGGER", 1);   // nine-tap kernel: staggered LDS-DMA issue (see the kernel)
    int blk = 0;
    for (int j = 0; j < WG_MAXI + 2; ++j) grp.blk_start[j] = INT_MAX;
    for (int j = 0; j < (int)L.size(); ++j) {
      const int i = L[j];
      grp.blk_start[j] = blk;
      fill_item(grp.it[j], items[i], plans[i], ws);
      blk += plans[i].blocks;
    }
    const ItemPlan& g0 = plans[L[0]];
    int rc;
    if (g0.variant == 2) rc = dtype == TDN_F16 ? launch_t9<true, 2>(grp, blk, stream) : launch_t9<false, 2>(grp, blk, stream);
    else if (g0.variant) rc = dtype == TDN_F16 ? launch_t9<true, 4>(grp, blk, stream) : launch_t9<false, 4>(grp, blk, stream);
    else rc = dtype == TDN_F16 ? launch_tap<true>(g0.shape, grp, blk, stream) : launch_tap<false>(g0.shape, grp, blk, stream);
    if (rc) return rc;
  }
  // one finalize launch per FIN_MAXI members
  for (int base = 0; base < n; base += FIN_MAXI) {
    FinGroup fg;
    memset(&fg, 0, sizeof(fg));
    const int cnt = (n - base < FIN_MAXI) ? n - base : FIN_MAXI;
    fg.nitems = cnt;
    for (int j = 0; j < FIN_MAXI + 2; ++j) fg.blk_start[j] = INT_MAX;
    int blk = 0;
    for (int j = 0; j < cnt; ++j) {
      const tdn_wgrad_item& it = items[base + j];
      const ItemPlan& g = plans[base + j];
      FinItem& f = fg.it[j];
      fg.blk_start[j] = blk;
      f.slab = g.direct ? nullptr : ws + g.off_out;
      f.colsum = ws + g.off_colsum;
      f.dotpart = g.ndot ? ws + g.off_dot : nullptr;
      f.w_fwd = (const bf16_t*)it.w_fwd;
      f.scale = it.scale; f.mean = it.mean; f.invstd = it.invstd;
      f.dw = it.dw; f.dgamma = it.dgamma; f.dbeta = it.dbeta; f.beta = it.beta;
      f.splitk = g.splitk; f.Cout = g.Cout; f.Ktot = g.Ktot; f.map_mode = g.map_mode; f.ndot = g.ndot;
      int k4p = 1;
      while (k4p < (g.Ktot >> 2) && k4p < 256) k4p <<= 1;
      f.K4P = k4p;
      blk += g.direct ? ceil_div(g.Cout, 256) : g.Cout;
    }
    if (dtype == TDN_F16)
      TDN_LAUNCH(wgrad_finalize_group_kernel<true>, dim3(blk), dim3(256), 0, stream, fg);
    else
      TDN_LAUNCH(wgrad_finalize_group_kernel<false>, dim3(blk), dim3(256), 0, stream, fg);
    TDN_LAUNCH_CHECK();
  }
  return 0;
}

// ---- single-layer entry points: groups of one -------------------------------------------------------------
static tdn_wgrad_item one_item(int kind, const void* x, const void* g, const void* w_fwd, const float* scale,
                               const float* mean, const float* invstd, float* dw, float* dgamma, float* dbeta,
                               float beta, int N, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                               int groups) {
  tdn_wgrad_item it;
  memset(&it, 0, sizeof(it));
  it.x = x; it.g = g; it.w_fwd = w_fwd; it.scale = scale; it.mean = mean; it.invstd = invstd;
  it.dw = dw; it.dgamma = dgamma; it.dbeta = dbeta; it.beta = beta; it.kind = kind;
  it.N = N; it.H = H; it.W = W; it.Cin = Cin; it.Cout = Cout; it.k = k; it.stride = stride; it.pad = pad;
  it.groups = groups;
  return it;
}
// shape-only queries: the planner never dereferences the tensors
static const void* const kShapeOnly = (const void*)(uintptr_t)256;

extern "C" int64_t tdn_conv2d_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int k, int stride,
                                               int pad) {
  const tdn_wgrad_item it = one_item(TDN_WGRAD_CONV, kShapeOnly, kShapeOnly, kShapeOnly, nullptr, nullptr, nullptr,
                                     (float*)kShapeOnly, nullptr, nullptr, 0.f, N, H, W, Cin, Cout, k, stride, pad, 1);
  return tdn_wgrad_group_workspace(&it, 1, TDN_BF16);
}

int tdn_wgrad_plan(int N, int H, int W, int Cin, int Cout, int k, int stride, int pad, int32_t* o) {
  const tdn_wgrad_item it = one_item(TDN_WGRAD_CONV, kShapeOnly, kShapeOnly, kShapeOnly, nullptr, nullptr, nullptr,
                                     (float*)kShapeOnly, nullptr, nullptr, 0.f, N, H, W, Cin, Cout, k, stride, pad, 1);
  std::vector<ItemPlan> plans;
  int64_t fl = 0;
  if (plan_group(&it, 1, plans, &fl)) return -1;
  const ItemPlan& w = plans[0];
  o[0] = Cout; o[1] = w.Ktot; o[2] = w.M; o[3] = w.variant ? (w.variant == 2 ? 64 : 128) : kTapShapes[w.shape].bmw;
  o[4] = w.variant ? 64 : kTapShapes[w.shape].bnw; o[5] = 64;
  o[6] = w.tiles_co * w.tiles_k; o[7] = w.splitk; o[8] = 1; o[9] = 1; o[10] = k * k; o[11] = w.splitk;
  o[12] = w.mchunk; o[13] = w.Ho; o[14] = w.Wo; o[15] = w.M;
  return 0;
}

extern "C" int tdn_conv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale,
                                const float* mean, const float* invstd, float* dw, float* dgamma,
                                float* dbeta, float beta, int N, int H, int W, int Cin, int Cout, int k,
                                int stride, int pad, void* workspace, int64_t workspace_bytes, int dtype,
                                void* stream) {
  TDN_CHECK(x && g && w_fwd && dw && workspace, "tdn_conv2d_wgrad: NULL pointer");
  const tdn_wgrad_item it = one_item(TDN_WGRAD_CONV, x, g, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H,
                                     W, Cin, Cout, k, stride, pad, 1);
  return tdn_wgrad_group(&it, 1, workspace, workspace_bytes, dtype, stream);
}

// Grouped conv weight gradient (see tdn_gconv2d_fwd): per 64-channel block a dense 64 x (taps * 64) product, of which
// the finalize pass keeps each output channel's own group: dw fp32 [C][k][k][cpg].
extern "C" int64_t tdn_gconv2d_wgrad_workspace(int N, int H, int W, int C, int groups, int k, int stride, int pad) {
  const tdn_wgrad_item it = one_item(TDN_WGRAD_GCONV, kShapeOnly, kShapeOnly, kShapeOnly, nullptr, nullptr, nullptr,
                                     (float*)kShapeOnly, nullptr, nullptr, 0.f, N, H, W, C, C, k, stride, pad, groups);
  return tdn_wgrad_group_workspace(&it, 1, TDN_BF16);
}

extern "C" int tdn_gconv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale,
                                 const float* mean, const float* invstd, float* dw, float* dgamma, float* dbeta,
                                 float beta, int N, int H, int W, int C, int groups, int k, int stride, int pad,
                                 void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK(x && g && w_fwd && dw && workspace, "tdn_gconv2d_wgrad: NULL pointer");
  const tdn_wgrad_item it = one_item(TDN_WGRAD_GCONV, x, g, w_fwd, scale, mean, invstd, dw, dgamma, dbeta, beta, N, H,
                                     W, C, C, k, stride, pad, groups);
  return tdn_wgrad_group(&it, 1, workspace, workspace_bytes, dtype, stream);
}

extern "C" int64_t tdn_stem_conv_wgrad_workspace(int N, int H, int W, int Cout) {
  const tdn_wgrad_item it = one_item(TDN_WGRAD_STEM, kShapeOnly, kShapeOnly, kShapeOnly, nullptr, nullptr, nullptr,
                                     (float*)kShapeOnly, nullptr, nullptr, 0.f, N, H, W, 3, Cout, 7, 2, 3, 1);
  return tdn_wgrad_group_workspace(&it, 1, TDN_BF16);
}

extern "C" int tdn_stem_conv_wgrad(const void* xp, const void* g, const void* w_stem, const float* scale,
                                   const float* mean, const float* invstd, float* dw, float* dgamma,
                                   float* dbeta, float beta, int N, int H, int W, int Cout, void* workspace,
                                   int64_t workspace_bytes, int dtype, void* stream) {
  TDN_CHECK(xp && g && w_stem && dw && workspace, "tdn_stem_conv_wgrad: NULL pointer");
  const tdn_wgrad_item it = one_item(TDN_WGRAD_STEM, xp, g, w_stem, scale, mean, invstd, dw, dgamma, dbeta, beta, N,
                                     H, W, 3, Cout, 7, 2, 3, 1);
  return tdn_wgrad_group(&it, 1, workspace, workspace_bytes, dtype, stream);
}
