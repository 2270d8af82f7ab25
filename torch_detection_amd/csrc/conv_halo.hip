// LDS-resident activation tile ("halo tile") convolution on MFMA (gfx950): forward and stride-1 input gradient of the
// 3x3 and 1x1 convs of the ResNet / FPN hot path.
//
// Replaces nn.Conv2d.forward as built by conv3x3_group / conv1x1_group (models/utils/layers.py:6-32) and called from
// models/backbone/resnet.py:97-119 and models/necks/fpn.py:92-108 — the "LDS-staged im2col tile" of the north star
// (halo rows of the NHWC input staged once, 9 taps x Cin as the K loop).
//
// GEMM view as in conv_igemm.hip:  D[n][m] = sum_k W[n][k] * X[m][k],  m = output pixel, n = output channel,
// k = (tap, input channel).  What differs is where X comes from.  conv_igemm.hip re-gathers the BM x 64 activation
// tile from L2 once per tap (nine LDS-DMA passes over the same pixels per channel chunk).  Here a workgroup owns a
// TH x TW patch of output pixels; per 64-channel chunk the (TH + 2h) x (TW + 2h) input patch is copied to LDS ONCE
// and the nine taps read it through shifted ds_read_b128 addresses; only the BN x 64 weight tile is streamed per
// (tap, chunk) K-step through a 3-deep LDS ring.  A CU ingests L2 -> LDS data at ~70 GB/s whatever the instruction
// (MI355X_MICROARCH.md, "Indexed rows: gather into LDS", "ring-gemm"), which is what bounds every
// 64..128-pixel tile of the generic kernel: the resident patch cuts the activation bytes ~7x for 3x3 convs, and for
// 1x1 convs (no halo, linear pixel tiles) it lets one workgroup run several output-channel tiles over the same
// resident pixels (activation-stationary).
//
// LDS image of one channel chunk: rows of 128 B (64 channels of one input pixel), row index hp = hr * pitch + hc over
// the haloed patch; the eight 16-byte slots of a row are XOR-swizzled with f(hc) = (hc >> 1) & 7 on the SOURCE address
// of the LDS-DMA (the destination of global_load_lds is lane-linear) and on the read address.  `pitch` is even, so a
// fragment's 16 consecutive pixels of one patch row hit 16 different (row parity, slot) pairs: conflict-free
// ds_read_b128 for every tap shift.  A patch whose width is not a multiple of 16 lets some fragments straddle two patch
// rows (two runs of pixels whose slots can collide pairwise: a 2-way conflict on those fragments only) — the planner
// weighs that against the pixels a multiple-of-16 width would waste on maps such as 50 x 84.
#include "common.h"
#include <type_traits>

struct HaloParams {
  const bf16_t* in;
  const bf16_t* wt;
  bf16_t* out;
  const float* scale;
  const float* shift;
  const bf16_t* addend;
  const bf16_t* mask;
  int Hin, Win, Cpix;        // input tensor (pixels per image, channels per pixel)
  int Hout, Wout, Cout, M;   // output tensor; M = images * Hout * Wout
  int sa;                    // input pixel = output pixel * sa (+ tap); sa == 2 only with linear tiles (1x1 / s2)
  int halo;                  // dilation * (k - 1) / 2
  int lin;                   // 1: tile = BM consecutive output pixels (1x1 convs); 0: TH x TW patch of one image
  int TH, TW, THh, TWh, tile_px;   // patch, haloed patch, TH * TW
  int pitch, hrows, pieces;  // LDS rows per patch row; rows per chunk image (multiple of 8); hrows / 8
  int tiles_w, tiles_img;    // patches per image row / per image
  int ngroups_n, nt_per_wg;  // workgroup columns; output-channel tiles each workgroup runs one after the other
  int ntiles, nwg_pad;
  int nchunks, xbuf, x_resident;   // Cin / 64; chunk images held in LDS; all of them (loaded once, first pass only)
  int wt_row, Ktap;          // elements per weight row; elements between two taps of a row
  int taps[9];               // (dh + 64) | (dw + 64) << 8 | widx << 16
  int addend_mode, addend_h, addend_w, relu, out_f32;
  unsigned ng_mul, ng_shr, ti_mul, ti_shr, tw_mul, tw_shr, pit_mul, pit_shr, TW_mul, TW_shr;
  unsigned hw_mul, hw_shr, w_mul, w_shr;   // output pixel index -> (image, oh, ow)
};

__device__ __forceinline__ int hdiv(int n, unsigned mul, unsigned shr) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}

__device__ __forceinline__ bf16x8_t lds_read_b128_u(unsigned addr) {
  return *(const TDN_LDS bf16x8_t*)(TDN_LDS char*)(size_t)addr;
}

// LDS-DMA with the destination given as a 32-bit LDS byte address (wave-uniform)
__device__ __forceinline__ void glds16_u(const void* gsrc, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"((const TDN_GLOBAL void*)gsrc),
               "s"(__builtin_amdgcn_readfirstlane(lds_addr))   // provably wave-uniform for the "s" constraint
               : "memory");
#endif
}

template <int N>
__device__ __forceinline__ void halo_wait_vm_and_barrier() {
  // lgkmcnt(0): this wave's LDS reads of the previous K-step have RETURNED before any wave may overwrite what they
  // read (ring slot, chunk image) — the compiler is free to sink the MFMAs that consume them below the barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// FM x FN 16x16 fragments per wave (wave tile 16FM pixels x 16FN channels), WM x WN waves: BM = 16 FM WM pixels,
// BN = 16 FN WN channels per pass.  NTAPS 9 (3x3, the tap loop is unrolled: shifted read addresses are three
// precomputed register sets, one per kernel column) or 1 (1x1).  XI9: activation-patch DMA pieces a wave issues per
// K-step of a 3x3 conv (1 or 2: the patch image of the next chunk has at most 8 XI9 pieces per wave; 1x1 tiles have
// exactly 2 FM / WN per wave and K-step).
// NWL > 0: the workgroup has NWL dedicated LOADER waves beside its WM x WN consumer waves.  An LDS-DMA instruction holds
// the issuing wave's instruction stream for ~100-250 cycles per 1 KiB piece (MI355X_MICROARCH.md "LDS-DMA piece issue
// cost"; measured here: the DMA issue of a 128x128 tile adds ~500 cycles to a K-step whose MFMAs take ~410, wherever
// it is placed in the wave's stream), so a wave that both loads and multiplies serialises the two.  Loader waves
// (one per SIMD, next to one consumer wave) take the issue stalls; consumers only read LDS and feed the matrix pipe.
template <int FM, int FN, int WM, int WN, int NST, int NTAPS, int XI9, bool F16, int ABL = 0, int NWL = 0>
__global__ __launch_bounds__((WM * WN + NWL) * 64, ((WM * WN + NWL) >= 16 ? 4 : ((WM * WN + NWL) > 8 ? 3 : 2))) void conv_halo_kernel(const HaloParams p) {
  extern __shared__ __attribute__((aligned(1024))) char halo_smem[];
  constexpr int NW = WM * WN;            // consumer waves
  constexpr int NWLD = NWL > 0 ? NWL : NW;   // waves that issue LDS-DMA
  constexpr int BM = FM * WM * 16, BN = FN * WN * 16;
  constexpr int WTN = FN * 16;
  constexpr int W_BYTES = BN * 128;
  constexpr int LW = BN / (8 * NWLD);    // weight-tile DMA pieces (1 KiB) per loading wave and K-step
  constexpr int CPL = 4 * FN;            // consecutive channels a lane owns per pixel (see conv_igemm.hip, WIDE)
  constexpr int NV = NTAPS == 9 ? 3 : 1;
  constexpr int XI = NTAPS == 9 ? XI9 : BM / (8 * NWLD);   // activation pieces per loading wave and K-step
  constexpr int XSTEPS = NTAPS == 9 ? 11 - NST : 1;      // K-steps of a chunk in which they are issued (see LA below)
  constexpr int PXM = XI * XSTEPS;                       // most pieces of one chunk image a wave can own
  static_assert(XI >= 1 && (NTAPS == 9 || BM % (8 * NWLD) == 0), "activation pieces do not split over the waves");
  static_assert(NST >= 3 && NST <= 9 && LW * (NST - 1) + (NTAPS == 9 ? 0 : XI * (NST - 2)) < 64, "ring depth / vmcnt range");
  static_assert(BN % (8 * NWLD) == 0 && LW >= 1, "weight tile does not split over the waves");
  static_assert(FN == 2 || FN == 4, "wide epilogue needs 8 or 16 channels per lane");
  static_assert(NTAPS == 1 || NTAPS == 9, "1x1 or 3x3");
  (void)BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = NWL > 0 && wave >= NW;          // wave-uniform role
  const int lw = NWL > 0 ? wave - NW : wave;             // index among the loading waves (loaders only if NWL > 0)
  const int wm = (is_loader ? 0 : wave) / WN, wn = (is_loader ? 0 : wave) % WN;
  const int fr = lane & 15, fq = lane >> 4;
  const int lrow = lane >> 3, lchunk = lane & 7;

  const int bid = blockIdx.x;
  const int tile = (bid & 7) * (p.nwg_pad >> 3) + (bid >> 3);
  if (tile >= p.ntiles) return;
  const int mt = hdiv(tile, p.ng_mul, p.ng_shr);
  const int ng = tile - mt * p.ngroups_n;
  int img = 0, oh0 = 0, ow0 = 0, m0 = 0;
  if (p.lin) {
    m0 = mt * BM;
  } else {
    img = hdiv(mt, p.ti_mul, p.ti_shr);
    const int rem = mt - img * p.tiles_img;
    const int tr = hdiv(rem, p.tw_mul, p.tw_shr);
    oh0 = tr * p.TH;
    ow0 = (rem - tr * p.tiles_w) * p.TW;
  }

  const unsigned sX = (unsigned)(size_t)(TDN_LDS char*)halo_smem;
  const unsigned xbytes = (unsigned)p.hrows * 128u;
  const unsigned sW = sX + (unsigned)p.xbuf * xbytes;
  const unsigned sSink = sW + NST * W_BYTES + (unsigned)wave * 1024u;   // 1 KiB per wave: destination of dummy copies

  // per-tap LDS row shift of the patch image (SGPRs); the weights of tap t are K-columns [t Ktap, (t + 1) Ktap) of a
  // weight row (host-checked: widx == t), so the weight loader just walks the row
  int dh_off[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) dh_off[t] = ((p.taps[t] & 0xff) - 64) * p.pitch * 128;

  // ---- fragment constants: output pixel of (fragment j, lane) and its LDS row for each kernel column ----
  int opix[FM];
  unsigned xoff[NV][FM];
  // Pixel of (fragment j, MFMA column fr): the 16 pixels of a fragment are dealt to the columns so that the eight
  // lanes a ds_read_b128 lane group takes from one k-quarter ({0-3, 12-15} or {4-11}) hold the even resp. the odd
  // pixels: the two k-quarters of a lane group then sit in opposite 128-byte halves of the bank rows for EVERY tap
  // shift (with pixels in column order the shifted taps were 2-way conflicts), and within a quarter the eight same-
  // parity pixels take eight different slots.  The epilogue uses the same map (MFMA output column = fr).
  const int po = fr < 4 ? 2 * fr : (fr < 12 ? 2 * (fr - 4) + 1 : 2 * (fr - 8));
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int pidx = (wm * FM + j) * 16 + po;
    int hp0, hc0, op;
    if (p.lin) {
      const int m = m0 + pidx;
      op = m < p.M ? m : -1;
      hp0 = pidx;
      hc0 = pidx;
    } else {
      int r = hdiv(pidx, p.TW_mul, p.TW_shr);
      int c = pidx - r * p.TW;
      const bool ok = pidx < p.tile_px && oh0 + r < p.Hout && ow0 + c < p.Wout;
      op = ok ? (img * p.Hout + oh0 + r) * p.Wout + ow0 + c : -1;
      if (pidx >= p.tile_px) { r = 0; c = 0; }   // fragments past the patch read row (0, 0): any valid LDS address
      hp0 = (r + p.halo) * p.pitch + c + p.halo;
      hc0 = c + p.halo;
    }
    opix[j] = op;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int dw = ((p.taps[v] >> 8) & 0xff) - 64;
      const int hp = hp0 + dw, hc = hc0 + dw;
      xoff[v][j] = (unsigned)(hp * 128 + ((fq ^ ((hc >> 1) & 7)) << 4));
    }
  }

  // ---- weight tile: loader offsets and fragment read offsets (row permutation / swizzle of conv_igemm.hip, WIDE) ----
  auto swz_w = [](int row) { return ((row >> 1) & 1) | (((row / CPL) & 3) << 1); };
  unsigned b_off[LW];
#pragma unroll
  for (int it = 0; it < LW; ++it) {
    const int r = (it * NWLD + lw) * 8 + lrow;
    b_off[it] = (unsigned)(((int64_t)r * p.wt_row + (lchunk ^ swz_w(r)) * 8) * 2);
  }
  const int w_row0 = (fr >> 2) * CPL + (fr & 3);
  const int f_rd_w = ((fr & 3) >> 1) | ((fr >> 2) << 1);
  unsigned rdw_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) rdw_off[kk] = (unsigned)((wn * WTN + w_row0) * 128 + (((kk * 4 + fq) ^ f_rd_w) << 4));

  const char* zero_src = (const char*)g_zero_page + lchunk * 16;
  const int n_base = ng * p.nt_per_wg * BN;
  const int nchunks = p.nchunks;
  const int S_total = p.nt_per_wg * nchunks * NTAPS;

  // ---- weight loader: tiles in K-step order (pass, chunk, tap), taps innermost; `wt_next` walks the weight rows ----
  int l_left = S_total, l_chunk = 0;
  const char* wt_next = (const char*)p.wt + (int64_t)n_base * p.wt_row * 2;
  const int tap_stride = p.Ktap * 2;
  const int64_t chunk_wrap = 128 - (int64_t)(NTAPS - 1) * tap_stride;           // last tap of a chunk -> next chunk
  const int64_t pass_wrap = (int64_t)BN * p.wt_row * 2 - (int64_t)nchunks * 128; // ... of the last chunk -> next pass
  // tl: tap of the tile being staged (compile-time in the unrolled 3x3 loop, where tile s + NST has tap (t + NST) % 9)
  auto w_piece = [&](int slot, int it) {   // piece `it` of the next weight tile (or a dummy past the end)
    if (l_left > 0) glds16_u(wt_next + b_off[it], sW + (unsigned)slot * W_BYTES + (unsigned)(it * NWLD + lw) * 1024u);
    else glds16_u(zero_src, sSink);        // keeps the counted vmcnt waits uniform
  };
  auto w_advance = [&](auto tlc) {
    constexpr int tl = decltype(tlc)::value;
    if (l_left > 0) {
      --l_left;
      if constexpr (tl + 1 == NTAPS) {
        wt_next += chunk_wrap;
        if (++l_chunk == nchunks) { l_chunk = 0; wt_next += pass_wrap; }
      } else {
        wt_next += tap_stride;
      }
    }
  };
  auto stage_w = [&](int slot, auto tlc) {
#pragma unroll
    for (int it = 0; it < LW; ++it) w_piece(slot, it);
    w_advance(tlc);
  };

  // ---- activation patch loader: piece q = LDS rows 8q .. 8q+7 of one chunk image; wave w owns pieces w + NW i ----
  // Per piece the lane's source is the same for every channel chunk but for + 128 B per chunk: its byte offset from
  // p.in (host-checked to fit 32 bits) is computed once; X_ZERO marks rows outside the patch / image (zero page).
  constexpr unsigned X_ZERO = 0xffffffffu;
  unsigned xsrc[PXM];
#pragma unroll
  for (int i = 0; i < PXM; ++i) {
    const int q = lw + NWLD * i;
    const int hp = q * 8 + lrow;
    bool ok;
    int hcx;
    unsigned pix = 0;
    if (p.lin) {
      const int m = m0 + hp;
      ok = m < p.M;
      hcx = hp;
      if (p.sa == 1) {
        pix = (unsigned)m;
      } else {
        const int im = hdiv(ok ? m : 0, p.hw_mul, p.hw_shr);
        const int rem = (ok ? m : 0) - im * (p.Hout * p.Wout);
        const int oh = hdiv(rem, p.w_mul, p.w_shr);
        const int ow = rem - oh * p.Wout;
        pix = (unsigned)((im * p.Hin + oh * p.sa) * p.Win + ow * p.sa);
      }
    } else {
      const int hr = hdiv(hp, p.pit_mul, p.pit_shr);
      const int hc = hp - hr * p.pitch;
      const int ih = oh0 - p.halo + hr, iw = ow0 - p.halo + hc;
      ok = hr < p.THh && hc < p.TWh && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      hcx = hc;
      pix = (unsigned)((img * p.Hin + ih) * p.Win + iw);
    }
    const int sl = lchunk ^ ((hcx >> 1) & 7);
    xsrc[i] = (ok && q < p.pieces) ? (pix * (unsigned)p.Cpix + sl * 8) * 2u : X_ZERO;
  }
  const char* zero_x = (const char*)g_zero_page;   // + (lane's slot) is irrelevant: the page is all zero
  auto issue_x_piece = [&](int cf, int i) {        // flat chunk cf = pass * nchunks + chunk; i compile-time after unrolling
    const int q = lw + NWLD * i;
    if (q < p.pieces) {                            // wave-uniform
      const char* in_c = (const char*)p.in + (cf % nchunks) * 128;
      const char* src = xsrc[i] != X_ZERO ? in_c + xsrc[i] : zero_x;
      glds16_u(src, sX + (unsigned)(cf % p.xbuf) * xbytes + (unsigned)q * 1024u);
    }
  };
  const int CF_total = p.nt_per_wg * nchunks;
  auto x_needed = [&](int cf) { return cf < CF_total && (!p.x_resident || cf < nchunks); };
  // Why the ring is deep: an LDS-DMA piece lands ~1.1 us (~2,500 cycles) after its issue under load
  // (MI355X_MICROARCH.md "ldsdma-fill"; the ablation runs in DESIGN.md), so a CU only reaches its ~40 B/clk ingest
  // rate with ~64-96 KB in flight: NST - 1 weight tiles are kept in flight (the 2-deep ring of conv_igemm.hip holds
  // one, which is what pins its K-step at ~1,200 cycles whatever the tile).
  // Counted waits: at the barrier of K-step s all but the youngest LW (NST - 2) copies have landed, i.e. weight
  // tile s+1 and every activation piece issued NST - 2 barriers ago or earlier.  So the pieces of the next chunk
  // image (3x3) go out in the first 11 - NST K-steps of a chunk, and a 1x1 layer (one K-step per chunk) copies
  // chunk s + NST - 1 at K-step s, like its weight tile.
  constexpr int LA = NTAPS == 9 ? 1 : NST - 1;

  constexpr int NM = FN * FM, NR = FN + FM;   // MFMAs and fragment reads per 32-deep sub-step
  constexpr int ND = LW + XI;                 // LDS-DMA pieces a loading wave issues per K-step
  constexpr int FIRST_WAIT = LW * (NST - 1) + (NTAPS == 9 ? 0 : XI * (LA - 1));

  // ROLE 0: every wave loads and multiplies (NWL == 0); 1: consumer wave; 2: loader wave.  All roles pass the same
  // barriers: one after the prologue and one per K-step.  The wait in front of a barrier is the same instruction for
  // every role — a consumer has no LDS-DMA outstanding, a loader no LDS reads.
  auto run = [&](auto rolec) {
    constexpr int ROLE = decltype(rolec)::value;
    constexpr bool LOADS = ROLE != 1, MULS = ROLE != 2;
    f32x4_t acc[FN][FM];
    bf16x8_t wfA[FN], xfA[FM], wfB[FN], xfB[FM];      // fragment sets: sub-step 0 (A) and 1 (B) of a K-step

    // ---- prologue: chunk image i (i < LA) then weight tile i, i < NST (issue order = completion order) ----
    auto prologue_stage = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (LOADS) {
        if constexpr (i < LA) {
          if (x_needed(i)) {
#pragma unroll
            for (int k = 0; k < PXM; ++k) issue_x_piece(i, k);
          } else if constexpr (NTAPS == 1) {   // keep the count of the first wait: dummies into the sink
#pragma unroll
            for (int k = 0; k < PXM; ++k) glds16_u(zero_src, sSink);
          }
        }
        if constexpr (i < NST) stage_w(i, std::integral_constant<int, i % NTAPS>{});
      }
    };
    prologue_stage(std::integral_constant<int, 0>{});
    prologue_stage(std::integral_constant<int, 1>{});
    prologue_stage(std::integral_constant<int, 2>{});
    prologue_stage(std::integral_constant<int, 3>{});
    prologue_stage(std::integral_constant<int, 4>{});
    prologue_stage(std::integral_constant<int, 5>{});
    prologue_stage(std::integral_constant<int, 6>{});
    prologue_stage(std::integral_constant<int, 7>{});
    prologue_stage(std::integral_constant<int, 8>{});
    // chunk image 0 and weight tile 0 (everything issued after them may still be in flight)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(FIRST_WAIT) : "memory");
    int slot = 0;
    if constexpr (MULS) {
      const unsigned xb = sX + (unsigned)dh_off[0];
#pragma unroll
      for (int i = 0; i < FN; ++i) wfA[i] = lds_read_b128_u(sW + i * 512 + rdw_off[0]);
#pragma unroll
      for (int j = 0; j < FM; ++j) xfA[j] = lds_read_b128_u(xb + xoff[0][j]);
    }

    for (int pass = 0; pass < p.nt_per_wg; ++pass) {
      if constexpr (MULS) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }

      for (int c = 0; c < nchunks; ++c) {
        const int cf = pass * nchunks + c;
        const unsigned sXc = sX + (unsigned)(cf % p.xbuf) * xbytes;
        const unsigned sXn = sX + (unsigned)((cf + 1) % p.xbuf) * xbytes;   // next chunk's image (first tap of it)
        const bool load_ahead = x_needed(cf + LA);

        // One fragment read of a set, in the order the MFMAs of the next cluster consume them:
        // r = 0: W fragment 0; 1 .. FM: pixel fragments; FM + 1 .. FM + FN - 1: W fragments 1 ..
        auto read_frag = [&](bf16x8_t (&wf)[FN], bf16x8_t (&xf)[FM], int r, unsigned wbase, unsigned xbase, int v,
                             int kk) {
          if constexpr (ABL == 3) return;
          if (r == 0) wf[0] = lds_read_b128_u(wbase + rdw_off[kk]);
          else if (r <= FM) xf[r - 1] = lds_read_b128_u((xbase + xoff[v][r - 1]) ^ (kk ? 64u : 0u));
          else wf[r - FM] = lds_read_b128_u(wbase + (r - FM) * 512 + rdw_off[kk]);
        };
        auto mfma_n = [&](bf16x8_t (&wf)[FN], bf16x8_t (&xf)[FM], int n) {
          const int i = n / FM, j = n % FM;
          if constexpr (ABL == 1) { asm volatile("" ::"v"(wf[i]), "v"(xf[j])); }
          else acc[i][j] = mfma16<F16>(wf[i], xf[j], acc[i][j]);
        };
        // K-step (c, t), software-pipelined over its two 32-deep sub-steps and interleaved instruction by instruction:
        //   cluster A(s): the MFMAs of sub-step 0, with the fragment reads of sub-step 1 (set B) in their first gaps
        //   counted wait + barrier: weight tile s+1 and the activation pieces about to be read have landed for every
        //     loading wave; every consumer's reads of ring slot s % NST have returned (lgkmcnt(0)), so the slot is
        //     handed to tile s + NST right away
        //   cluster B(s): the MFMAs of sub-step 1, with the reads of A(s+1) in the first gaps (and, when the wave also
        //     loads, its LDS-DMA pieces in the later ones)
        // A fragment is consumed at least NM - NR + 1 MFMAs after its read was issued.  sched_barrier(0) pins the order.
        auto kstep = [&](auto tc) {
          constexpr int t = decltype(tc)::value;
          const int slot_s = slot;   // ring slot of K-step s: s % NST
          const unsigned sWs = sW + (unsigned)slot_s * W_BYTES;
          const unsigned xb = sXc + (unsigned)dh_off[t];
          if constexpr (MULS) {
#pragma unroll
            for (int n = 0; n < NM; ++n) {
              mfma_n(wfA, xfA, n);
              __builtin_amdgcn_sched_barrier(0);
              if (n < NR) {
                read_frag(wfB, xfB, n, sWs, xb, t % NV, 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int r = NM; r < NR; ++r) read_frag(wfB, xfB, r, sWs, xb, t % NV, 1);   // (only if NR > NM)
            __builtin_amdgcn_sched_barrier(0);
          }
          halo_wait_vm_and_barrier<LW * (NST - 2)>();
          slot = (slot + 1 == NST) ? 0 : slot + 1;
          constexpr int tn = (t + 1 == NTAPS) ? 0 : t + 1;
          const unsigned sWn = sW + (unsigned)slot * W_BYTES;
          const unsigned xbn = ((t + 1 == NTAPS) ? sXn : sXc) + (unsigned)dh_off[tn];
          // DMA piece d of this K-step: activation pieces first (they must be older than the weight tile, see above)
          auto dma_piece = [&](int d) {
            if constexpr (ABL == 2 || !LOADS) return;
            if (d < XI) {
              if constexpr (t < XSTEPS) { if (load_ahead) issue_x_piece(cf + LA, t * XI + d); }
            } else {
              w_piece(slot_s, d - XI);
            }
          };
          if constexpr (MULS) {
#pragma unroll
            for (int n = 0; n < NM; ++n) {
              mfma_n(wfB, xfB, n);
              __builtin_amdgcn_sched_barrier(0);
              if (n < NR) {
                read_frag(wfA, xfA, n, sWn, xbn, tn % NV, 0);
                __builtin_amdgcn_sched_barrier(0);
              } else if (LOADS && n - NR < ND) {
                dma_piece(n - NR);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int r = NM; r < NR; ++r) read_frag(wfA, xfA, r, sWn, xbn, tn % NV, 0);
#pragma unroll
            for (int d = (NM > NR ? NM - NR : 0); d < ND; ++d) dma_piece(d);   // pieces that found no MFMA gap
          } else {
#pragma unroll
            for (int d = 0; d < ND; ++d) dma_piece(d);
          }
          if constexpr (ABL != 2 && LOADS) w_advance(std::integral_constant<int, (t + NST) % NTAPS>{});
          __builtin_amdgcn_sched_barrier(0);
        };
        kstep(std::integral_constant<int, 0>{});
        if constexpr (NTAPS == 9) {
          kstep(std::integral_constant<int, 1>{});
          kstep(std::integral_constant<int, 2>{});
          kstep(std::integral_constant<int, 3>{});
          kstep(std::integral_constant<int, 4>{});
          kstep(std::integral_constant<int, 5>{});
          kstep(std::integral_constant<int, 6>{});
          kstep(std::integral_constant<int, 7>{});
          kstep(std::integral_constant<int, 8>{});
        }
      }

      if constexpr (MULS) {
      // ---- epilogue of this pass: the lane owns channels ch_base .. ch_base + CPL - 1 of its FM pixels ----
      const int ch_base = n_base + pass * BN + wn * WTN + fq * CPL;
      f32x4_t sc[FN], sh[FN];
  #pragma unroll
      for (int i = 0; i < FN; ++i) {
        sc[i] = p.scale ? *(const f32x4_t*)(p.scale + ch_base + i * 4) : (f32x4_t){1.f, 1.f, 1.f, 1.f};
        sh[i] = p.shift ? *(const f32x4_t*)(p.shift + ch_base + i * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
  #pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int64_t opx = opix[j];
        if (opx < 0) continue;
        int64_t apix = opx;
        if (p.addend_mode == TDN_ADD_UP2X || p.addend_mode == TDN_ADD_SUMPOOL2) {
          const int im = hdiv((int)opx, p.hw_mul, p.hw_shr);
          const int rem = (int)opx - im * (p.Hout * p.Wout);
          const int oh = hdiv(rem, p.w_mul, p.w_shr);
          const int ow = rem - oh * p.Wout;
          if (p.addend_mode == TDN_ADD_UP2X) apix = ((int64_t)im * p.addend_h + (oh >> 1)) * p.addend_w + (ow >> 1);
          else apix = ((int64_t)im * p.addend_h + 2 * oh) * p.addend_w + 2 * ow;
        }
        f32x4_t v[FN];
  #pragma unroll
        for (int i = 0; i < FN; ++i) v[i] = acc[i][j] * sc[i] + sh[i];
        if (p.addend_mode == TDN_ADD_SUMPOOL2) {
          // sum in a fixed order: (0,0) + (0,1) + (1,0) + (1,1)
          const bf16_t* ap = p.addend + apix * p.Cout + ch_base;
          const bf16_t* rows[4] = {ap, ap + p.Cout, ap + (int64_t)p.addend_w * p.Cout,
                                   ap + (int64_t)(p.addend_w + 1) * p.Cout};
  #pragma unroll
          for (int h = 0; h < FN / 2; ++h) {
            float acc8[8];
  #pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bf16x8_t r = *(const bf16x8_t*)(rows[q] + h * 8);
  #pragma unroll
              for (int e = 0; e < 8; ++e) acc8[e] = q == 0 ? elem_to_f32<F16>(r[e]) : acc8[e] + elem_to_f32<F16>(r[e]);
            }
  #pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * h][e] += acc8[e]; v[2 * h + 1][e] += acc8[4 + e]; }
          }
        } else if (p.addend_mode != TDN_ADD_NONE) {
          const bf16_t* ap = p.addend + apix * p.Cout + ch_base;
  #pragma unroll
          for (int h = 0; h < FN / 2; ++h) {
            const bf16x8_t r = *(const bf16x8_t*)(ap + h * 8);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * h][e] += elem_to_f32<F16>(r[e]);
              v[2 * h + 1][e] += elem_to_f32<F16>(r[4 + e]);
            }
          }
        }
        if (p.relu) {
  #pragma unroll
          for (int i = 0; i < FN; ++i)
  #pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
          if (p.relu == 2) {
  #pragma unroll
            for (int i = 0; i < FN; ++i)
  #pragma unroll
              for (int e = 0; e < 4; ++e) v[i][e] = relu6_top<F16>(v[i][e]);
          }
        }
        if (p.mask) {
  #pragma unroll
          for (int h = 0; h < FN / 2; ++h) {
            const bf16x8_t mk = *(const bf16x8_t*)(p.mask + opx * p.Cout + ch_base + h * 8);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * h][e] = (elem_to_f32<F16>(mk[e]) > 0.f) ? v[2 * h][e] : 0.f;
              v[2 * h + 1][e] = (elem_to_f32<F16>(mk[4 + e]) > 0.f) ? v[2 * h + 1][e] : 0.f;
            }
          }
        }
        if (p.out_f32) {
  #pragma unroll
          for (int i = 0; i < FN; ++i) *(f32x4_t*)((float*)p.out + opx * p.Cout + ch_base + i * 4) = v[i];
        } else {
  #pragma unroll
          for (int h = 0; h < FN / 2; ++h) {
            bf16x8_t o;
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = f32_to_elem<F16>(v[2 * h][e]);
              o[4 + e] = f32_to_elem<F16>(v[2 * h + 1][e]);
            }
            *(bf16x8_t*)(p.out + opx * p.Cout + ch_base + h * 8) = o;
          }
        }
      }
      }
    }
  };
  if constexpr (NWL == 0) {
    run(std::integral_constant<int, 0>{});
  } else {
    if (is_loader) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
  }
  // the dummy tail copies must have landed before the workgroup's LDS is handed on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static void hdiv_init(unsigned d, unsigned* mul, unsigned* shr) {
  if (d <= 1) { *mul = 0; *shr = 0; return; }
  unsigned lg = 0;
  while ((1ull << lg) < d) ++lg;
  const unsigned pw = 31 + lg;
  *mul = (unsigned)(((1ull << pw) + d - 1) / d);
  *shr = pw - 32;
}

// Tile configurations: {FM, FN, WM, WN, NST, XI9}; BM = 16 FM WM pixels, BN = 16 FN WN channels, WM WN waves, NST-deep
// weight ring, XI9 activation pieces per wave and K-step (3x3 only).  Separate tables for 3x3 and 1x1 layers: the
// LDS split between patch images and the weight ring differs (a 3x3 needs two patch images of 9 K-steps each, a 1x1
// streams one pixel-tile chunk per K-step through an NST - 1 deep ring of its own).
struct HaloCfg { int fm, fn, wm, wn, nst, xi9, nwl = 0; };
static const HaloCfg kHalo3[] = {
    {4, 4, 2, 2, 7, 2},   // 0  128 x 128, 4 waves, 6 weight tiles (96 KB) in flight: mid-size layers, one workgroup per CU
    {4, 4, 4, 2, 4, 1},   // 1  256 x 128, 8 waves: large-M layers
    {4, 4, 4, 1, 4, 2},   // 2  256 x  64, 4 waves: Cout = 64 at large M (layer1)
    {4, 2, 2, 2, 5, 1},   // 3  128 x  64, 4 waves (wave tile 64 x 32)
    {2, 4, 2, 2, 7, 1},   // 4   64 x 128, 4 waves (wave tile 32 x 64): smallest-M layers
    {4, 4, 2, 2, 3, 1},   // 5  configuration 0 with the shallow ring (A/B)
    {2, 4, 4, 2, 7, 1},   // 6  128 x 128, 8 waves (wave tile 32 x 64)
    {4, 4, 4, 2, 3, 1},   // 7  configuration 1 with the shallow ring (A/B)
    {4, 4, 2, 2, 7, 2, 4},   // 8  128 x 128, 4 consumer + 4 loader waves
    {2, 4, 2, 2, 7, 1, 4},   // 9   64 x 128, 4 consumer + 4 loader waves
    {4, 2, 2, 2, 5, 1, 4},   // 10 128 x  64, 4 consumer + 4 loader waves
    {4, 4, 2, 2, 4, 1, 4},   // 11 configuration 8 with a 4-deep ring
    {4, 4, 4, 2, 4, 2, 4},   // 12 256 x 128, 8 consumer + 4 loader waves (3 waves per SIMD: <= 168 registers)
    {4, 2, 2, 4, 4, 1, 8},   // 13 128 x 128, 8 consumer (wave tile 64 x 32) + 8 loader waves: <= 128 registers
    {4, 2, 2, 4, 4, 1, 4},   // 14 128 x 128, 8 consumer (wave tile 64 x 32) + 4 loader waves: <= 168 registers
    {2, 4, 4, 2, 4, 1, 4},   // 15 128 x 128, 8 consumer (wave tile 32 x 64) + 4 loader waves
};
static const HaloCfg kHalo1[] = {
    {4, 4, 2, 2, 5, 1},   // 0  128 x 128, 4 waves
    {2, 4, 4, 2, 5, 1},   // 1  128 x 128, 8 waves (wave tile 32 x 64)
    {4, 4, 4, 1, 4, 1},   // 2  256 x  64, 4 waves: Cout = 64
    {4, 2, 2, 2, 5, 1},   // 3  128 x  64, 4 waves
    {2, 4, 2, 2, 6, 1},   // 4   64 x 128, 4 waves
    {4, 4, 2, 2, 3, 1},   // 5  configuration 0 with the shallow ring (A/B)
    {4, 4, 4, 2, 3, 1},   // 6  256 x 128, 8 waves
    {2, 4, 2, 4, 4, 1},   // 7   64 x 256, 8 waves
    {4, 4, 2, 2, 5, 1, 4},   // 8  128 x 128, 4 consumer + 4 loader waves
    {2, 4, 2, 2, 6, 1, 4},   // 9   64 x 128, 4 consumer + 4 loader waves
    {4, 2, 2, 2, 5, 1, 4},   // 10 128 x  64, 4 consumer + 4 loader waves
    {4, 4, 2, 2, 3, 1, 4},   // 11 configuration 8 with the shallow ring
    {4, 2, 2, 2, 5, 1, 8},   // 12 128 x  64, 4 consumer + 8 loader waves
    {2, 4, 2, 2, 6, 1, 8},   // 13  64 x 128, 4 consumer + 8 loader waves
    {2, 2, 2, 2, 6, 1, 8},   // 14  64 x  64, 4 consumer + 8 loader waves (wave tile 32 x 32)
};
static const int kNumHalo3 = (int)(sizeof(kHalo3) / sizeof(kHalo3[0]));
static const int kNumHalo1 = (int)(sizeof(kHalo1) / sizeof(kHalo1[0]));
static inline const HaloCfg& halo_cfg(int k, int id) { return k == 3 ? kHalo3[id] : kHalo1[id]; }

struct HaloPlan {
  int cfg, TH, TW, pitch, hrows, xbuf, nt_per_wg;
  size_t lds;
};

static int halo_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

// Patch shape for a BM-pixel tile on an H x W map: least wasted work over TW (multiples of 2; widths that are not a
// multiple of 16 let fragments straddle patch rows and cost a wider LDS pitch), TH = BM / TW.  Cost per image =
// patches x (BM pixels of MFMA work + `xw` x haloed-patch rows of LDS-DMA).
static void halo_pick_patch(int H, int W, int halo, int bm, int* TH, int* TW, int* pitch, int* hrows) {
  double best = 1e30;
  for (int tw = 2; tw <= bm && tw <= ((W + 15) & ~15); tw += 2) {
    const int th = bm / tw;
    if (th < 1) break;
    if (th * tw < bm - 15) continue;          // leave at most one idle fragment per tile
    const int pt = tw + 2 * halo;               // even: tw and 2 * halo are
    const int rows = (th + 2 * halo) * pt;
    const double tiles = (double)((H + th - 1) / th) * ((W + tw - 1) / tw);
    const double straddle = (tw % 16 == 0) ? 1.0 : 1.10;   // 2-way LDS conflicts on the straddling fragments
    const double cost = tiles * (bm * straddle + 0.35 * rows);
    if (cost < best) { best = cost; *TH = th; *TW = tw; *pitch = pt; *hrows = (rows + 7) & ~7; }
  }
}

struct HaloShape {
  int N, H, W;          // output = input spatial size (stride-1 "same" conv), or output size for lin / sa = 2
  int Hin, Win;
  int Cin, Cout;        // GEMM K channels / N channels
  int k, sa, halo;
  int taps[9];
  int wt_row, Ktap;
};

// TDN_HALO: bit 0 = 3x3 layers, bit 1 = 1x1 layers through the halo kernel (default both); 0 = generic kernel only
static int halo_mode() { return halo_env_int("TDN_HALO", 3); }

// Choose configuration, patch, residency.  Returns false when the halo kernel does not apply.
static bool halo_make_plan(const HaloShape& s, HaloPlan* pl) {
  if (halo_mode() == 0) return false;
  if (halo_env_int("TDN_SPLITK", 0) != 0) return false;   // the opt-in cross-workgroup split-K lives in the generic kernel
  if (s.Cin % 64 != 0 || s.Cout % 64 != 0) return false;
  if (s.k == 3 && (s.sa != 1 || s.halo > 2)) return false;
  if (s.k == 3 && !(halo_mode() & 1)) return false;
  if (s.k == 1 && !(halo_mode() & 2)) return false;
  const long M = (long)s.N * s.H * s.W;
  // Which shapes the halo kernel takes by default — where it measured faster than the generic kernel on MI355X
  // (scripts/halo_bench.py, profiles/r03_halo_bench_*.log): 3x3 stride-1 layers with >= 128 output channels at mid
  // sizes (loader + consumer waves) and at the largest size (256 x 128 tile).  -1: leave the shape to conv_igemm.hip.
  // TDN_HALO_CFG3 / TDN_HALO_CFG1 force a configuration wherever it applies (tests, sweeps); 1x1 layers are never
  // taken by default: their 1..32-step K loops are dominated by per-workgroup fixed costs, which the generic kernel's
  // many small co-resident workgroups hide better (measured 1.3-2x slower here).
  int cfg = -1;
  if (s.k == 3 && s.Cout % 128 == 0) {
    // The largest layer (fpn_convs.0 and its dgrad, M = 134,400): the 256 x 128 tile is 5 % faster than the generic
    // 192 x 256 tile alone (167 vs 175 us) but the whole step is 0.7-0.9 % SLOWER with it (three interleaved pairs on
    // one box: 469.8 vs 473.1 img/s) — it runs beside the FPN's nine-tap weight-gradient group and co-runs worse.
    // TDN_HALO_BIG=1 selects it.
    if (M >= 100000) cfg = halo_env_int("TDN_HALO_BIG", 0) ? 1 : -1;
    else if (M >= 6000 && M < 30000) cfg = 11;      // layer2 per image, layer3 / P4 per batch: 128 x 128, 4 + 4 waves
    else if (M < 6000 && s.Cin >= 256 && (s.Cin >= 512 || M >= 3000))
      cfg = 10;   // layer3 / layer4 per image: 128 x 64, 4 + 4 waves (P5's 256-channel conv, M = 2,100: generic 13 vs 16 us)
  }
  cfg = halo_env_int(s.k == 3 ? "TDN_HALO_CFG3" : "TDN_HALO_CFG1", cfg);
  if (cfg < 0 || cfg >= (s.k == 3 ? kNumHalo3 : kNumHalo1)) return false;   // (-1: not taken)
  const HaloCfg& c = halo_cfg(s.k, cfg);
  const int bm = c.fm * c.wm * 16, bn = c.fn * c.wn * 16;
  const int nw = c.nwl > 0 ? c.nwl : c.wm * c.wn;   // loading waves
  if (s.Cout % bn != 0) return false;
  pl->cfg = cfg;
  if (s.k == 1) {
    pl->TH = 1; pl->TW = bm; pl->pitch = bm; pl->hrows = bm;
  } else {
    pl->TH = 0;
    halo_pick_patch(s.H, s.W, s.halo, bm, &pl->TH, &pl->TW, &pl->pitch, &pl->hrows);
    const int th = halo_env_int("TDN_HALO_TH", 0), tw = halo_env_int("TDN_HALO_TW", 0);
    if (th > 0 && tw > 0 && th * tw <= bm && tw % 2 == 0) {
      pl->TH = th; pl->TW = tw;
      pl->pitch = tw + 2 * s.halo;
      pl->hrows = ((th + 2 * s.halo) * pl->pitch + 7) & ~7;
    }
    if (pl->TH < 1) return false;
  }
  const int nchunks = s.Cin / 64;
  const size_t wring = (size_t)c.nst * bn * 128 + (size_t)(c.wm * c.wn + c.nwl) * 1024;   // ring + the dummy-copy sink
  const size_t xchunk = (size_t)pl->hrows * 128;
  const size_t budget = 160 * 1024;
  // Chunk images held in LDS.  3x3: two (the next chunk is copied under this one) unless all of them fit.  1x1:
  // a ring of NST - 1 (chunk s + NST - 1 is copied at K-step s) unless all of them fit; output-channel passes
  // (activation-stationary, nt > 1) need all of them.
  const int xmin = s.k == 3 ? (nchunks < 2 ? nchunks : 2) : (nchunks < c.nst - 1 ? nchunks : c.nst - 1);
  pl->xbuf = xmin;
  if (xchunk * nchunks + wring <= budget && halo_env_int("TDN_HALO_XBUF", 0) != 2) pl->xbuf = nchunks;
  const int ntn = s.Cout / bn;
  int nt = 1;
  if (s.k == 1 && pl->xbuf == nchunks && ntn >= 4 && M <= 20000) nt = 2;   // conv3 of layer3 / layer4: two passes
  nt = halo_env_int("TDN_HALO_NT", nt);
  if (nt < 1 || ntn % nt != 0 || pl->xbuf != nchunks) nt = 1;
  pl->nt_per_wg = nt;
  pl->lds = xchunk * pl->xbuf + wring;
  if (pl->lds > budget) return false;
  if (s.k == 3 && (pl->hrows / 8 + nw - 1) / nw > c.xi9 * (11 - c.nst)) return false;   // patch pieces per wave
  if ((int64_t)s.N * s.Hin * s.Win * s.Cin * 2 >= (1ll << 32) - 256) return false;   // 32-bit source offsets
  return true;
}

template <int FM, int FN, int WM, int WN, int NST, int NTAPS, int XI9, bool F16, int ABL = 0, int NWL = 0>
static int halo_launch(const HaloParams& p, size_t lds, hipStream_t stream) {
  static tdn_attr_once attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<FM, FN, WM, WN, NST, NTAPS, XI9, F16, ABL, NWL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(halo kernel LDS) failed: %s", hipGetErrorString(e));
    attr_once.mark();
  }
  dim3 grid(p.nwg_pad, 1, 1), block((WM * WN + NWL) * 64, 1, 1);
  TDN_LAUNCH((conv_halo_kernel<FM, FN, WM, WN, NST, NTAPS, XI9, F16, ABL, NWL>), grid, block, lds, stream, p);
  TDN_LAUNCH_CHECK();
  return 0;
}

template <bool F16>
static int halo_dispatch3(int cfg, const HaloParams& p, size_t lds, hipStream_t stream) {
  switch (cfg) {   // kHalo3
    case 0: return halo_launch<4, 4, 2, 2, 7, 9, 2, F16>(p, lds, stream);
    case 1: return halo_launch<4, 4, 4, 2, 4, 9, 1, F16>(p, lds, stream);
    case 2: return halo_launch<4, 4, 4, 1, 4, 9, 2, F16>(p, lds, stream);
    case 3: return halo_launch<4, 2, 2, 2, 5, 9, 1, F16>(p, lds, stream);
    case 4: return halo_launch<2, 4, 2, 2, 7, 9, 1, F16>(p, lds, stream);
    case 5: return halo_launch<4, 4, 2, 2, 3, 9, 1, F16>(p, lds, stream);
    case 6: return halo_launch<2, 4, 4, 2, 7, 9, 1, F16>(p, lds, stream);
    case 7: return halo_launch<4, 4, 4, 2, 3, 9, 1, F16>(p, lds, stream);
    case 8: return halo_launch<4, 4, 2, 2, 7, 9, 2, F16, 0, 4>(p, lds, stream);
    case 9: return halo_launch<2, 4, 2, 2, 7, 9, 1, F16, 0, 4>(p, lds, stream);
    case 10: return halo_launch<4, 2, 2, 2, 5, 9, 1, F16, 0, 4>(p, lds, stream);
    case 11: return halo_launch<4, 4, 2, 2, 4, 9, 1, F16, 0, 4>(p, lds, stream);
    case 12: return halo_launch<4, 4, 4, 2, 4, 9, 2, F16, 0, 4>(p, lds, stream);
    case 13: return halo_launch<4, 2, 2, 4, 4, 9, 1, F16, 0, 8>(p, lds, stream);
    case 14: return halo_launch<4, 2, 2, 4, 4, 9, 1, F16, 0, 4>(p, lds, stream);
    case 15: return halo_launch<2, 4, 4, 2, 4, 9, 1, F16, 0, 4>(p, lds, stream);
    default: TDN_CHECK(false, "bad 3x3 halo config %d", cfg); return -1;
  }
}
template <bool F16>
static int halo_dispatch1(int cfg, const HaloParams& p, size_t lds, hipStream_t stream) {
  switch (cfg) {   // kHalo1
    case 0: return halo_launch<4, 4, 2, 2, 5, 1, 1, F16>(p, lds, stream);
    case 1: return halo_launch<2, 4, 4, 2, 5, 1, 1, F16>(p, lds, stream);
    case 2: return halo_launch<4, 4, 4, 1, 4, 1, 1, F16>(p, lds, stream);
    case 3: return halo_launch<4, 2, 2, 2, 5, 1, 1, F16>(p, lds, stream);
    case 4: return halo_launch<2, 4, 2, 2, 6, 1, 1, F16>(p, lds, stream);
    case 5: return halo_launch<4, 4, 2, 2, 3, 1, 1, F16>(p, lds, stream);
    case 6: return halo_launch<4, 4, 4, 2, 3, 1, 1, F16>(p, lds, stream);
    case 7: return halo_launch<2, 4, 2, 4, 4, 1, 1, F16>(p, lds, stream);
    case 8: return halo_launch<4, 4, 2, 2, 5, 1, 1, F16, 0, 4>(p, lds, stream);
    case 9: return halo_launch<2, 4, 2, 2, 6, 1, 1, F16, 0, 4>(p, lds, stream);
    case 10: return halo_launch<4, 2, 2, 2, 5, 1, 1, F16, 0, 4>(p, lds, stream);
    case 11: return halo_launch<4, 4, 2, 2, 3, 1, 1, F16, 0, 4>(p, lds, stream);
    case 12: return halo_launch<4, 2, 2, 2, 5, 1, 1, F16, 0, 8>(p, lds, stream);
    case 13: return halo_launch<2, 4, 2, 2, 6, 1, 1, F16, 0, 8>(p, lds, stream);
    case 14: return halo_launch<2, 2, 2, 2, 6, 1, 1, F16, 0, 8>(p, lds, stream);
    default: TDN_CHECK(false, "bad 1x1 halo config %d", cfg); return -1;
  }
}

// Fills the launch parameters from the shape, the plan and the epilogue fields already in `p`.
static void halo_fill(HaloParams& p, const HaloShape& s, const HaloPlan& pl) {
  const HaloCfg& c = halo_cfg(s.k, pl.cfg);
  const int bm = c.fm * c.wm * 16, bn = c.fn * c.wn * 16;
  p.Hin = s.Hin; p.Win = s.Win; p.Cpix = s.Cin;
  p.Hout = s.H; p.Wout = s.W; p.Cout = s.Cout; p.M = s.N * s.H * s.W;
  p.sa = s.sa; p.halo = s.halo; p.lin = s.k == 1 ? 1 : 0;
  p.TH = pl.TH; p.TW = pl.TW; p.THh = pl.TH + 2 * s.halo; p.TWh = pl.TW + 2 * s.halo; p.tile_px = pl.TH * pl.TW;
  p.pitch = pl.pitch; p.hrows = pl.hrows; p.pieces = pl.hrows / 8;
  int mtiles;
  if (p.lin) {
    p.tiles_w = 1; p.tiles_img = 1;
    mtiles = (p.M + bm - 1) / bm;
  } else {
    p.tiles_w = (s.W + pl.TW - 1) / pl.TW;
    p.tiles_img = p.tiles_w * ((s.H + pl.TH - 1) / pl.TH);
    mtiles = s.N * p.tiles_img;
  }
  p.nt_per_wg = pl.nt_per_wg;
  p.ngroups_n = s.Cout / bn / pl.nt_per_wg;
  p.ntiles = mtiles * p.ngroups_n;
  p.nwg_pad = (p.ntiles + 7) & ~7;
  p.nchunks = s.Cin / 64; p.xbuf = pl.xbuf; p.x_resident = pl.xbuf >= p.nchunks ? 1 : 0;
  p.wt_row = s.wt_row; p.Ktap = s.Ktap;
  for (int i = 0; i < 9; ++i) p.taps[i] = s.taps[i];
  hdiv_init((unsigned)p.ngroups_n, &p.ng_mul, &p.ng_shr);
  hdiv_init((unsigned)p.tiles_img, &p.ti_mul, &p.ti_shr);
  hdiv_init((unsigned)p.tiles_w, &p.tw_mul, &p.tw_shr);
  hdiv_init((unsigned)p.pitch, &p.pit_mul, &p.pit_shr);
  hdiv_init((unsigned)p.TW, &p.TW_mul, &p.TW_shr);
  hdiv_init((unsigned)(s.H * s.W), &p.hw_mul, &p.hw_shr);
  hdiv_init((unsigned)s.W, &p.w_mul, &p.w_shr);
}

static int halo_run(HaloParams& p, const HaloShape& s, const HaloPlan& pl, int dtype, hipStream_t stream) {
  halo_fill(p, s, pl);
  // host-side shape checks: the kernel's LDS addressing assumes them
  TDN_CHECK(p.pitch % 2 == 0 && p.pitch >= p.TWh && p.hrows % 8 == 0 && p.hrows >= p.THh * p.pitch,
            "halo plan: bad patch geometry (pitch %d, rows %d)", p.pitch, p.hrows);
  if (s.k == 3) {
    for (int t = 0; t < 9; ++t)
      TDN_CHECK(((p.taps[t] >> 8) & 0xff) == ((p.taps[t % 3] >> 8) & 0xff) && (p.taps[t] >> 16) == t,
                "halo plan: taps are not (kh, kw) ordered");
#ifdef TDN_TRACE_BUILD   // timing-only ablation builds of two configurations (make TRACE=1): wrong results by construction
    if (const int abl = halo_env_int("TDN_HALO_ABL", 0)) {
      if (pl.cfg == 0 && abl == 1) return halo_launch<4, 4, 2, 2, 7, 9, 2, false, 1>(p, pl.lds, stream);
      if (pl.cfg == 0 && abl == 2) return halo_launch<4, 4, 2, 2, 7, 9, 2, false, 2>(p, pl.lds, stream);
      if (pl.cfg == 0 && abl == 3) return halo_launch<4, 4, 2, 2, 7, 9, 2, false, 3>(p, pl.lds, stream);
      if (pl.cfg == 11 && abl == 1) return halo_launch<4, 4, 2, 2, 4, 9, 1, false, 1, 4>(p, pl.lds, stream);
      if (pl.cfg == 11 && abl == 2) return halo_launch<4, 4, 2, 2, 4, 9, 1, false, 2, 4>(p, pl.lds, stream);
      if (pl.cfg == 11 && abl == 3) return halo_launch<4, 4, 2, 2, 4, 9, 1, false, 3, 4>(p, pl.lds, stream);
      if (pl.cfg == 1 && abl == 1) return halo_launch<4, 4, 4, 2, 4, 9, 1, false, 1>(p, pl.lds, stream);
      if (pl.cfg == 1 && abl == 2) return halo_launch<4, 4, 4, 2, 4, 9, 1, false, 2>(p, pl.lds, stream);
      if (pl.cfg == 1 && abl == 3) return halo_launch<4, 4, 4, 2, 4, 9, 1, false, 3>(p, pl.lds, stream);
    }
#endif
    // TDN_TAG_DOMINANT (set by bench.py around exactly the launches it brackets with HIP events): the same code under
    // a symbol of its own (template argument ABL = 9 changes nothing but the name), so that rocprofv3 --stats lists
    // those launches — neck.fpn_convs.0 forward and its dgrad — on a line of their own
    if (pl.cfg == 1 && dtype != TDN_F16 && p.M >= 100000 && p.Cout == 256 && p.nchunks == 4 && getenv("TDN_TAG_DOMINANT"))
      return halo_launch<4, 4, 4, 2, 4, 9, 1, false, 9>(p, pl.lds, stream);
    if (dtype == TDN_F16) return halo_dispatch3<true>(pl.cfg, p, pl.lds, stream);
    return halo_dispatch3<false>(pl.cfg, p, pl.lds, stream);
  }
  if (dtype == TDN_F16) return halo_dispatch1<true>(pl.cfg, p, pl.lds, stream);
  return halo_dispatch1<false>(pl.cfg, p, pl.lds, stream);
}

static void halo_epilogue(HaloParams& p, const tdn_epilogue* ep) {
  p.scale = nullptr; p.shift = nullptr; p.addend = nullptr; p.mask = nullptr;
  p.addend_mode = TDN_ADD_NONE; p.addend_h = 0; p.addend_w = 0; p.relu = 0; p.out_f32 = 0;
  if (!ep) return;
  p.out_f32 = ep->out_f32 ? 1 : 0;
  p.scale = ep->scale; p.shift = ep->shift; p.relu = ep->relu;
  p.mask = (const bf16_t*)ep->mask_src;
  if (ep->addend_mode != TDN_ADD_NONE) {
    p.addend = (const bf16_t*)ep->addend;
    p.addend_mode = ep->addend_mode;
    p.addend_h = ep->addend_h;
    p.addend_w = ep->addend_w;
  }
}

static inline int halo_pack_tap(int dh, int dw, int widx) { return (dh + 64) | ((dw + 64) << 8) | (widx << 16); }

// Forward conv through the halo kernel.  Returns 1 if launched, 0 if the shape is left to conv_igemm.hip, < 0 on error.
// (The epilogue was validated by the caller: fill_epilogue of conv_igemm.hip.)
int tdn_halo_conv_fwd(const void* x, const void* w_fwd, void* y, int N, int H, int W, int Cin, int Cout, int k,
                      int stride, int pad, const tdn_epilogue* ep, int dtype, hipStream_t stream) {
  if (k == 3 && stride != 1) return 0;
  HaloShape s;
  const int d = k == 3 ? pad : 1;
  s.N = N; s.Hin = H; s.Win = W; s.Cin = Cin; s.Cout = Cout; s.k = k; s.sa = stride;
  s.H = k == 3 ? H : (H - 1) / stride + 1;
  s.W = k == 3 ? W : (W - 1) / stride + 1;
  s.halo = k == 3 ? d : 0;
  s.wt_row = k * k * Cin; s.Ktap = Cin;
  for (int i = 0; i < 9; ++i) s.taps[i] = halo_pack_tap(0, 0, 0);
  if (k == 3)
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) s.taps[kh * 3 + kw] = halo_pack_tap(kh * d - pad, kw * d - pad, kh * 3 + kw);
  HaloPlan pl;
  if (!halo_make_plan(s, &pl)) return 0;
  HaloParams p;
  halo_epilogue(p, ep);
  p.in = (const bf16_t*)x; p.wt = (const bf16_t*)w_fwd; p.out = (bf16_t*)y;
  const int rc = halo_run(p, s, pl, dtype, stream);
  return rc < 0 ? rc : 1;
}

// Stride-1 input gradient: the same conv over g with the per-tap transposed weights w_dgrad[Cin][kh][kw][Cout]
// (tap (kh, kw) reads g at (h + pad - kh d, w + pad - kw d)).
int tdn_halo_conv_dgrad(const void* g, const void* w_dgrad, void* dx, int N, int H, int W, int Cin, int Cout, int k,
                        int stride, int pad, const tdn_epilogue* ep, int dtype, hipStream_t stream) {
  if (stride != 1) return 0;
  HaloShape s;
  const int d = k == 3 ? pad : 1;
  s.N = N; s.H = H; s.W = W; s.Hin = H; s.Win = W; s.Cin = Cout; s.Cout = Cin; s.k = k; s.sa = 1;
  s.halo = k == 3 ? d : 0;
  s.wt_row = k * k * Cout; s.Ktap = Cout;
  for (int i = 0; i < 9; ++i) s.taps[i] = halo_pack_tap(0, 0, 0);
  if (k == 3)
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) s.taps[kh * 3 + kw] = halo_pack_tap(pad - kh * d, pad - kw * d, kh * 3 + kw);
  HaloPlan pl;
  if (!halo_make_plan(s, &pl)) return 0;
  HaloParams p;
  halo_epilogue(p, ep);
  p.in = (const bf16_t*)g; p.wt = (const bf16_t*)w_dgrad; p.out = (bf16_t*)dx;
  const int rc = halo_run(p, s, pl, dtype, stream);
  return rc < 0 ? rc : 1;
}

// What tdn_conv2d_plan reports for a shape the halo kernel takes: o[3..6] = BM, BN, 64, workgroups.
int tdn_halo_plan(int kind, int N, int H, int W, int Cin, int Cout, int k, int stride, int pad, int32_t* o) {
  HaloShape s;
  const int d = k == 3 ? pad : 1;
  if (kind == 0) {
    if (k == 3 && stride != 1) return 0;
    s.N = N; s.Hin = H; s.Win = W; s.Cin = Cin; s.Cout = Cout; s.k = k; s.sa = stride;
    s.H = k == 3 ? H : (H - 1) / stride + 1;
    s.W = k == 3 ? W : (W - 1) / stride + 1;
  } else {
    if (stride != 1) return 0;
    s.N = N; s.H = H; s.W = W; s.Hin = H; s.Win = W; s.Cin = Cout; s.Cout = Cin; s.k = k; s.sa = 1;
  }
  s.halo = k == 3 ? d : 0;
  HaloPlan pl;
  if (!halo_make_plan(s, &pl)) return 0;
  HaloParams p;
  for (int i = 0; i < 9; ++i) s.taps[i] = halo_pack_tap(0, 0, 0);
  s.wt_row = 0; s.Ktap = 0;
  halo_fill(p, s, pl);
  const HaloCfg& c = halo_cfg(k, pl.cfg);
  o[3] = c.fm * c.wm * 16; o[4] = c.fn * c.wn * 16; o[5] = 64; o[6] = p.nwg_pad; o[7] = 1;
  o[8] = 100 + pl.cfg;                      // grid_z slot: 100 + halo configuration id (generic kernel: split count)
  o[11] = pl.TH * 1000 + pl.TW;             // split-K slot: the patch, TH * 1000 + TW (1x1: 1000 + BM)
  o[12] = pl.xbuf * 100 + pl.nt_per_wg;     // chunk images in LDS * 100 + output-channel passes per workgroup
  return 1;
}
