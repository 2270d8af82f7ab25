// Box ops: anchor grid, pairwise IoU, greedy NMS.  HBM / latency-bound integer-index kernels.
//
// The reference has no implementation (core/__init__.py is empty); semantics are SURVEY.md Appendix B,
// consistent with the conventions the reference does pin: inclusive "+1" pixel boxes
// (datasets/utils/bbox.py:39,375-377), xyxy float32, x-fastest grid enumeration
// (datasets/dataset_transforms.py:120-131).  Arithmetic is strict IEEE fp32 in the oracle's operation
// order (this file is compiled with -ffp-contract=off; divisions are __fdiv_rn) so that results are
// bit-identical to oracle/box_ref.c.
#include "common.h"
#include <string.h>

// ---- anchor grid -----------------------------------------------------------------------------------
__global__ void anchor_grid_kernel(const float* __restrict__ base, int A, int featH, int featW, int stride,
                                   int valid_h, int valid_w, float* __restrict__ anchors, uint8_t* valid) {
  const int total = featH * featW * A;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = i % A;
    const int cell = i / A;
    const int x = cell % featW, y = cell / featW;
    const float sx = (float)(x * stride), sy = (float)(y * stride);
    const f32x4_t b = *(const f32x4_t*)(base + a * 4);
    f32x4_t o;
    o[0] = b[0] + sx;
    o[1] = b[1] + sy;
    o[2] = b[2] + sx;
    o[3] = b[3] + sy;
    *(f32x4_t*)(anchors + (int64_t)i * 4) = o;
    if (valid) valid[i] = (x < valid_w && y < valid_h) ? 1 : 0;
  }
}

extern "C" int tdn_anchor_grid(const float* base_anchors, int A, int featH, int featW, int stride, int valid_h,
                               int valid_w, float* anchors, uint8_t* valid, void* stream) {
  TDN_CHECK(A > 0 && featH >= 0 && featW >= 0 && stride > 0, "tdn_anchor_grid: bad shape");
  const int64_t total = (int64_t)featH * featW * A;
  TDN_CHECK(total < (1ll << 30), "tdn_anchor_grid: too many anchors");
  if (total == 0) return 0;
  TDN_CHECK(base_anchors && anchors, "tdn_anchor_grid: NULL pointer");
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  TDN_LAUNCH(anchor_grid_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, base_anchors, A, featH,
                     featW, stride, valid_h, valid_w, anchors, valid);
  TDN_LAUNCH_CHECK();
  return 0;
}

// All levels of a pyramid in ONE launch (five launches of 4..70 us worth of work cost ~9 us each alone): level
// descriptors in the kernel-argument block, outputs concatenated level after level (order (y, x, anchor) inside a
// level, exactly as tdn_anchor_grid writes them).
constexpr int ANCHOR_MAXL = 8;
struct AnchorLevels {
  int nlevels, total;
  int start[ANCHOR_MAXL + 1];      // first anchor of level l; start[nlevels] = total
  int A[ANCHOR_MAXL], featW[ANCHOR_MAXL], stride[ANCHOR_MAXL], valid_h[ANCHOR_MAXL], valid_w[ANCHOR_MAXL];
  const float* base[ANCHOR_MAXL];
};

__global__ void anchor_pyramid_kernel(const AnchorLevels L, float* __restrict__ anchors, uint8_t* valid) {
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < L.total; g += gridDim.x * blockDim.x) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < ANCHOR_MAXL; ++k) l += (k < L.nlevels && g >= L.start[k]) ? 1 : 0;
    const int i = g - L.start[l];
    const int A = L.A[l], featW = L.featW[l], stride = L.stride[l];
    const int a = i % A;
    const int cell = i / A;
    const int x = cell % featW, y = cell / featW;
    const float sx = (float)(x * stride), sy = (float)(y * stride);
    const f32x4_t b = *(const f32x4_t*)(L.base[l] + a * 4);
    f32x4_t o;
    o[0] = b[0] + sx;
    o[1] = b[1] + sy;
    o[2] = b[2] + sx;
    o[3] = b[3] + sy;
    *(f32x4_t*)(anchors + (int64_t)g * 4) = o;
    if (valid) valid[g] = (x < L.valid_w[l] && y < L.valid_h[l]) ? 1 : 0;
  }
}

extern "C" int tdn_anchor_pyramid(const tdn_anchor_level* levels, int nlevels, float* anchors, uint8_t* valid,
                                  void* stream) {
  TDN_CHECK(levels != nullptr && nlevels > 0 && nlevels <= ANCHOR_MAXL, "tdn_anchor_pyramid: 1..%d levels",
            ANCHOR_MAXL);
  AnchorLevels L;
  memset(&L, 0, sizeof(L));
  L.nlevels = nlevels;
  int64_t total = 0;
  for (int l = 0; l < nlevels; ++l) {
    const tdn_anchor_level& s = levels[l];
    TDN_CHECK(s.A > 0 && s.featH >= 0 && s.featW >= 0 && s.stride > 0, "tdn_anchor_pyramid: level %d: bad shape", l);
    TDN_CHECK(s.base_anchors != nullptr, "tdn_anchor_pyramid: level %d: NULL base anchors", l);
    L.start[l] = (int)total;
    L.A[l] = s.A; L.featW[l] = s.featW > 0 ? s.featW : 1; L.stride[l] = s.stride;
    L.valid_h[l] = s.valid_h; L.valid_w[l] = s.valid_w; L.base[l] = s.base_anchors;
    total += (int64_t)s.featH * s.featW * s.A;
    TDN_CHECK(total < (1ll << 30), "tdn_anchor_pyramid: too many anchors");
  }
  for (int l = nlevels; l <= ANCHOR_MAXL; ++l) L.start[l] = (int)total;
  L.total = (int)total;
  if (total == 0) return 0;
  TDN_CHECK(anchors != nullptr, "tdn_anchor_pyramid: NULL output");
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  TDN_LAUNCH(anchor_pyramid_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, L, anchors, valid);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- IoU -------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_area(const f32x4_t b) {
  return __fmul_rn(__fadd_rn(__fsub_rn(b[2], b[0]), 1.0f), __fadd_rn(__fsub_rn(b[3], b[1]), 1.0f));
}

// IoU of boxes a, b with both areas supplied (box_area is a pure function of its box)
__device__ __forceinline__ float box_iou2(const f32x4_t a, const float area_a, const f32x4_t b, const float area_b) {
  const float ltx = fmaxf(a[0], b[0]), lty = fmaxf(a[1], b[1]);
  const float rbx = fminf(a[2], b[2]), rby = fminf(a[3], b[3]);
  const float w = fmaxf(__fadd_rn(__fsub_rn(rbx, ltx), 1.0f), 0.0f);
  const float h = fmaxf(__fadd_rn(__fsub_rn(rby, lty), 1.0f), 0.0f);
  const float inter = __fmul_rn(w, h);
  const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
  return __fdiv_rn(inter, uni);
}

// A thread owns four consecutive columns (boxes of `b` and their areas stay in registers) and walks IOU_ROWS rows of
// `a` (workgroup-uniform: scalar loads): no index division, 16 B of `b` traffic per IOU_ROWS outputs, and a wave
// writes 1 KB of contiguous output per row.
constexpr int IOU_ROWS = 16;
__global__ __launch_bounds__(256) void iou_pairwise_kernel(const float* __restrict__ a, int N,
                                                           const float* __restrict__ b, int M,
                                                           float* __restrict__ out) {
  const int M4 = (M + 3) >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;   // column quad
  if (q >= M4) return;
  const int j0 = q * 4;
  const bool vec_ok = (M & 3) == 0;
  f32x4_t bb[4];
  float ab[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = j0 + e;
    bb[e] = (j < M) ? *(const f32x4_t*)(b + (int64_t)j * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    ab[e] = box_area(bb[e]);
  }
  for (int ib = blockIdx.y * IOU_ROWS; ib < N; ib += gridDim.y * IOU_ROWS)
  for (int i = ib; i < min(N, ib + IOU_ROWS); ++i) {
    const f32x4_t ba = *(const f32x4_t*)(a + (int64_t)i * 4);
    const float area_a = box_area(ba);
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (j0 + e < M) ? box_iou2(ba, area_a, bb[e], ab[e]) : 0.f;
    float* o = out + (int64_t)i * M + j0;
    if (vec_ok) {
      *(f32x4_t*)o = (f32x4_t){r[0], r[1], r[2], r[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j0 + e < M) o[e] = r[e];
    }
  }
}

extern "C" int tdn_bbox_iou_pairwise(const float* a, int N, const float* b, int M, float* iou, void* stream) {
  TDN_CHECK(N >= 0 && M >= 0, "tdn_bbox_iou_pairwise: negative size");
  if (N == 0 || M == 0) return 0;
  TDN_CHECK(a && b && iou, "tdn_bbox_iou_pairwise: NULL pointer");
  const int M4 = (M + 3) / 4;
  int gy = (N + IOU_ROWS - 1) / IOU_ROWS;
  if (gy > 65535) gy = 65535;   // the kernel strides over row blocks
  TDN_LAUNCH(iou_pairwise_kernel, dim3((M4 + 255) / 256, gy), dim3(256), 0, (hipStream_t)stream, a, N, b, M, iou);
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- NMS -------------------------------------------------------------------------------------------
// (1) rank[i] = #{ j : s[j] > s[i]  or  (s[j] == s[i] and j < i) }  -> stable descending order.
__global__ void nms_rank_kernel(const float* __restrict__ scores, int N, int jchunk, int* rank) {
  __shared__ float sj[1024];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float si = (i < N) ? scores[i] : 0.f;
  const int jb = blockIdx.y * jchunk, je = min(N, jb + jchunk);
  int cnt = 0;
  for (int base = jb; base < je; base += 1024) {
    const int n = min(1024, je - base);
    for (int k = threadIdx.x; k < n; k += blockDim.x) sj[k] = scores[base + k];
    __syncthreads();
    if (i < N) {
      for (int k = 0; k < n; ++k) {
        const float s = sj[k];
        const int j = base + k;
        cnt += (s > si || (s == si && j < i)) ? 1 : 0;
      }
    }
    __syncthreads();
  }
  if (i < N && cnt) atomicAdd(&rank[i], cnt);
}

__global__ void nms_scatter_kernel(const float* __restrict__ boxes, const int* __restrict__ rank, int N, int* order,
                                   float* sboxes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int r = rank[i];
  order[r] = i;
  *(f32x4_t*)(sboxes + (int64_t)r * 4) = *(const f32x4_t*)(boxes + (int64_t)i * 4);
}

// (2) 64-bit suppression words: mask[i][cb] bit b  <=>  j = cb*64+b > i  and  iou(i, j) > thr   (sorted order).
//     One wave per 64 x 64 block: lane l owns COLUMN box j = cb*64 + l (its area computed once); for every row r of
//     the block the 64 predicates of the lanes are gathered into the row's word with a wavefront ballot, and lane r
//     keeps it.  Blocks under the diagonal are never read by the scan and are skipped.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, int N, float thr, int nblk,
                                                      unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  __shared__ f32x4_t rbox[64];
  const int t = threadIdx.x;
  const int ir = rb * 64 + t;
  rbox[t] = (ir < N) ? *(const f32x4_t*)(sboxes + (int64_t)ir * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int j = cb * 64 + t;
  const bool jvalid = j < N;
  const f32x4_t bj = jvalid ? *(const f32x4_t*)(sboxes + (int64_t)j * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float area_j = box_area(bj);
  __syncthreads();
  unsigned long long mine = 0ull;
  const int nrow = min(64, N - rb * 64);
  for (int r = 0; r < nrow; ++r) {
    const int i = rb * 64 + r;
    const f32x4_t bi = rbox[r];                       // LDS broadcast
    const float area_i = box_area(bi);
    // argument order as in the oracle's nms: iou(box_i, box_j)
    const bool hit = jvalid && j > i && box_iou2(bi, area_i, bj, area_j) > thr;
    const unsigned long long word = __ballot(hit);
    mine = (t == r) ? word : mine;
  }
  if (ir < N) mask[(int64_t)ir * nblk + cb] = mine;
}

// (3) serial keep scan by ONE wave: 64-box chunks; within a chunk the diagonal words are resolved with
//     readlane + bit ops, then the kept rows are OR-ed into the running removal bitmap (lane = word).
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ order, int N, int nblk,
                                                      uint8_t* keep, int64_t* kept_idx, int* num_kept) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];  // [nblk] removal bitmap
  const int lane = threadIdx.x;
  for (int w = lane; w < nblk; w += 64) remv[w] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int c = 0; c < nblk; ++c) {
    const int i = c * 64 + lane;
    const unsigned long long diag = (i < N) ? mask[(int64_t)i * nblk + c] : 0ull;
    const int nvalid = min(64, N - c * 64);
    unsigned long long alive = ~remv[c];
    if (nvalid < 64) alive &= (1ull << nvalid) - 1ull;
    unsigned long long keepbits = 0ull;
    const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
    while (alive) {
      const int b = __builtin_ctzll(alive);
      keepbits |= 1ull << b;
      alive &= ~(1ull << b);
      const unsigned lo = __builtin_amdgcn_readlane(dlo, b), hi = __builtin_amdgcn_readlane(dhi, b);
      alive &= ~(((unsigned long long)hi << 32) | lo);
    }
    // outputs for this chunk
    if (i < N) {
      const bool k = (keepbits >> lane) & 1ull;
      const int oi = order[i];
      keep[oi] = k ? 1 : 0;
      if (k) {
        const int pos = cnt + __builtin_popcountll(keepbits & ((1ull << lane) - 1ull));
        kept_idx[pos] = (int64_t)oi;
      }
    }
    cnt += __builtin_popcountll(keepbits);
    // OR kept rows into remv for the remaining words; lane handles words c+1+lane, +64, ...
    for (int w = c + 1 + lane; w < nblk; w += 64) {
      unsigned long long acc = 0ull;
      unsigned long long kb = keepbits;
      while (kb) {
        // up to 4 independent row loads in flight
        int b0 = __builtin_ctzll(kb); kb &= kb - 1;
        unsigned long long v0 = mask[(int64_t)(c * 64 + b0) * nblk + w], v1 = 0, v2 = 0, v3 = 0;
        if (kb) { int b1 = __builtin_ctzll(kb); kb &= kb - 1; v1 = mask[(int64_t)(c * 64 + b1) * nblk + w]; }
        if (kb) { int b2 = __builtin_ctzll(kb); kb &= kb - 1; v2 = mask[(int64_t)(c * 64 + b2) * nblk + w]; }
        if (kb) { int b3 = __builtin_ctzll(kb); kb &= kb - 1; v3 = mask[(int64_t)(c * 64 + b3) * nblk + w]; }
        acc |= (v0 | v1) | (v2 | v3);
      }
      remv[w] |= acc;
    }
    __syncthreads();  // single wave: orders this chunk's remv writes before the next chunk's read
  }
  for (int k = cnt + lane; k < N; k += 64) kept_idx[k] = -1;
  if (lane == 0) *num_kept = cnt;
}

// (3') the same scan by a 1024-thread workgroup, in super-chunks of 16 chunks (1024 boxes):
//   A  all threads: the super-chunk's 1024 x 16-word diagonal band of the mask -> LDS (128 KB), one row per thread;
//   B  wave 0: the 16 chunks in order, entirely from LDS.  Resolving a chunk's diagonal word is serial only over the
//      boxes that overlap a LATER box of the same chunk (rows with a non-zero diagonal word — found with one wavefront
//      ballot; a box with an all-zero row suppresses nothing inside the chunk, so its fate is simply its bit once the
//      earlier non-zero rows have been applied): ~1 iteration per chunk on sparse inputs instead of one per kept box.
//      Then keep flags and compacted indices are written (a lane's slot = kept boxes before the chunk + popcount of the
//      kept bits below the lane, i.e. its rank in the ballot), and the kept rows are OR-ed into the removal words of
//      the band's later chunks: lane = (word, quarter of the rows), quarters combined with two wave shuffles;
//   C  all threads: wave w ORs the kept rows of chunk w into the removal words past the band — lanes = consecutive
//      words (coalesced), eight rows in flight — and adds its result to the bitmap with one LDS atomic per word.
// The single-wave kernel above pays a global-memory round trip per chunk on its critical path (157 at N = 10k) and
// one serial step per kept box; here the serial part touches LDS only and global latency is paid twice per 1024 boxes.
constexpr int NMS_SC = 16;                       // chunks per super-chunk
constexpr int NMS_SCROWS = NMS_SC * 64;          // 1024 rows = threads
constexpr int NMS_DPITCH = NMS_SCROWS + 1;       // words; odd pitch: the 16 words of one row sit in 16 different banks
constexpr int NMS_BLOCK_MAX_NBLK = 3000;         // removal bitmap + band must fit 160 KB of LDS

__global__ __launch_bounds__(1024) void nms_scan_block_kernel(const unsigned long long* __restrict__ mask,
                                                              const int* __restrict__ order, int N, int nblk,
                                                              uint8_t* keep, int64_t* kept_idx, int* num_kept) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
  unsigned long long* remv = sm;                                  // [nblk]
  unsigned long long* D = sm + ((nblk + 1) & ~1);                 // [NMS_SC][NMS_DPITCH]
  unsigned long long* kbits = D + NMS_SC * NMS_DPITCH;            // [NMS_SC]
  int* ord = (int*)(kbits + NMS_SC);                              // [NMS_SCROWS] original indices of the band's boxes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int w = tid; w < nblk; w += 1024) remv[w] = 0ull;
  int cnt = 0;                                                    // wave 0 only
  const int nsuper = (nblk + NMS_SC - 1) / NMS_SC;
  for (int s = 0; s < nsuper; ++s) {
    const int c0 = s * NMS_SC;
    const int nch = min(NMS_SC, nblk - c0);
    const int row = c0 * 64 + tid;
    // ---- A: diagonal band -> LDS ----
    // 16 lanes read the 16 words of one row (128 contiguous bytes): a wave instruction covers 4 rows = 4-8 cache
    // lines, not 64 rows of one word each
    {
      const int k = tid & 15;
#pragma unroll 4
      for (int j = 0; j < NMS_SC; ++j) {
        const int r = (tid >> 4) + 64 * j;              // row of the band
        const int grow = c0 * 64 + r;
        unsigned long long v = 0ull;
        if (grow < N && k < nch) v = mask[(int64_t)grow * nblk + c0 + k];
        D[k * NMS_DPITCH + r] = v;
      }
      ord[tid] = (row < N) ? order[row] : 0;          // no global load is left on wave 0's serial path below
    }
    __syncthreads();
    // ---- B: serial resolution of the band, wave 0 ----
    if (wave == 0) {
      const int tw = lane & 15, part = lane >> 4;     // band OR: target word offset, quarter of the chunk's rows
      for (int kc = 0; kc < nch; ++kc) {
        const int c = c0 + kc;
        const int i = c * 64 + lane;
        const unsigned long long diag = D[kc * NMS_DPITCH + kc * 64 + lane];
        const int nvalid = min(64, N - c * 64);
        unsigned long long alive = ~remv[c];
        if (nvalid < 64) alive &= (1ull << nvalid) - 1ull;
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        unsigned long long todo = alive & __ballot(diag != 0ull);   // boxes that can suppress inside this chunk
        while (todo) {
          const int b = __builtin_ctzll(todo);
          const unsigned lo = __builtin_amdgcn_readlane(dlo, b), hi = __builtin_amdgcn_readlane(dhi, b);
          const unsigned long long rowbits = ((unsigned long long)hi << 32) | lo;   // only bits above b
          alive &= ~rowbits;
          todo &= ~rowbits;
          todo &= todo - 1;                           // b itself is done (and kept)
        }
        const unsigned long long keepbits = alive;
        if (i < N) {
          const bool k = (keepbits >> lane) & 1ull;
          const int oi = ord[kc * 64 + lane];
          keep[oi] = k ? 1 : 0;
          if (k) kept_idx[cnt + __builtin_popcountll(keepbits & ((1ull << lane) - 1ull))] = (int64_t)oi;
        }
        cnt += __builtin_popcountll(keepbits);
        if (lane == 0) kbits[kc] = keepbits;
        // kept rows -> removal words of the band's later chunks
        {
          const int k2 = kc + 1 + tw;
          unsigned long long acc = 0ull;
          if (k2 < nch) {
            unsigned long long kb = (keepbits >> (part * 16)) & 0xFFFFull;
            const unsigned long long* col = D + k2 * NMS_DPITCH + kc * 64 + part * 16;
            while (kb) {
              const int b0 = __builtin_ctzll(kb); kb &= kb - 1;
              unsigned long long v0 = col[b0], v1 = 0ull, v2 = 0ull, v3 = 0ull;
              if (kb) { const int b1 = __builtin_ctzll(kb); kb &= kb - 1; v1 = col[b1]; }
              if (kb) { const int b2 = __builtin_ctzll(kb); kb &= kb - 1; v2 = col[b2]; }
              if (kb) { const int b3 = __builtin_ctzll(kb); kb &= kb - 1; v3 = col[b3]; }
              acc |= (v0 | v1) | (v2 | v3);
            }
          }
          acc |= __shfl_xor(acc, 16, 64);
          acc |= __shfl_xor(acc, 32, 64);
          if (part == 0 && k2 < nch && acc) remv[c0 + k2] |= acc;
        }
        __threadfence_block();                        // the removal words written above are read by every lane next round
      }
    }
    __syncthreads();
    // ---- C: kept rows of the band -> removal words past the band ----
    // wave w takes the kept rows of chunk w, lanes take consecutive words (512-byte coalesced reads), sixteen rows in
    // flight per lane; one LDS atomic per lane and word slot at the end
    const int wbeg = c0 + nch;
    if (wbeg < nblk && wave < nch) {
      const unsigned long long keepbits = kbits[wave];
      const unsigned long long* base = mask + (int64_t)(c0 + wave) * 64 * nblk;
      for (int wb = wbeg; wb < nblk; wb += 64) {
        const int w = wb + lane;
        if (w < nblk) {
          unsigned long long acc = 0ull, kb = keepbits;
          while (kb) {
            unsigned long long v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              v[e] = 0ull;
              if (kb) {
                const int b = __builtin_ctzll(kb);
                kb &= kb - 1;
                v[e] = base[(int64_t)b * nblk + w];
              }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc |= v[e];
          }
          if (acc) atomicOr(&remv[w], acc);
        }
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
    for (int k = cnt + lane; k < N; k += 64) kept_idx[k] = -1;
    if (lane == 0) *num_kept = cnt;
  }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~255ll; }

extern "C" int64_t tdn_nms_workspace(int N) {
  if (N <= 0) return 256;
  const int64_t nblk = (N + 63) / 64;
  return align256((int64_t)N * 4) * 2 + align256((int64_t)N * 16) + align256((int64_t)N * nblk * 8) + 256;
}

extern "C" int tdn_nms(const float* boxes, const float* scores, int N, float iou_thr, uint8_t* keep,
                       int64_t* kept_idx, int32_t* num_kept, void* workspace, int64_t workspace_bytes,
                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  TDN_CHECK(N >= 0 && N <= 64 * 8000, "tdn_nms: N=%d out of range (0..512000)", N);
  TDN_CHECK(num_kept != nullptr, "tdn_nms: NULL num_kept");
  if (N == 0) {
    TDN_MEMSET_ASYNC(num_kept, 0, sizeof(int32_t), st);
    return 0;
  }
  TDN_CHECK(boxes && scores && keep && kept_idx && workspace, "tdn_nms: NULL pointer");
  TDN_CHECK(workspace_bytes >= tdn_nms_workspace(N), "tdn_nms: workspace too small");
  TDN_CHECK(((uintptr_t)workspace & 255) == 0, "tdn_nms: workspace must be 256-byte aligned");
  const int nblk = (N + 63) / 64;
  char* ws = (char*)workspace;
  int* rank = (int*)ws; ws += align256((int64_t)N * 4);
  int* order = (int*)ws; ws += align256((int64_t)N * 4);
  float* sboxes = (float*)ws; ws += align256((int64_t)N * 16);
  unsigned long long* mask = (unsigned long long*)ws;
  TDN_MEMSET_ASYNC(rank, 0, (size_t)N * 4, st);
  const int nb = (N + 255) / 256;
  int jsplit = 1024 / nb;  // aim for ~1024 blocks
  if (jsplit < 1) jsplit = 1;
  if (jsplit > (N + 1023) / 1024) jsplit = (N + 1023) / 1024;
  const int jchunk = ((N + jsplit - 1) / jsplit + 1023) / 1024 * 1024;
  jsplit = (N + jchunk - 1) / jchunk;
  TDN_LAUNCH(nms_rank_kernel, dim3(nb, jsplit), dim3(256), 0, st, scores, N, jchunk, rank);
  TDN_LAUNCH_CHECK();
  TDN_LAUNCH(nms_scatter_kernel, dim3(nb), dim3(256), 0, st, boxes, rank, N, order, sboxes);
  TDN_LAUNCH_CHECK();
  TDN_LAUNCH(nms_mask_kernel, dim3(nblk, nblk), dim3(64), 0, st, sboxes, N, iou_thr, nblk, mask);
  TDN_LAUNCH_CHECK();
  const bool one_wave = getenv("TDN_NMS_ONEWAVE") && getenv("TDN_NMS_ONEWAVE")[0] == '1';   // A/B runs
  if (nblk <= NMS_BLOCK_MAX_NBLK && !one_wave) {
    const size_t lds = (size_t)(((nblk + 1) & ~1) + NMS_SC * NMS_DPITCH + NMS_SC) * 8 + NMS_SCROWS * 4;
    static tdn_attr_once attr_once;
    if (attr_once.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)nms_scan_block_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      TDN_CHECK(e == hipSuccess, "hipFuncSetAttribute(nms scan LDS) failed: %s", hipGetErrorString(e));
      attr_once.mark();
    }
    TDN_LAUNCH(nms_scan_block_kernel, dim3(1), dim3(1024), lds, st, mask, order, N, nblk, keep, kept_idx,
                       num_kept);
  } else {
    TDN_LAUNCH(nms_scan_kernel, dim3(1), dim3(64), (size_t)nblk * 8, st, mask, order, N, nblk, keep, kept_idx,
                       num_kept);
  }
  TDN_LAUNCH_CHECK();
  return 0;
}

// ---- box delta (de)normalisation: datasets/utils/bbox.py:118-166 of the reference --------------------------
// normalize: bbox <- (bbox - means) / stds, IN PLACE like the reference (bbox.sub_(means).div_(stds), bbox.py:140);
// denormalize: out = bbox * stds + means with means/stds tiled over the 4C columns (bbox.py:161-165).
// Separate IEEE sub/div and mul/add (no FMA) so results are bit-identical to the PyTorch-CPU reference.
__global__ void bbox_normalize_kernel(float* bbox, int64_t n4, f32x4_t means, f32x4_t stds) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4_t v = *(f32x4_t*)(bbox + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __fdiv_rn(__fsub_rn(v[e], means[e]), stds[e]);
    *(f32x4_t*)(bbox + i * 4) = v;
  }
}

__global__ void bbox_denormalize_kernel(const float* __restrict__ bbox, float* __restrict__ out, int64_t n4,
                                        f32x4_t means, f32x4_t stds) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t v = *(const f32x4_t*)(bbox + i * 4);
    f32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(v[e], stds[e]), means[e]);
    *(f32x4_t*)(out + i * 4) = o;
  }
}

extern "C" int tdn_bbox_normalize(float* bbox, int64_t rows, const float* means4, const float* stds4, void* stream) {
  TDN_CHECK(rows >= 0 && means4 && stds4, "tdn_bbox_normalize: bad arguments");
  if (rows == 0) return 0;
  TDN_CHECK(bbox != nullptr, "tdn_bbox_normalize: NULL bbox");
  const f32x4_t m = {means4[0], means4[1], means4[2], means4[3]}, s = {stds4[0], stds4[1], stds4[2], stds4[3]};
  int64_t grid = (rows + 255) / 256;
  if (grid > 4096) grid = 4096;
  TDN_LAUNCH(bbox_normalize_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, bbox, rows, m, s);
  TDN_LAUNCH_CHECK();
  return 0;
}

extern "C" int tdn_bbox_denormalize(const float* bbox, float* out, int64_t rows, int cols, const float* means4,
                                    const float* stds4, void* stream) {
  TDN_CHECK(rows >= 0 && cols > 0 && cols % 4 == 0 && means4 && stds4, "tdn_bbox_denormalize: cols must be 4C");
  const int64_t n4 = rows * (cols / 4);
  if (n4 == 0) return 0;
  TDN_CHECK(bbox && out, "tdn_bbox_denormalize: NULL pointer");
  const f32x4_t m = {means4[0], means4[1], means4[2], means4[3]}, s = {stds4[0], stds4[1], stds4[2], stds4[3]};
  int64_t grid = (n4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  TDN_LAUNCH(bbox_denormalize_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, bbox, out, n4, m, s);
  TDN_LAUNCH_CHECK();
  return 0;
}
