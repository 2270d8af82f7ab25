/*
 * tdn.h — C ABI of libtdn.so, the MI355X (gfx950) native hot path behind the
 * Torch_Detection module registry.
 *
 * The reference (TCGGroup/Torch_Detection) has NO native interface: every hot
 * operation is a torch.nn call made from Python.  Each entry point below names
 * the reference call site (file:line, relative to the reference root) whose
 * arithmetic it replaces.  The only intended binder is ctypes from
 * torch_detection_amd/_lib.py (see INTEGRATION.md for the stub a reference
 * maintainer would add).
 *
 * Conventions
 *   - All tensors are caller-owned device memory (PyTorch allocations).  The
 *     library never allocates, frees or synchronises; every kernel is enqueued
 *     on the hipStream_t passed as `stream` (void*; NULL = default stream).
 *   - Activations / gradients are NHWC ("channels last"), 2-byte elements: the
 *     `dtype` argument says which — TDN_BF16 (bfloat16) or TDN_F16 (IEEE half;
 *     the reference's model.half()).  Every 16-bit operand of one call has that
 *     type; accumulation, BN/bias constants and all parameter gradients are fp32.
 *     Weights are pre-packed K-major in the same type by tdn_pack_conv_weight.
 *   - Return value: 0 = ok, negative = error; tdn_last_error() returns a
 *     thread-local message.  Nothing throws across the boundary.
 *   - Re-entrant: no global mutable state besides the thread-local error text
 *     (and the optional diagnostics buffer of tdn_debug_trace).
 */
#ifndef TDN_H_
#define TDN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDN_VERSION 100 /* 0.1.0 */

enum { TDN_BF16 = 0, TDN_F16 = 1 };

/* epilogue addend modes */
enum {
  TDN_ADD_NONE = 0,
  TDN_ADD_SAME = 1,     /* addend has the output's shape (residual `out += residual`, resnet.py:57,117) */
  TDN_ADD_UP2X = 2,     /* addend is the 2x coarser map, nearest-upsampled (fpn.py:99-101)                */
  TDN_ADD_SUMPOOL2 = 3  /* addend is the 2x finer map, 2x2 sum-pooled (adjoint of fpn.py:99-101)         */
};

/* Fused epilogue applied to the fp32 accumulator of a conv / dgrad GEMM:
 *   v = acc * scale[c] + shift[c]          (eval-mode BatchNorm2d folded, layers.py:50-54; or conv bias, layers.py:93)
 *   v += addend(...)                        (see modes above)
 *   if relu:      v = max(v, 0)             (nn.ReLU, resnet.py:35,90,217); relu == 2: v = min(v, 6) as well
 *                                           (nn.ReLU6, layers.py:117-118), values under 6 stay under 6 when stored
 *   if mask_src:  v = mask_src > 0 ? v : 0  (adjoint of that ReLU, from the saved forward output)
 *   out = (bf16) v          (or fp32 when out_f32 != 0: pre-rounding value, used for 1e-3 parity checks
 *                             and for fp32 module outputs)
 */
typedef struct tdn_epilogue {
  const float* scale;    /* [Cout] or NULL (= 1) */
  const float* shift;    /* [Cout] or NULL (= 0) */
  const void* addend;    /* NHWC, Cout channels, or NULL */
  int32_t addend_mode;   /* TDN_ADD_* */
  int32_t addend_h;      /* spatial size of the addend tensor (UP2X / SUMPOOL2) */
  int32_t addend_w;
  int32_t relu;          /* 0 none / 1 ReLU / 2 ReLU6 */
  const void* mask_src;  /* NHWC, same shape as the output, or NULL */
  int32_t out_f32;       /* 0: out is bf16 NHWC; 1: out is float32 NHWC */
  int32_t reserved;
  /* Optional scratch for cross-workgroup split-K (small-M, long-K layers: too few output tiles for 256 CUs, so a tile's
   * K range is cut over several workgroups and the last one to arrive sums the fp32 partials in split order —
   * deterministic).  Layout: the first TDN_SPLITK_TICKET_BYTES bytes are per-tile arrival counters and must be ZERO
   * before the first use (the kernels leave them zero); the rest holds partial tiles.  Must not be shared by launches
   * that may run concurrently (one buffer per stream).  NULL / too small: the launch simply does not split. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
} tdn_epilogue;
#define TDN_SPLITK_TICKET_BYTES 65536
/* One-time device probe (allocates and frees a few KB, synchronises the device — call it outside any stream capture;
 * torch_detection_amd._lib.load() does): do workgroups with equal blockIdx.x always run on the same XCD?  1 yes, 0 no.
 * The XCD-local form of the split-K exchange (TDN_SPLITK=1) is only used when it returned 1. */
int tdn_probe_xcd_mapping(void);

const char* tdn_last_error(void);
int tdn_version(void);

/* ---- weight / norm preparation ------------------------------------------------ */

/* Eval-mode BatchNorm2d folded to a per-channel affine (layers.py:50-54, resnet.py:270-276):
 *   invstd = 1/sqrt(var+eps); scale = gamma*invstd; shift = beta - mean*scale. */
int tdn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, int C, float* scale, float* shift, float* invstd, void* stream);

/* Pack an nn.Conv2d weight (layers.py:12,25,40,93) for the GEMM kernels.
 *   w: fp32, logical [Cout][Cin][kh][kw] with element strides (s_o, s_i, s_h, s_w).
 *   w_fwd:   bf16 [Cout][kh][kw][Cin]                       (forward / wgrad-finalize operand)
 *   w_dgrad: bf16 [Cin][kh][kw][Cout] = scale[co] * w      (may be NULL; dgrad operand, BN scale folded)
 */
int tdn_pack_conv_weight(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w,
                         int Cout, int Cin, int kh, int kw, const float* scale,
                         void* w_fwd, void* w_dgrad, int dtype, void* stream);

/* Stem variant (conv7x7_group(3,64,stride=2), resnet.py:214): w fp32 [64][3][7][7] contiguous ->
 * bf16 [Cout][7][8][4] (kw padded 7->8, channels padded 3->4, pads zero). */
int tdn_pack_stem_weight(const float* w, int Cout, void* w_fwd, int dtype, void* stream);

/* ---- convolution (nn.Conv2d.forward via resnet.py:42-59,97-119,253-258; fpn.py:92-108) ---- */

/* y[N][Ho][Wo][Cout] = epilogue(conv(x[N][H][W][Cin], w_fwd)), square kernel k in {1,3},
 * stride in {1,2}; "same" padding: k = 1: pad 0; k = 3: pad = dilation in 1..32, exactly as conv3x3_group builds its
 * convs (padding = dilation, models/utils/layers.py:20-32; ResNet(dilations=...), resnet.py:187-233) — the dilation of
 * a 3x3 conv is read from `pad` in every tdn_*conv2d_* entry point.
 * Requires Cin % 64 == 0 and Cout % 64 == 0. */
int tdn_conv2d_fwd(const void* x, const void* w_fwd, void* y, int N, int H, int W, int Cin,
                   int Cout, int k, int stride, int pad, const tdn_epilogue* ep, int dtype,
                   void* stream);

/* dx[N][H][W][Cin] = epilogue(conv_transpose(g[N][Ho][Wo][Cout], w_dgrad)) — input gradient of
 * the conv above (autograd of nn.Conv2d; no explicit reference line: the reference never calls
 * backward). */
int tdn_conv2d_dgrad(const void* g, const void* w_dgrad, void* dx, int N, int H, int W, int Cin,
                     int Cout, int k, int stride, int pad, const tdn_epilogue* ep, int dtype,
                     void* stream);

/* Workspace bytes tdn_conv2d_wgrad needs for this shape. */
int64_t tdn_conv2d_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int k, int stride,
                                    int pad);

/* Weight / affine gradients of  y = (conv(x, w)) * scale + shift  given g = dL/d(y pre-activation):
 *   dw[Cout][kh][kw][Cin] (fp32; channels_last view of the nn.Conv2d grad)  = beta*dw + scale * (g^T im2col(x))
 *   BN mode (mean, invstd != NULL): dgamma = beta*dgamma + invstd*(sum_k w*G - mean*sum_m g), dbeta = beta*dbeta + sum_m g
 *   bias mode (mean == NULL):       dbeta (= dbias) = beta*dbeta + sum_m g ; dgamma ignored (may be NULL)
 * scale may be NULL (= 1). */
int tdn_conv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale,
                     const float* mean, const float* invstd, float* dw, float* dgamma,
                     float* dbeta, float beta, int N, int H, int W, int Cin, int Cout, int k,
                     int stride, int pad, void* workspace, int64_t workspace_bytes, int dtype,
                     void* stream);

/* Grouped launch: the weight / affine gradients of SEVERAL convs (e.g. every conv of one ResNet stage, or of the
 * FPN) in as few launches as the tile shapes allow — autograd of the nn.Conv2d / BatchNorm2d parameters created at
 * models/backbone/resnet.py:74-91,214-216 and models/necks/fpn.py:44-58, which the reference leaves to one autograd
 * node per layer.  Members are independent; each is described like one tdn_conv2d_wgrad / tdn_stem_conv_wgrad /
 * tdn_gconv2d_wgrad call.  The library cuts every member's pixel range (the GEMM's reduction index) into splits
 * sized so that the GROUP fills the chip — a member that can be reduced by one workgroup per tile writes its
 * gradient directly (no fp32 partial slabs) — and reduces the rest, plus every member's BN / bias gradients, in ONE
 * finalize launch, in a fixed order (bit-reproducible run to run).  items / n: HOST array. */
enum { TDN_WGRAD_CONV = 0, TDN_WGRAD_STEM = 1, TDN_WGRAD_GCONV = 2 };
typedef struct tdn_wgrad_item {
  const void* x;       /* conv input, NHWC 16-bit [N][H][W][Cin]; TDN_WGRAD_STEM: the staged image xp (tdn_stage_image) */
  const void* g;       /* dL/d(pre-activation output), NHWC 16-bit [N][Ho][Wo][Cout] */
  const void* w_fwd;   /* packed forward weights (tdn_pack_conv_weight / _stem_ / _gconv_); read for dgamma */
  const float* scale;  /* BN scale or NULL (= 1) */
  const float* mean;   /* BN mean, or NULL: bias mode */
  const float* invstd; /* BN 1/sqrt(var + eps), or NULL */
  float* dw;           /* as in the single-layer call of the member's kind */
  float* dgamma;       /* may be NULL in bias mode */
  float* dbeta;        /* may be NULL */
  float beta;          /* 0: overwrite, else accumulate beta * old + new */
  int32_t kind;        /* TDN_WGRAD_* */
  int32_t N, H, W;     /* input size (STEM: the image size, even) */
  int32_t Cin, Cout;   /* GCONV: Cin = Cout = C */
  int32_t k, stride, pad;   /* ignored for STEM */
  int32_t groups;      /* GCONV only */
  int32_t reserved;
} tdn_wgrad_item;
int64_t tdn_wgrad_group_workspace(const tdn_wgrad_item* items, int n, int dtype);
int tdn_wgrad_group(const tdn_wgrad_item* items, int n, void* workspace, int64_t workspace_bytes, int dtype,
                    void* stream);
/* Host-only: the decomposition tdn_wgrad_group would use.  per_item[n][8] = {kernel (0 tap-per-tile, 1 nine-tap),
 * tile_co, tile_ci, splits, pixels per split, direct (1: no slabs), workgroups, fp32 slab bytes / 1024};
 * totals[4] = {gradient-kernel launches, finalize launches, workgroups, slab KiB}. */
int tdn_wgrad_group_plan(const tdn_wgrad_item* items, int n, int dtype, int32_t* per_item, int32_t* totals);

/* ---- operand preparation of many conv units in one launch -----------------------------------------------------
 * What tdn_bn_fold + tdn_pack_conv_weight do per layer (the eval-mode BN fold of models/utils/layers.py:50-54 as the
 * reference's default bn_eval=True makes it, resnet.py:270-276, and the 16-bit copies of nn.Conv2d.weight), for a
 * whole list of plain convolutions (groups = 1, channels multiples of 64) at once: a training step re-derives every
 * operand after the optimizer update, and 113 launches become ceil(n / 30).  Results are bit-identical to the
 * per-layer calls.  gamma == NULL: no norm (scale 1, `fold` untouched).  items / n: HOST array. */
typedef struct tdn_prep_item {
  const float* w;            /* fp32 logical [Cout][Cin][kh][kw], element strides s_o, s_i, s_h, s_w */
  void* w_fwd;               /* 16-bit [Cout][kh][kw][Cin] */
  void* w_dgrad;             /* 16-bit [Cin][kh][kw][Cout] = elem(elem(w) * scale[co]); may be NULL */
  const float* gamma;        /* BatchNorm2d weight / bias / running_mean / running_var, or all NULL */
  const float* beta;
  const float* mean;
  const float* var;
  float* fold;               /* fp32 [3][Cout]: scale, shift, invstd (written when gamma != NULL) */
  int64_t s_o, s_i, s_h, s_w;
  int32_t Cout, Cin, kh, kw;
  float eps;
  int32_t reserved;
} tdn_prep_item;
int tdn_prepare_group(const tdn_prep_item* items, int n, int dtype, void* stream);

/* ---- one residual Bottleneck per launch (models/backbone/resnet.py:97-119) ------------------
 * Stride-1 Bottleneck without a downsample branch (resnet.py:110-118), C mid channels, 4C in / out:
 *   forward   out1 = relu(bn1(conv1(in)));  out2 = relu(bn2(conv2(out1)));  out3 = relu(bn3(conv3(out2)) + in)
 *   dgrad     out1 = mask1 . conv3^T(in);   out2 = mask2 . conv2^T(out1);   out3 = mask3 . (conv1^T(out2) + in)
 *             (in = dL/d(block output before its ReLU), already masked; mask_k . v keeps v where mask_k > 0;
 *              mask1 = saved conv2 output, mask2 = saved conv1 output, mask3 = the block input, NULL = no mask)
 * in ONE launch: the intermediate tensors are handed from GEMM to GEMM through LDS (a workgroup owns 8 x 16 output
 * pixels and recomputes conv1 on the one-pixel halo) and are only WRITTEN to out1 / out2 (the weight gradients read
 * them later).  Bit-identical to the three tdn_conv2d_fwd / tdn_conv2d_dgrad launches it replaces.
 *   forward: w1 / w2 / w3 = w_fwd of conv1 / conv2 / conv3, scale_k / shift_k their folded BN (NULL = 1 / 0)
 *   dgrad:   w1 / w2 / w3 = w_dgrad of conv3 / conv2 / conv1 (BN scale folded), scale / shift unused
 * tdn_bottleneck_supported: 1 if this build has a kernel for (C, stride, dilation); callers fall back to per-conv
 * launches otherwise. */
typedef struct tdn_bottleneck_args {
  const void* in;       /* NHWC 16-bit [N][H][W][4C] */
  const void* w1;       /* [C][1][1][4C] */
  const void* w2;       /* [C][3][3][C] */
  const void* w3;       /* [4C][1][1][C] */
  const float* scale1; const float* shift1;   /* [C] */
  const float* scale2; const float* shift2;   /* [C] */
  const float* scale3; const float* shift3;   /* [4C] */
  const void* mask1;    /* [N][H][W][C] */
  const void* mask2;    /* [N][H][W][C] */
  const void* mask3;    /* [N][H][W][4C] */
  void* out1;           /* [N][H][W][C] */
  void* out2;           /* [N][H][W][C] */
  void* out3;           /* [N][H][W][4C] */
  int32_t N, H, W, C;
  /* ReLU bit planes, optional (all NULL = off): 1 bit per element, bit c % 32 of the 32-bit word c / 32 of the pixel.
   *   forward: written —  bits1 = out1 > 0, bits2 = out2 > 0, bits3 = in > 0 (any subset)
   *   dgrad:   read INSTEAD of mask1 / mask2 / mask3 (all three or none): bits2 masks out1, bits1 masks out2, bits3
   *            masks out3 — i.e. the three planes the forward call of the same block wrote.  1/16 of the mask bytes. */
  void* bits1;          /* [N][H][W][C / 32]  uint32 */
  void* bits2;          /* [N][H][W][C / 32]  uint32 */
  void* bits3;          /* [N][H][W][4C / 32] uint32 */
} tdn_bottleneck_args;
int tdn_bottleneck_supported(int H, int W, int C, int stride, int dilation);
int tdn_bottleneck_fwd(const tdn_bottleneck_args* a, int dtype, void* stream);
int tdn_bottleneck_dgrad(const tdn_bottleneck_args* a, int dtype, void* stream);

/* The stage's FIRST Bottleneck where it keeps the resolution (layer1.0: models/backbone/resnet.py:130-136 builds a
 * 1x1 conv + BN `downsample` because inplanes != planes * 4; :113-114 adds it instead of x): the block input has Cin
 * channels and the residual branch is downsample(x).  Built for Cin == C == 64, stride 1.
 *   forward:  b.in [N][H][W][Cin]; b.w1 [C][1][1][Cin]; b.out3 [N][H][W][4C] = relu(bn3(conv3(out2)) + addend) with
 *             addend [N][H][W][4C] = the downsample branch (a tdn_conv2d_fwd launch of the caller's), or computed in
 *             the launch from wd / scale_d / shift_d (see below);
 *             bits1 / bits2 optional, bits3 must be NULL
 *   dgrad:    b.in = g [N][H][W][4C]; b.w3 = conv1 w_dgrad [Cin][1][1][C]; b.out3 = dx [N][H][W][Cin] =
 *             conv1^T(out2) + addend with addend [N][H][W][Cin] = the downsample conv's input gradient; the block input
 *             comes from the max pool: no mask3 / bits3; bits1 and bits2 (both or none) replace mask2 / mask1
 * Bit-identical to the three tdn_conv2d_fwd / tdn_conv2d_dgrad launches it replaces. */
typedef struct tdn_bottleneck_head_args {
  tdn_bottleneck_args b;
  const void* addend;
  /* INSTEAD of addend: the downsample conv's own operands — forward: w_fwd [4C][1][1][Cin] and folded BN [4C]
   * (NULL = 1 / 0); dgrad: w_dgrad [Cin][1][1][4C] (scale_d / shift_d unused).  The branch is then computed inside the
   * launch (same K order, affine and 16-bit rounding as its own launch would apply: still bit-identical) and never
   * travels through HBM */
  const void* wd;
  const float* scale_d; const float* shift_d;
} tdn_bottleneck_head_args;
int tdn_bottleneck_head_supported(int H, int W, int Cin, int C, int stride, int dilation);
int tdn_bottleneck_head_fwd(const tdn_bottleneck_head_args* a, int dtype, void* stream);
int tdn_bottleneck_head_dgrad(const tdn_bottleneck_head_args* a, int dtype, void* stream);

/* ---- grouped convolution (ResNeXt) -----------------------------------
 * conv3x3_group(..., groups=cardinality) of models/backbone/resnext.py:26-28,82-83: C channels in and out,
 * `groups` groups.  Computed in block-diagonal form: every 64-channel block of the output multiplies only the same
 * 64 input channels, with zeros outside the true groups — requires C % 64 == 0 and (C / groups) | 64.
 *   tdn_pack_gconv_weight: w fp32 logical [C][C/groups][kh][kw] (element strides) ->
 *       w_fwd [C][kh][kw][64], w_dgrad [C][kh][kw][64] (BN scale folded; may be NULL)
 *   tdn_gconv2d_fwd / _dgrad: as tdn_conv2d_fwd / _dgrad (same epilogue) on those operands
 *   tdn_gconv2d_wgrad: dw fp32 [C][kh][kw][C/groups] (= channels_last bytes of the grouped parameter); dgamma / dbeta /
 *       scale / mean / invstd / beta as in tdn_conv2d_wgrad */
int tdn_pack_gconv_weight(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int C, int groups,
                          int kh, int kw, const float* scale, void* w_fwd, void* w_dgrad, int dtype, void* stream);
int tdn_gconv2d_fwd(const void* x, const void* w_fwd, void* y, int N, int H, int W, int C, int groups, int k,
                    int stride, int pad, const tdn_epilogue* ep, int dtype, void* stream);
int tdn_gconv2d_dgrad(const void* g, const void* w_dgrad, void* dx, int N, int H, int W, int C, int groups, int k,
                      int stride, int pad, const tdn_epilogue* ep, int dtype, void* stream);
int64_t tdn_gconv2d_wgrad_workspace(int N, int H, int W, int C, int groups, int k, int stride, int pad);
int tdn_gconv2d_wgrad(const void* x, const void* g, const void* w_fwd, const float* scale, const float* mean,
                      const float* invstd, float* dw, float* dgamma, float* dbeta, float beta, int N, int H, int W,
                      int C, int groups, int k, int stride, int pad, void* workspace, int64_t workspace_bytes,
                      int dtype, void* stream);

/* ---- stem (resnet.py:214-218,254-258) ------------------------------------------ */

/* NCHW image (fp32, arbitrary element strides) -> zero-padded NHWC4 bf16 staging buffer
 * xp[N][H+6][W+8][4] (3 px halo top/left/bottom, 3+2 right; channel 3 = 0).  This is the
 * device-side form of the pad/transpose of dataset_transforms.py:37-44. */
int tdn_stage_image(const float* img, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N,
                    int H, int W, void* xp, int dtype, void* stream);

/* y[N][H/2][W/2][64] = relu(bn(conv7x7 s2 p3 (img)))  from the staged image. H, W even. */
int tdn_stem_conv_fwd(const void* xp, const void* w_stem, void* y, int N, int H, int W, int Cout,
                      const tdn_epilogue* ep, int dtype, void* stream);

/* The stem in one launch: conv7x7/s2 (tdn_stem_conv_fwd) + eval-mode BN (scale, shift: tdn_bn_fold) + ReLU +
 * MaxPool2d(3, 2, 1) (tdn_maxpool3x3s2_fwd) — resnet.py:214-218, 254-258.  y[N][Ho][Wo][64] and idx (window
 * position of the first maximum) are bit-identical to the two separate calls; the stem's full-size activation is
 * never written (backward: tdn_maxpool3x3s2_relu_bwd reads the ReLU mask from y).  Cout must be 64; H, W even. */
int tdn_stem_pool_fwd(const void* xp, const void* w_stem, const float* scale, const float* shift,
                      void* y, uint8_t* idx, int N, int H, int W, int Cout, int dtype, void* stream);

int64_t tdn_stem_conv_wgrad_workspace(int N, int H, int W, int Cout);

/* dw fp32 [Cout][3][7][7] contiguous (+ BN grads as in tdn_conv2d_wgrad). */
int tdn_stem_conv_wgrad(const void* xp, const void* g, const void* w_stem, const float* scale,
                        const float* mean, const float* invstd, float* dw, float* dgamma,
                        float* dbeta, float beta, int N, int H, int W, int Cout, void* workspace,
                        int64_t workspace_bytes, int dtype, void* stream);

/* ---- pooling / resampling -------------------------------------------------------- */

/* nn.MaxPool2d(3, stride=2, padding=1) (resnet.py:218,258) on NHWC; idx[N][Ho][Wo][C] (uint8) records
 * the window position (kh*3+kw) of the first maximum, PyTorch's tie rule. C % 8 == 0. */
int tdn_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C,
                         int dtype, void* stream);

/* dx[N][H][W][C] = relu_mask(x) * scatter(dy by idx): adjoint of maxpool (and, when mask_src != NULL,
 * of the in-place ReLU before it, resnet.py:257). */
int tdn_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, const void* mask_src, void* dx, int N,
                         int H, int W, int C, int dtype, void* stream);

/* The same adjoint of ReLU -> maxpool (resnet.py:257-258) with the ReLU mask taken from the pool's OUTPUT
 * y_pooled[N][Ho][Wo][C] instead of its full-size input: a window's value is the value of the element it
 * selected, so the gradient of a window passes where y_pooled > 0.  Same dx bit for bit; reads a tensor a
 * quarter of the size, and the stem's activation need not be kept for backward. */
int tdn_maxpool3x3s2_relu_bwd(const void* dy, const uint8_t* idx, const void* y_pooled, void* dx, int N,
                              int H, int W, int C, int dtype, void* stream);

/* F.max_pool2d(x, 1, stride=2) (fpn.py:116): y[N][ceil(H/2)][ceil(W/2)][C] = x[:, ::2, ::2, :]. */
int tdn_subsample2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype,
                       void* stream);

/* dx = dx_in (may be NULL = 0) + scatter(dy) : adjoint of the above, fused with the accumulation. */
int tdn_subsample2_bwd(const void* dy, const void* dx_in, void* dx, int N, int H, int W, int C,
                       int dtype, void* stream);

/* out = (a (+ b)) masked by mask_src > 0  (b, mask_src may be NULL). Element-wise, n elements, n % 8 == 0. */
int tdn_add_relu_mask(const void* a, const void* b, const void* mask_src, void* out, int64_t n,
                      int dtype, void* stream);

/* bf16 NHWC <-> fp32 NCHW (logical, element strides) boundary converters. */
/* ConvModule activation / pre-activation pieces (models/utils/layers.py:57-135: activation='relu6',
 * activate_last=False).  tdn_clamp_max: y = min(y, hi) in place (the upper clamp of nn.ReLU6 after a ReLU epilogue).
 * tdn_act_mask: out = g where 0 < y < hi, else 0 — the activation's backward from its saved OUTPUT y (hi = +inf:
 * ReLU, 6: ReLU6).  n % 8 == 0. */
int tdn_clamp_max(void* y, float hi, int64_t n, int dtype, void* stream);
int tdn_act_mask(const void* g, const void* y, void* out, float hi, int64_t n, int dtype, void* stream);
/* Pre-activation order (layers.py:129-134: norm -> activate -> conv): BatchNorm2d in eval mode on the conv's INPUT,
 * folded to y = act(x * scale[c] + shift[c]) (tdn_bn_fold), act 0 none / 1 ReLU / 2 ReLU6; NHWC [npix][C], C % 8 == 0.
 * Backward, from g = dL/dy already masked by the activation (tdn_act_mask): dx = g * scale[c],
 * dbeta[c] = sum g, dgamma[c] = invstd[c] * sum g * (x - mean[c]); beta != 0 accumulates into dgamma / dbeta.
 * Deterministic (per-chunk partial sums in the workspace, added in order). */
int tdn_channel_affine_fwd(const void* x, const float* scale, const float* shift, void* y, int64_t npix, int C,
                           int act, int dtype, void* stream);
int64_t tdn_channel_affine_bwd_workspace(int64_t npix, int C);
int tdn_channel_affine_bwd(const void* g, const void* x, const float* scale, const float* mean, const float* invstd,
                           void* dx, float* dgamma, float* dbeta, float beta, int64_t npix, int C, void* workspace,
                           int64_t workspace_bytes, int dtype, void* stream);

int tdn_nchw_f32_to_nhwc(const float* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w,
                         int N, int C, int H, int W, void* dst, int dtype, void* stream);
/* 16-bit source (bf16 or fp16 bits, element strides of the logical NCHW tensor) -> contiguous NHWC of the same type */
int tdn_nchw16_to_nhwc(const void* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int N, int C, int H,
                       int W, void* dst, void* stream);
int tdn_nhwc_to_nchw_f32(const void* src, int N, int C, int H, int W, float* dst, int dtype,
                         void* stream);

/* ---- prepared launch lists -----------------------------------------------------------------------------------------
 * One C call that enqueues a whole recorded step (the launches behind ResNet.forward, models/backbone/resnet.py:253-268,
 * FPN.forward, models/necks/fpn.py:88-125, and their backward) for callers that cannot capture a hipGraph.
 *   tdn_plan_begin()            start recording: every launch the library makes from now on is also kept (stream,
 *                               kernel, arguments by value); one recording at a time, process-wide
 *   tdn_plan_event_record(s)    the host recorded an event on stream s here -> plan-local event id (-1: not recording)
 *   tdn_plan_stream_wait(s, id) the host made stream s wait for that event here
 *   tdn_plan_end()              stop recording -> plan handle (NULL on error)
 *   tdn_plan_run(plan)          enqueue everything again, same streams, same order, same dependencies; no sync
 *   tdn_plan_stats(plan, out)   out[3] = {launches, event records, stream waits}; RETURNS the number of launches that were
 *                               made by OTHER threads while the plan was being recorded and were therefore NOT kept
 *                               (0 = the plan holds every launch of the recorded step; < 0: bad handle)
 *   tdn_plan_free(plan)
 * Pointers held by the recorded arguments must still refer to the same buffers when the plan runs (the host keeps the
 * recorded step's tensors alive in a private pool, like a captured graph does). */
int tdn_plan_begin(void);
void* tdn_plan_end(void);
int tdn_plan_event_record(void* stream);
int tdn_plan_stream_wait(void* stream, int event_id);
int tdn_plan_run(void* plan);
int tdn_plan_stats(void* plan, int32_t* out3);
int tdn_plan_free(void* plan);

/* ---- box ops (absent from the reference: core/__init__.py is empty; semantics = SURVEY Appendix B,
 *      conventions pinned by datasets/utils/bbox.py:39,375-377 and dataset_transforms.py:120-131) ---- */

/* anchors[(y*featW + x)*A + a][4] = base[a] + (x*stride, y*stride, x*stride, y*stride);
 * valid[...] = x < valid_w && y < valid_h (uint8, may be NULL). */
int tdn_anchor_grid(const float* base_anchors, int A, int featH, int featW, int stride,
                    int valid_h, int valid_w, float* anchors, uint8_t* valid, void* stream);

/* The whole pyramid in one launch: `nlevels` (<= 8) levels, outputs concatenated level after level — anchors fp32
 * [sum_l featH_l*featW_l*A_l][4], valid (may be NULL) one byte per anchor; inside a level exactly what tdn_anchor_grid
 * writes.  levels: HOST array; base_anchors: device pointers. */
typedef struct tdn_anchor_level {
  const float* base_anchors;   /* device, fp32 [A][4] */
  int32_t A, featH, featW, stride, valid_h, valid_w;
} tdn_anchor_level;
int tdn_anchor_pyramid(const tdn_anchor_level* levels, int nlevels, float* anchors, uint8_t* valid, void* stream);

/* iou[N][M] (fp32) of inclusive-pixel xyxy boxes, '+1' convention, IEEE fp32 (no contraction). */
int tdn_bbox_iou_pairwise(const float* a, int N, const float* b, int M, float* iou, void* stream);

int64_t tdn_nms_workspace(int N);

/* Greedy NMS: stable sort by score desc (ties: lower index first), suppress iou > thr.
 *   keep[N] uint8 (original order), kept_idx[N] int64 (score order, first *num_kept valid),
 *   num_kept: device int32. */
int tdn_nms(const float* boxes, const float* scores, int N, float iou_thr, uint8_t* keep,
            int64_t* kept_idx, int32_t* num_kept, void* workspace, int64_t workspace_bytes,
            void* stream);

/* Box delta (de)normalisation (reference: datasets/utils/bbox.py:118-166).
 *   normalize:   bbox[rows][4] <- (bbox - means) / stds, IN PLACE like `bbox.sub_(means).div_(stds)` (bbox.py:140)
 *   denormalize: out[rows][cols] = bbox * stds + means, means/stds tiled over cols = 4C (bbox.py:161-165)
 * means4 / stds4 are HOST arrays of 4 floats. Separate IEEE sub/div and mul/add: bit-identical to PyTorch-CPU. */
int tdn_bbox_normalize(float* bbox, int64_t rows, const float* means4, const float* stds4, void* stream);
int tdn_bbox_denormalize(const float* bbox, float* out, int64_t rows, int cols, const float* means4,
                         const float* stds4, void* stream);

/* ---- GroupNorm ----------------------------------------------------
 * nn.GroupNorm(get_group_gn(planes), planes) — models/utils/layers.py:50-54,138-154 (32 groups, eps 1e-5, biased
 * variance) — after a conv of ResNet(use_gn=True) (models/backbone/resnet.py:42-59,97-119,254-257) or of a
 * ConvModule with GN (layers.py:122-135), fused with the residual add and ReLU that follow it.
 *   (relu: 0 none, 1 ReLU, 2 ReLU6 — as in the conv epilogue; the same for tdn_bn_train_fwd)
 *   tdn_gn_fwd: z (N,H,W,C) raw conv output -> y = relu?((z - mu) * rstd * gamma + beta (+ addend)); addend_mode
 *               TDN_ADD_SAME (same shape: the residual) or TDN_ADD_UP2X ((N,H/2,W/2,C): FPN top-down, fpn.py:98-100);
 *               stats (N,C,2) float = per-channel (mu, rstd) of the channel's group, kept for the backward.
 *   tdn_gn_bwd: g = dL/dy (already ReLU-masked) -> dz = dL/dz (16-bit, feeds tdn_conv2d_dgrad / _wgrad with no BN
 *               fold), dgamma, dbeta (float; acc = 1 accumulates into them, 0 overwrites).
 * C a power of two in 64..2048, G | C.  workspace: tdn_gn_workspace() bytes, 16-byte aligned. */
int64_t tdn_gn_workspace(int N, int H, int W, int C, int G);
int tdn_gn_fwd(const void* z, const float* gamma, const float* beta, int N, int H, int W, int C, int G, float eps,
               const void* addend, int addend_mode, int relu, void* y, float* stats, void* workspace,
               int64_t workspace_bytes,
               int dtype, void* stream);
int tdn_gn_bwd(const void* g, const void* z, const float* stats, const float* gamma, int N, int H, int W, int C,
               int G, void* dz, float* dgamma, float* dbeta, float acc, void* workspace, int64_t workspace_bytes,
               int dtype, void* stream);

/* ---- BatchNorm2d with batch statistics (training mode) ------------------------------------
 * nn.BatchNorm2d as norm_layer builds it (models/utils/layers.py:50-54) when the backbone is NOT told to keep BN in eval
 * mode — ResNet(bn_eval=False), models/backbone/resnet.py:270-276.  Same passes as GroupNorm with the statistics taken
 * per channel over (N, H, W); running_mean / running_var (may both be NULL) are updated in place:
 * r = (1 - momentum) r + momentum * batch value, unbiased variance.  stats (N,C,2) as in tdn_gn_fwd.
 * Workspace: tdn_gn_workspace(N, H, W, C, C). */
int tdn_bn_train_fwd(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, int N, int H, int W, int C, float eps, const void* addend, int addend_mode,
                     int relu, void* y, float* stats, void* workspace, int64_t workspace_bytes, int dtype,
                     void* stream);
int tdn_bn_train_bwd(const void* g, const void* z, const float* stats, const float* gamma, int N, int H, int W, int C,
                     void* dz, float* dgamma, float* dbeta, float acc, void* workspace, int64_t workspace_bytes,
                     int dtype, void* stream);

/* ---- image batch staging -------------------------------------------
 * One launch for what the reference does per image on the host and then in collate():
 *   img_normalize            datasets/utils/image.py:87-105     (img - mean) / std, float32
 *   img_flip (horizontal)    datasets/utils/image.py:220-249
 *   img_pad_size_divisor     datasets/utils/image.py:300-347    zero pad bottom/right
 *   HWC -> CHW               datasets/dataset_transforms.py:44
 *   collate (stack, pad 0)   datasets/loader/collate.py:42-63   pad every sample to the batch maximum
 * imgs: HOST array of N device pointers to H_i x W_i x 3 pixels (src_kind 0 = uint8, 1 = float32), already resized;
 * hw: HOST int32 [N][2] = (H_i, W_i); flip: HOST [N] flags or NULL; mean3 / std3: HOST float[3], in the images'
 * channel order.  Hb x Wb = batch size (>= every image, normally rounded up to the size divisor by the caller).
 * out_kind 0: float32 (N, 3, Hb, Wb), bit-identical to the reference chain (two IEEE operations per element).
 * out_kind 1: the stem's staged input (N, Hb+6, Wb+8, 4) in `dtype` — exactly tdn_stage_image of the out_kind-0
 *             batch, without the float32 round trip.  N <= TDN_COLLATE_MAX per call. */
#define TDN_COLLATE_MAX 16
int tdn_collate_images(const void* const* imgs, const int32_t* hw, const uint8_t* flip, int N, int src_kind,
                       const float* mean3, const float* std3, int Hb, int Wb, void* out, int out_kind, int dtype,
                       void* stream);

/* ---- host-only introspection (no GPU needed; used by CPU tests) --------------------- */

/* Describes the GEMM decomposition the library would launch for a conv: fills out[0..15] with
 * {M, Ngemm, Kgemm, BM, BN, BK, grid_x, grid_y, grid_z, nclasses, ntaps(class0), splitk, ...}.
 * kind: 0 = fwd, 1 = dgrad, 2 = wgrad.  A shape taken by the LDS-resident patch kernel (csrc/conv_halo.hip: 3x3
 * stride-1 convs) reports grid_z = 100 + its configuration id, out[11] = patch rows * 1000 + patch columns and
 * out[12] = chunk images held in LDS * 100 + output-channel passes per workgroup. */
int tdn_conv2d_plan(int kind, int N, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                    int32_t* out16);

/* Diagnostics only: registers a device buffer of `bytes` bytes into which the tracing builds of the conv GEMM
 * kernel (selected with the TDN_GEMM_CFG environment variable, scripts/trace_gemm.py) write 32 64-bit shader-clock
 * stamps per workgroup.  buf = NULL disables tracing.  No reference counterpart. */
int tdn_debug_trace(void* buf, long long bytes);

#ifdef __cplusplus
}
#endif
#endif /* TDN_H_ */
