// Shared device/host helpers for libtdn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include "../../include/tdn.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define TDN_LDS __attribute__((address_space(3)))
#define TDN_GLOBAL __attribute__((address_space(1)))

// ---- error plumbing -------------------------------------------------------------------
void tdn_set_error(const char* fmt, ...);
#define TDN_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      tdn_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)
#define TDN_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      tdn_set_error("HIP launch failed at %s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -2;                                                        \
    }                                                                   \
  } while (0)

// ---- launch recorder (plan.hip) ---------------------------------------------------------------
// Every kernel launch of the library goes through TDN_LAUNCH.  Normally that is just the launch.  While a plan is
// being recorded (tdn_plan_begin ... tdn_plan_end) the launch is ALSO kept, with all its arguments by value, as a
// closure that tdn_plan_run can issue again on the same stream: a prepared step is then one C call that enqueues the
// whole launch list, instead of ~230 trips through the host-side operator layer.
#include <functional>
bool tdn_plan_recording();
void tdn_plan_push(hipStream_t stream, std::function<void(hipStream_t)> fn);
#define TDN_LAUNCH(kernel, grid, block, lds, stream, ...)                                                     \
  do {                                                                                                        \
    hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__);                         \
    if (tdn_plan_recording())                                                                                 \
      tdn_plan_push((hipStream_t)(stream), [=](hipStream_t s__) {                                             \
        hipLaunchKernelGGL(kernel, grid, block, lds, s__, __VA_ARGS__);                                       \
      });                                                                                                     \
  } while (0)
#define TDN_MEMSET_ASYNC(ptr, value, bytes, stream)                                                           \
  do {                                                                                                        \
    (void)hipMemsetAsync(ptr, value, bytes, (hipStream_t)(stream));                                           \
    if (tdn_plan_recording())                                                                                 \
      tdn_plan_push((hipStream_t)(stream), [=](hipStream_t s__) { (void)hipMemsetAsync(ptr, value, bytes, s__); }); \
  } while (0)

#define TDN_CHECK_DTYPE(dtype) \
  TDN_CHECK((dtype) == TDN_BF16 || (dtype) == TDN_F16, "dtype %d is neither TDN_BF16 nor TDN_F16", (int)(dtype))
// launch kernel<F16> chosen by the run-time dtype code
#define TDN_LAUNCH_T(kernel, dtype, grid, block, stream, ...)                                  \
  do {                                                                                         \
    if ((dtype) == TDN_F16) TDN_LAUNCH(kernel<true>, grid, block, 0, stream, __VA_ARGS__);  \
    else TDN_LAUNCH(kernel<false>, grid, block, 0, stream, __VA_ARGS__);                \
  } while (0)

// ---- small device helpers -------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }

// 16-bit element type by template flag.  Tensors are carried as bf16_t* (a 2-byte element either way: addressing,
// LDS-DMA, swizzles and transposing reads do not care); only the conversions and the MFMA opcode differ for
// TDN_F16.  Both conversions round to nearest even, like PyTorch's .bfloat16() / .half().
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

template <bool F16>
__device__ __forceinline__ float elem_to_f32(bf16_t raw) {
  if constexpr (F16) return (float)__builtin_bit_cast(f16_t, raw);
  else return (float)raw;
}
template <bool F16>
__device__ __forceinline__ bf16_t f32_to_elem(float v) {
  if constexpr (F16) return __builtin_bit_cast(bf16_t, (f16_t)v);
  else return (bf16_t)v;
}
template <bool F16>
__device__ __forceinline__ f32x4_t load4_f32(const bf16_t* p) {
  const bf16x4_t r = *(const bf16x4_t*)p;
  f32x4_t v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = elem_to_f32<F16>(r[e]);
  return v;
}
template <bool F16>
__device__ __forceinline__ void store4_f32(bf16_t* p, f32x4_t v) {
  bf16x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f32_to_elem<F16>(v[e]);
  *(bf16x4_t*)p = o;
}
// D = A(16 x 32) * B(32 x 16) + C on the matrix cores; operands are 8 consecutive k per lane
template <bool F16>
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                  0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// a * b as an fp32 value of its own.  Without the barrier the backend may fold  (f16)(a * b)  into v_fma_mixlo_f16 —
// ONE rounding of the exact product instead of the two of torch's  (x.float() * s).half()  — and does so in some
// kernels and not in others (seen in the weight-pack kernels: 1-4 elements per tensor differed by an ulp).
__device__ __forceinline__ float mul_f32_rounded(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

// Upper knee of nn.ReLU6 for a value about to be stored in the 16-bit element type: 6 from 6 up, and below 6 never
// more than the largest element value under 6 (bf16 5.96875, fp16 5.99609375) — rounding to nearest would otherwise
// turn (5.984, 6) into 6.0, and the backward pass, which reads the mask 0 < y < 6 from the stored output, would drop
// those elements' gradients.
template <bool F16>
__device__ __forceinline__ float relu6_top(float v) {
  return v >= 6.f ? 6.f : fminf(v, F16 ? 5.99609375f : 5.96875f);
}

// 16-byte async global -> LDS copy. LDS destination = wave-uniform `lds_base` + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const TDN_GLOBAL void*)gsrc, (TDN_LDS void*)lds_base, 16, 0, 0);
}

// The same copy issued through inline asm, for loops that prefetch stage t+1 while they still read stage t.  The
// compiler treats the builtin above as an LDS store that any later ds_read may alias and puts `s_waitcnt vmcnt(0)` in
// front of the first such read, i.e. it waits for the prefetch it has just issued.  Hidden in asm, the copy is ordered
// by the kernel's own `s_waitcnt vmcnt` + `s_barrier` only.  `lds_base` must be wave-uniform.
__device__ __forceinline__ void glds16_async(const void* gsrc, void* lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds_addr = (unsigned)(size_t)(TDN_LDS char*)lds_base;
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"((const TDN_GLOBAL void*)gsrc), "s"(lds_addr) : "memory");   // m0 is reserved: the compiler never keeps a value in it across statements
#endif
}

__device__ __forceinline__ bf16x8_t lds_read_b128(const void* p) {
  return *(const TDN_LDS bf16x8_t*)((const TDN_LDS char*)p);
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p addresses row q / cols 4p..4p+3 of a 4x16 block of
// 16-bit elements; lane i receives column i (rows 0..3 in elements 0..3).
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((TDN_LDS s16x4_t*)(p));
}

// 256 zero bytes: LDS-DMA source for padding / out-of-range rows (one copy per translation unit)
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) applies to ONE device: remembered per kernel instantiation AND device
// (`static tdn_attr_once once; if (once.need()) { ...set...; once.mark(); }`), so that a second GPU used from the same
// process does not launch without it.
struct tdn_attr_once {
  bool done[64] = {};
  int dev = 0;
  bool need() {
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return true; }
    return !done[dev];
  }
  void mark() { if (dev >= 0) done[dev] = true; }
};

// XCD-aware bijective remap: blocks b, b+8, b+16.. share an XCD (observed round-robin placement,
// speed only) -> give each XCD a contiguous chunk of tile ids so neighbours share L2 lines.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
