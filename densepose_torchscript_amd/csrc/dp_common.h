// Shared device/host helpers for the gfx950 DensePose kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/densepose_hip.h"
#include "dp_policy.h"

// ---- error reporting (host) -------------------------------------------------------------------
extern thread_local char dp_err_buf[512];
inline int dp_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dp_err_buf, sizeof(dp_err_buf), fmt, ap);
  va_end(ap);
  return code;
}
#define DP_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return dp_fail(DP_ERR_BAD_ARG, __VA_ARGS__);   \
  } while (0)
inline int dp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dp_fail(DP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DP_OK;
}

// ---- 16-bit storage <-> f32 (device) ---------------------------------------------------------------
// Storage types: float, uint16_t (= bf16 bit pattern) and f16_t (IEEE half, reference `.half()` mode
// /root/reference/export.py:36-37). Arithmetic outside the MFMA operands is always fp32.
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __builtin_bit_cast(float, lo16 << 16); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  // a vector conversion: hipcc emits ONE v_cvt_pk_bf16_f32 for the pair (RNE, NaN stays NaN); two scalar casts cost
  // two of them plus a shift and an or
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t lo16) { return (float)__builtin_bit_cast(f16_t, (uint16_t)lo16); }
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  // RNE like torch's .half(); values beyond 65504 become inf exactly as they do in the reference's fp16 mode
  f16_t ha = (f16_t)a, hb = (f16_t)b;
  return (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int kDtype = DP_F32;
  static constexpr int kChunk = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ uint32_t pack2(float, float) { return 0u; }  // never used: fp32 is stored unpacked
  __device__ static __forceinline__ float unpack(uint32_t) { return 0.f; }
};
template <>
struct Elem<uint16_t> {  // bf16 storage
  static constexpr int kDtype = DP_BF16;
  static constexpr int kChunk = 8;
  __device__ static __forceinline__ float load(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void store(uint16_t* p, float v) {
    bf16_t h = (bf16_t)v;
    *p = __builtin_bit_cast(uint16_t, h);
  }
  __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
  __device__ static __forceinline__ float unpack(uint32_t lo16) { return bf16_bits_to_f32(lo16); }
};
template <>
struct Elem<f16_t> {  // IEEE half storage
  static constexpr int kDtype = DP_F16;
  static constexpr int kChunk = 8;
  __device__ static __forceinline__ float load(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void store(f16_t* p, float v) { *p = (f16_t)v; }
  __device__ static __forceinline__ uint32_t pack2(float a, float b) { return pack_f16x2(a, b); }
  __device__ static __forceinline__ float unpack(uint32_t lo16) { return f16_bits_to_f32(lo16); }
};

// load / store 4 consecutive channels (8-byte aligned for 16-bit storage, 16-byte for f32)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <typename T>
__device__ __forceinline__ float4 load4(const T* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(Elem<T>::unpack(u.x & 0xffffu), Elem<T>::unpack(u.x >> 16), Elem<T>::unpack(u.y & 0xffffu),
                     Elem<T>::unpack(u.y >> 16));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <typename T>
__device__ __forceinline__ void store4(T* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(Elem<T>::pack2(v.x, v.y), Elem<T>::pack2(v.z, v.w));
}

// 8 consecutive channels: ONE 16-byte access for 16-bit storage, two for f32
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&o)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  o[0] = Elem<T>::unpack(u.x & 0xffffu); o[1] = Elem<T>::unpack(u.x >> 16);
  o[2] = Elem<T>::unpack(u.y & 0xffffu); o[3] = Elem<T>::unpack(u.y >> 16);
  o[4] = Elem<T>::unpack(u.z & 0xffffu); o[5] = Elem<T>::unpack(u.z >> 16);
  o[6] = Elem<T>::unpack(u.w & 0xffffu); o[7] = Elem<T>::unpack(u.w >> 16);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]),
                                            Elem<T>::pack2(v[6], v[7]));
}

// dp_conv_ws.hip: the weight-stationary 3x3 kernel (128 -> 128 and 256 -> 256 channels) behind dp_conv2d_nhwc (kernel class 6)
bool dp_conv_wsr_ok(const dp_conv_params* p);
int dp_conv_wsr_launch(const dp_conv_params* p, dp_stream_t stream);
// dp_conv_wq.hip: the weight-stationary 3x3 kernel on v_mfma_f32_32x32x16, one wave per SIMD (256 -> 256 channels; kernel class 10)
bool dp_conv_wsq_ok(const dp_conv_params* p);
int dp_conv_wsq_launch(const dp_conv_params* p, dp_stream_t stream);
// dp_conv_rows.hip: the row-streaming K-split weight-stationary 3x3 kernel (512 / 256 input channels) behind dp_conv2d_nhwc (kernel class 7)
bool dp_conv_rows_ok(const dp_conv_params* p);
bool dp_conv_rows2_ok(const dp_conv_params* p);     // its 32-pixel form (kernel class 8)
int dp_conv_rows_launch(const dp_conv_params* p, dp_stream_t stream);
// dp_conv_pw.hip: the weight-stationary pointwise kernel (K = 512 / 1024 / 2048 channels) behind dp_conv2d_nhwc (kernel class 9)
bool dp_conv_pws_ok(const dp_conv_params* p);
int dp_conv_pws_launch(const dp_conv_params* p, dp_stream_t stream);

// dp_pair256.hip: the res4 form (256 -> 1024 -> 256) of dp_bottleneck_pair_nhwc
int dp_pair256_launch(const dp_pair_params* p, dp_stream_t stream);

// Packed weight matrices are stored in 1 KiB TILES of 16 rows x 64 bytes of K (round 3): tile (rg, plane) of a matrix with
// n_planes = Kpad * esize / 64 K planes sits at ((rg * n_planes) + plane) * 1024, row-major inside. One LDS-DMA wave instruction (16
// rows of one plane, what every convolution kernel stages) then reads 8 consecutive whole cache lines instead of 16 half lines from
// rows Kpad * esize bytes apart. Byte offset of 16-byte chunk `chunk` (0..3) of row `row` in plane `plane`:
__host__ __device__ inline long long dp_wtile_off(int row, int plane, int chunk, int n_planes) {
  return ((long long)(row >> 4) * n_planes + plane) * 1024 + (row & 15) * 64 + chunk * 16;
}

static inline hipStream_t as_stream(dp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
