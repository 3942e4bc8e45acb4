// MFMA operand conventions shared by the convolution kernels (dp_conv.hip, dp_bottleneck.hip).
#pragma once
#include "dp_common.h"
#include <type_traits>

namespace {

// native clang vectors (HIP's uint4/int4 are structs; arrays of them ended up in scratch memory)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int r) { return (-(r >> 2)) & 3; }

template <typename T>
struct Mma;
template <>
struct Mma<uint16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<f16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    // whole-vector bit_cast, then index: bit_cast of a single ext-vector element (a.y ...) silently read element 0
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
  }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define DP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define DP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

}  // namespace
