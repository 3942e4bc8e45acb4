// Stand-alone micro-benchmark: cycles per v_mfma_f32_16x16x32_bf16 inside instruction streams shaped like the weight-stationary kernels'
// matrix phases (64 MFMAs per phase: 32 distinct A fragments from the register file, B fragments in a small rotating set, 4 accumulators).
//   build:  hipcc --offload-arch=gfx950 -O3 -o build/mfma_rate tools/mfma_rate.hip      run:  build/mfma_rate
// Variants (all on operands in registers, no memory traffic in the timed region):
//   0  same A, same B, 4 accumulators in rotation                                  (the guide's bare loop)
//   1  32 distinct A fragments (128 VGPRs), 2 B fragments, 4 accumulators           (operand pattern of conv1x1_pws_kernel)
//   2  variant 1 with 16 accumulators (distance 16 between dependent MFMAs)
//   3  variant 1 with an s_nop 1 between the MFMA pairs                              (does a gap change the pace?)
//   4  variant 1, A operands in AGPRs is not expressible in HIP: instead the SAME A for the two MFMAs of a pair and a new B each (swapped roles)
// Each with 1 and 2 waves per SIMD (256- / 512-thread workgroups, one workgroup per CU, every CU busy).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512, 2) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[32], b[4];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = src[(i * 64 + lane) & 4095];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(2048 + i * 64 + lane) & 4095];
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 32; ++f) {      // one "fragment" = two MFMAs, as in the kernels
      if constexpr (V == 0) {
        MMA(a[0], b[0], acc[(2 * f) & 3]);
        MMA(a[0], b[0], acc[(2 * f + 1) & 3]);
      } else if constexpr (V == 1 || V == 3) {
        const int c = f >> 1, pt = f & 1;
        MMA(a[c], b[pt], acc[pt * 2]);
        MMA(a[16 + c], b[pt], acc[pt * 2 + 1]);
        if constexpr (V == 3) asm volatile("s_nop 1");
      } else if constexpr (V == 2) {
        const int c = f >> 1, pt = f & 1;
        MMA(a[c], b[pt], acc[(f * 2) & 15]);
        MMA(a[16 + c], b[pt], acc[(f * 2 + 1) & 15]);
      } else {
        const int c = f >> 1, pt = f & 1;
        MMA(b[pt], a[c], acc[pt * 2]);
        MMA(b[pt], a[16 + c], acc[pt * 2 + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// 32x32x16 (round 6): 32 distinct A fragments, 2 B fragments, 2 accumulators of 16 registers - the operand pattern of conv3x3_wsq_kernel
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MMA32(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
__global__ __launch_bounds__(512, 2) void mfma32_loop(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[32], b[2];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = src[(i * 64 + lane) & 4095];
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = src[(2048 + i * 64 + lane) & 4095];
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int f = 0; f < 32; ++f) {
      MMA32(a[f], b[f & 1], acc[f & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

void run32(int threads, const u32x4* src, float* out, unsigned long long* cyc, int cus) {
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma32_loop, dim3(cus), dim3(threads), 0, 0, src, out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma32_loop, dim3(cus), dim3(threads), 0, 0, src, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = cus * threads / 64;
  std::vector<unsigned long long> h(nw);
  hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[nw / 2];
  const double per_wave = med / (32.0 * iters);
  const double tf = 2.0 * 32 * 32 * 16 * 32.0 * iters * nw / (ms * 1e-3) / 1e12;
  printf("%-58s %d waves/SIMD: %6.2f cycles per MFMA in a wave's stream, %6.2f per MFMA on the SIMD, %7.0f TFLOP/s, clock %.2f GHz\n",
         "5 v_mfma_f32_32x32x16_bf16: 32 distinct A, 2 B, 2 accumulators", threads / 256, per_wave, per_wave / (threads / 256), tf, med / (ms * 1e-3) / 1e9);
}

template <int V>
void run(const char* what, int threads, const u32x4* src, float* out, unsigned long long* cyc, int cus) {
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mfma_loop<V>), dim3(cus), dim3(threads), 0, 0, src, out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mfma_loop<V>), dim3(cus), dim3(threads), 0, 0, src, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = cus * threads / 64;
  std::vector<unsigned long long> h(nw);
  hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[nw / 2];
  const double per_wave = med / (64.0 * iters);                 // cycles per MFMA of one wave's stream
  const double per_simd = per_wave / (threads / 256);           // ... per MFMA of the SIMD (two streams share the pipe at 512 threads)
  const double tf = 2.0 * 16 * 16 * 32 * 64.0 * iters * nw / (ms * 1e-3) / 1e12;
  printf("%-58s %d waves/SIMD: %6.2f cycles per MFMA in a wave's stream, %6.2f per MFMA on the SIMD, %7.0f TFLOP/s, clock %.2f GHz\n", what, threads / 256,
         per_wave, per_simd, tf, med / (ms * 1e-3) / 1e9);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<unsigned short> h(4096 * 8);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));   // bf16 values around 0.01 .. 0.03: random mantissas, no overflow
  u32x4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 4096 * 16); hipMalloc(&out, cus * 512 * 4); hipMalloc(&cyc, cus * 8 * 8);
  hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<0>("0 same A, same B, 4 accumulators", threads, src, out, cyc, cus);
    run<1>("1 32 distinct A (weights), 2 B, 4 accumulators", threads, src, out, cyc, cus);
    run<2>("2 32 distinct A, 2 B, 16 accumulators", threads, src, out, cyc, cus);
    run<3>("3 variant 1 + s_nop 1 between the pairs", threads, src, out, cyc, cus);
    run<4>("4 roles swapped: B-side register varies, A fixed per pair", threads, src, out, cyc, cus);
    run32(threads, src, out, cyc, cus);
  }
  return 0;
}
