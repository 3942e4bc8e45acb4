// Stand-alone micro-benchmark: what one CU's LDS delivers to ds_read_b128 (1 KiB per wave instruction, conflict-free 16-byte
// slots), alone and beside an MFMA stream at the fragment : MFMA ratios of the weight-stationary kernels.
//   build:  hipcc --offload-arch=gfx950 -O3 -o build/lds_rate tools/lds_rate.hip      run:  build/lds_rate
// Variants (8 waves per CU = 2 per SIMD, one workgroup per CU, every CU busy; 64 KiB of LDS read round and round):
//   0  reads only, D reads in flight per wave (D = 4, 8, 16)                  -> bytes per clock per CU the LDS sustains
//   1  R reads per 9 MFMAs (R = 5: the 3x3 kernel at 3 rows per step; R = 3: one read per three MFMAs; R = 0: MFMAs alone)
//      reads issued 6 ahead of their use, every MFMA waits for "its" fragment like the kernels do
//      -> cycles per MFMA as a function of the LDS traffic beside it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// Lane -> byte offset inside the wave's region, for the access shapes the kernels use (lane = (pixel fr = lane & 15, chunk fq = lane >> 4)):
//   0  contiguous: lane * 16
//   1  the weight-stationary 3x3 kernels' ring rows: pixel pitch 2 * 256 + 32 = 544 B, chunk fq * 16
//   2  the same with a 16-byte pad (528 B)
//   3  the LDS-ring kernels' 64-byte K planes: row fr * 64 + ((fq ^ swz(fr)) << 4), swz(r) = (-(r >> 2)) & 3
//   4  the pointwise kernels' 1 KiB pixel rows with chunk X at X ^ (pixel & 15): fr * 1024 + ((fq ^ fr) << 4)   (16 KiB per read: 4 reads per wave region)
//   5  the res2 tail's 128-byte pixels, chunk c at c ^ (pixel & 7): fr * 128 + ((fq ^ (fr & 7)) << 4)
template <int PAT>
__device__ __forceinline__ int lane_off(int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  if constexpr (PAT == 0) return lane * 16;
  else if constexpr (PAT == 1) return fr * 544 + fq * 16;
  else if constexpr (PAT == 2) return fr * 528 + fq * 16;
  else if constexpr (PAT == 3) return fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  else if constexpr (PAT == 4) return fr * 1024 + ((fq ^ fr) << 4);
  else return fr * 128 + ((fq ^ (fr & 7)) << 4);
}
template <int PAT> constexpr int pat_step() { return PAT == 4 ? 64 : (PAT == 1 || PAT == 2 ? 64 : 1024); }   // distance between a wave's successive reads

template <int D, int PAT = 0>
__global__ __launch_bounds__(512, 2) void lds_reads(float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  // a wave walks its own region (8 KiB; shapes whose 16 rows span more share 16 KiB between two waves - reads only, no hazard)
  const unsigned char* base = smem + (PAT == 4 ? (wave >> 1) * 16384 : (PAT == 1 || PAT == 2 ? wave * 8192 % 49152 : wave * 8192)) + lane_off<PAT>(lane);
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = *reinterpret_cast<const u32x4*>(base + ((k + it) & (PAT == 4 ? 3 : 7)) * pat_step<PAT>());   // (the iteration in the address: not loop-invariant)
#pragma unroll
    for (int k = 0; k < D; ++k) acc += v[k];
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = (float)(acc[0] + acc[1] + acc[2] + acc[3]);
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// R fragment reads per group of 9 MFMAs (three accumulators, three weights per fragment like the 3x3 kernel's row taps)
template <int R, int AHEAD, int PAT = 0>
__global__ __launch_bounds__(512, 2) void lds_mfma(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = src[i & 4095];
  u32x4 a[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = src[(i * 64 + lane) & 4095];
  __syncthreads();
  const unsigned char* base = smem + (PAT == 4 ? (wave >> 1) * 16384 : (PAT == 1 || PAT == 2 ? wave * 8192 % 49152 : wave * 8192)) + lane_off<PAT>(lane);
  f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  constexpr int NR = R == 0 ? 1 : R;
  u32x4 bf[AHEAD + 1];
#pragma unroll
  for (int k = 0; k < AHEAD + 1; ++k) bf[k] = *reinterpret_cast<const u32x4*>(base + (k & (PAT == 4 ? 3 : 7)) * pat_step<PAT>());
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // one iteration = AHEAD + 1 groups of 9 MFMAs; MFMA m of group g uses fragment g * R + floor(m * R / 9); (AHEAD + 1) R fragments per
    // iteration keep the ring of AHEAD + 1 registers in step from one iteration to the next
#pragma unroll
    for (int g = 0; g < AHEAD + 1; ++g) {
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int fi = g * NR + (m * NR) / 9;
        const bool newfrag = R > 0 && (m == 0 || (m * NR) / 9 != ((m - 1) * NR) / 9);
        if (newfrag)   // the read AHEAD fragments further on goes into the register the previous fragment has just left
          bf[(fi + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const u32x4*>(base + ((fi + AHEAD + it) & (PAT == 4 ? 3 : 7)) * pat_step<PAT>());
        __builtin_amdgcn_sched_barrier(0);
        MMA(a[m], bf[fi % (AHEAD + 1)], acc[m % 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0] + acc[1] + acc[2];
  out[blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

// the 32x32x16 instruction (twice the FLOPs of 16x16x32 per instruction): R fragment reads per group of 9 MFMAs, as above
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int R, int AHEAD>
__global__ __launch_bounds__(512, 2) void lds_mfma32(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = src[i & 4095];
  u32x4 a[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = src[(i * 64 + lane) & 4095];
  __syncthreads();
  const unsigned char* base = smem + wave * 8192 + lane * 16;
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  constexpr int NR = R == 0 ? 1 : R;
  u32x4 bf[AHEAD + 1];
#pragma unroll
  for (int k = 0; k < AHEAD + 1; ++k) bf[k] = *reinterpret_cast<const u32x4*>(base + (k & 7) * 1024);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < AHEAD + 1; ++g) {
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int fi = g * NR + (m * NR) / 9;
        const bool newfrag = R > 0 && (m == 0 || (m * NR) / 9 != ((m - 1) * NR) / 9);
        if (newfrag) bf[(fi + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const u32x4*>(base + ((fi + AHEAD + it) & 7) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, bf[fi % (AHEAD + 1)]), acc[m % 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += acc[i][k];
  out[blockIdx.x * blockDim.x + tid] = sum;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

static double median_cycles(unsigned long long* cyc, int n) {
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), cyc, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[n / 2];
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  u32x4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 4096 * sizeof(u32x4)); hipMalloc(&out, (size_t)cus * 512 * sizeof(float)); hipMalloc(&cyc, (size_t)cus * 8 * sizeof(unsigned long long));
  std::vector<unsigned short> h(4096 * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3f00 + (rand() & 0xff));     // bf16 values around 0.5 .. 1
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 2000;
#define READS(D, P) { hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_reads<D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((lds_reads<D, P>), dim3(cus), dim3(512), 65536, 0, out, cyc, iters); hipDeviceSynchronize(); \
    const double c = median_cycles(cyc, cus * 8); \
    printf("reads only, access shape %d, %2d in flight per wave: %.1f cycles per 1 KiB read and wave, %.1f B/clk per CU\n", P, D, c / (iters * D), 8.0 * 1024 * iters * D / c); }
  READS(4, 0) READS(8, 0) READS(16, 0)
  READS(16, 1) READS(16, 2) READS(16, 3) READS(16, 4) READS(16, 5) READS(8, 1) READS(8, 3) READS(8, 4) READS(8, 5)
#define MIX(R, A, T, P) { hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma<R, A, P>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((lds_mfma<R, A, P>), dim3(cus), dim3(T), 65536, 0, src, out, cyc, iters); hipDeviceSynchronize(); \
    const double c = median_cycles(cyc, cus * (T / 64)); const int wps = T / 256; \
    printf("%d wave(s) per SIMD, access shape %d, %d fragment reads per 9 MFMAs, %2d ahead: %.2f cycles per MFMA and SIMD, LDS %.1f B/clk per CU\n", wps, P, R, A, \
           c / (iters * 9.0 * (A + 1) * wps), (T / 64) * 1024.0 * iters * (A + 1) * R / c); }
  MIX(0, 6, 512, 0) MIX(1, 6, 512, 0) MIX(3, 6, 512, 0) MIX(5, 6, 512, 0) MIX(9, 6, 512, 0)
  MIX(0, 6, 256, 0) MIX(3, 6, 256, 0) MIX(5, 6, 256, 0) MIX(9, 6, 256, 0) MIX(5, 12, 256, 0) MIX(9, 12, 256, 0) MIX(5, 3, 256, 0) MIX(5, 3, 512, 0)
  MIX(5, 6, 512, 1) MIX(5, 12, 512, 1) MIX(9, 6, 512, 1) MIX(5, 6, 256, 1) MIX(5, 12, 256, 1) MIX(5, 6, 512, 3) MIX(3, 6, 512, 3) MIX(3, 6, 512, 4) MIX(5, 6, 512, 5)
#define MIX32(R, A, T) { hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_mfma32<R, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((lds_mfma32<R, A>), dim3(cus), dim3(T), 65536, 0, src, out, cyc, iters); hipDeviceSynchronize(); \
    const double c = median_cycles(cyc, cus * (T / 64)); const int wps = T / 256; \
    printf("32x32x16: %d wave(s) per SIMD, %d fragment reads per 9 MFMAs, %2d ahead: %.2f cycles per MFMA and SIMD (= %.2f per 16x16x32's worth of FLOPs)\n", wps, R, A, \
           c / (iters * 9.0 * (A + 1) * wps), c / (iters * 9.0 * (A + 1) * wps) / 2); }
  MIX32(0, 6, 512) MIX32(3, 6, 512) MIX32(9, 6, 512) MIX32(9, 6, 256) MIX32(9, 3, 256)
  return 0;
}
