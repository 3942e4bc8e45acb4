// Micro-benchmark (MI355X): what one CU's vector-memory path delivers into LDS (LDS-DMA) or registers as a function of the
// ACCESS SHAPE of a 1 KiB wave instruction - 16 rows x 64 B (the conv ring kernels' K planes), 8 x 128 B, 4 x 256 B, 1 x 1 KiB -
// the row pitch, the level that serves the bytes (all workgroups reading the same rows = L2, private rows re-read = Infinity
// Cache, private rows read once = HBM) and the number of pieces a wave keeps in flight.
//   hipcc --offload-arch=gfx950 -O3 -o build/dma_rate tools/dma_rate.hip && build/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N <= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

struct Args {
  const char* src;
  unsigned src_bytes;
  int pitch;        // bytes between rows
  int rows;         // rows per workgroup tile (multiple of 8 * 1024 / RB)
  int kbytes;       // bytes of a row that are read (multiple of RB)
  int shared_rows;  // the first shared_rows rows of the tile are the same for every workgroup, the rest private
  int reps;
  unsigned* sink;
};

// RB = contiguous bytes per row and piece; F = pieces in flight per wave; REG = 1: loads to registers instead of LDS
template <int RB, int F, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const Args p) {
  constexpr int SLOTS = 128 / WAVES;   // 1 KiB ring slots per wave (128 KiB of LDS in all)
  static_assert(F < SLOTS, "pieces in flight must fit the wave's ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int RPP = 1024 / RB;     // rows per piece
  constexpr int LPR = RB / 16;       // lanes per row
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src), 0, p.src_bytes, 0x00020000);
  const int lrow = lane / LPR, lch = lane % LPR;
  const int groups = p.rows / RPP;           // row groups per plane
  const int planes = p.kbytes / RB;
  const int priv_rows = p.rows - p.shared_rows;
  unsigned char* const ring = smem + wave * (SLOTS * 1024);
  int q = 0;
  for (int rep = 0; rep < p.reps; ++rep) {
    for (int s = 0; s < planes; ++s) {
      for (int g = wave; g < groups; g += WAVES) {
        const int r = g * RPP + lrow;
        const long long row = r < p.shared_rows ? r : (long long)p.shared_rows + (long long)blockIdx.x * priv_rows + (r - p.shared_rows);
        const int off = (int)(row * p.pitch + s * RB + lch * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + (q & (SLOTS - 1)) * 1024), 16, off, 0, 0, 0);
        wait_vm<F>();
        ++q;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x] == 0x7f && p.reps < 0) p.sink[0] = 1;
}

// register form with F independent loads in flight: a burst of F loads, then their use
template <int RB, int F, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void reg_kernel(const Args p) {
  constexpr int RPP = 1024 / RB, LPR = RB / 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.src), 0, p.src_bytes, 0x00020000);
  const int lrow = lane / LPR, lch = lane % LPR;
  const int groups = p.rows / RPP, planes = p.kbytes / RB;
  const int priv_rows = p.rows - p.shared_rows;
  const int total = planes * (groups / WAVES);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int rep = 0; rep < p.reps; ++rep) {
    for (int q0 = 0; q0 < total; q0 += F) {
      u32x4 v[F];
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const int q = q0 + k < total ? q0 + k : total - 1;
        const int s = q / (groups / WAVES), g = wave + WAVES * (q % (groups / WAVES));
        const int r = g * RPP + lrow;
        const long long row = r < p.shared_rows ? r : (long long)p.shared_rows + (long long)blockIdx.x * priv_rows + (r - p.shared_rows);
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row * p.pitch + s * RB + lch * 16), 0, 0);
      }
#pragma unroll
      for (int k = 0; k < F; ++k) acc ^= v[k];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) p.sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K>
static double run(K kern, int waves, int lds, const Args& a, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, a);
  CK(hipEventRecord(e0));
  const int L = 5;
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / L * 1e-3;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const size_t bytes = 3ull << 30;   // 3 GiB: > Infinity Cache for the read-once cases, < 4 GiB for 32-bit buffer offsets... see below
  char* src = nullptr;
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 1, bytes));
  unsigned* sink = nullptr;
  CK(hipMalloc(&sink, 64));
  printf("%d CUs. one workgroup per CU; GB/s per CU (chip TB/s)\n", cus);
  printf("%-34s %-28s %10s %10s %10s %10s\n", "source", "form", "16x64B", "8x128B", "4x256B", "1x1KiB");
  struct Case { const char* name; int pitch, rows, kbytes, shared_rows, reps; };
  // rows = 384 per workgroup (like a 128-pixel x 256-cout ring tile); pitch 2048 = a 1024-channel bf16 row
  const Case cases[] = {
      {"L2: all WGs same 384 rows x 2KB", 2048, 384, 2048, 384, 16},
      {"IC: private 384 rows x 1KB, re-read", 1024, 384, 1024, 0, 16},
      {"mix: 256 shared + 128 priv x 2KB", 2048, 384, 2048, 256, 8},
      {"HBM: private 384 rows x 16KB once", 16384, 384, 16384, 0, 1},
      {"IC: private 128 rows, pitch 4KB, 2KB read", 4096, 128, 2048, 0, 16},
  };
  for (const Case& c : cases) {
    Args a;
    a.src = src; a.pitch = c.pitch; a.rows = c.rows; a.kbytes = c.kbytes; a.shared_rows = c.shared_rows; a.reps = c.reps; a.sink = sink;
    const long long need = ((long long)c.shared_rows + (long long)cus * (c.rows - c.shared_rows)) * c.pitch;
    if (need > (long long)bytes || need >= (1ll << 31)) { printf("%-34s skipped (needs %lld bytes)\n", c.name, need); continue; }
    a.src_bytes = (unsigned)need;
    const double total = (double)cus * c.rows * c.kbytes * c.reps;
    auto line = [&](const char* form, double t0, double t1, double t2, double t3) {
      printf("%-34s %-28s", c.name, form);
      for (double t : {t0, t1, t2, t3}) printf(" %5.1f(%4.1f)", total / t / cus * 1e-9, total / t * 1e-12);
      printf("\n");
      fflush(stdout);
    };
    constexpr int L = 128 * 1024;
    line("LDS-DMA 8 waves, 2 in flight", run(dma_kernel<64, 2, 8>, 8, L, a, cus), run(dma_kernel<128, 2, 8>, 8, L, a, cus),
         run(dma_kernel<256, 2, 8>, 8, L, a, cus), run(dma_kernel<1024, 2, 8>, 8, L, a, cus));
    line("LDS-DMA 8 waves, 4 in flight", run(dma_kernel<64, 4, 8>, 8, L, a, cus), run(dma_kernel<128, 4, 8>, 8, L, a, cus),
         run(dma_kernel<256, 4, 8>, 8, L, a, cus), run(dma_kernel<1024, 4, 8>, 8, L, a, cus));
    line("LDS-DMA 8 waves, 8 in flight", run(dma_kernel<64, 8, 8>, 8, L, a, cus), run(dma_kernel<128, 8, 8>, 8, L, a, cus),
         run(dma_kernel<256, 8, 8>, 8, L, a, cus), run(dma_kernel<1024, 8, 8>, 8, L, a, cus));
    line("LDS-DMA 8 waves, 12 in flight", run(dma_kernel<64, 12, 8>, 8, L, a, cus), run(dma_kernel<128, 12, 8>, 8, L, a, cus),
         run(dma_kernel<256, 12, 8>, 8, L, a, cus), run(dma_kernel<1024, 12, 8>, 8, L, a, cus));
    line("LDS-DMA 4 waves, 8 in flight", run(dma_kernel<64, 8, 4>, 4, L, a, cus), run(dma_kernel<128, 8, 4>, 4, L, a, cus),
         run(dma_kernel<256, 8, 4>, 4, L, a, cus), run(dma_kernel<1024, 8, 4>, 4, L, a, cus));
    line("LDS-DMA 4 waves, 24 in flight", run(dma_kernel<64, 24, 4>, 4, L, a, cus), run(dma_kernel<128, 24, 4>, 4, L, a, cus),
         run(dma_kernel<256, 24, 4>, 4, L, a, cus), run(dma_kernel<1024, 24, 4>, 4, L, a, cus));
    line("registers 8 waves, 8 in flight", run(reg_kernel<64, 8, 8>, 8, 0, a, cus), run(reg_kernel<128, 8, 8>, 8, 0, a, cus),
         run(reg_kernel<256, 8, 8>, 8, 0, a, cus), run(reg_kernel<1024, 8, 8>, 8, 0, a, cus));
    line("registers 4 waves, 16 in flight", run(reg_kernel<64, 16, 4>, 4, 0, a, cus), run(reg_kernel<128, 16, 4>, 4, 0, a, cus),
         run(reg_kernel<256, 16, 4>, 4, 0, a, cus), run(reg_kernel<1024, 16, 4>, 4, 0, a, cus));
  }
  CK(hipFree(src));
  CK(hipFree(sink));
  return 0;
}
