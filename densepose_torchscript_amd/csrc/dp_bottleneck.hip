// Bottleneck tail in ONE launch for the 64-channel (res2) blocks of the ResNet trunk:
//
//     t2  = relu(conv2_3x3(t1) + b2)                      64 -> 64, 9 taps          (resnet.py:195-197)
//     out = relu(conv3_1x1(t2) + b3 + residual)           64 -> 256                 (resnet.py:199-205)
//     t1' = relu(conv1'_1x1(out) + b1')                   256 -> 64: conv1 of the NEXT block (resnet.py:192-193), optional
//
// First block of the stage (round 5): the projection shortcut rides in conv3's K axis instead of a residual tensor,
//     out = relu([W3 | Ws] [t2 ; x] + b3 + bs)            (64 + 64) -> 256          (resnet.py:199-205 with 189-190)
// (pack.dual_source_pointwise: K planes 0, 1 = conv3, planes 2, 3 = the shortcut over the block input x) - the 256-channel
// shortcut tensor is never written nor read back; the two extra weight planes take conv1''s place in LDS, so this form has no
// next-conv1 stage (strip walker only).
//
// (/root/reference/detectron2/modeling/backbone/resnet.py:189-205, FrozenBN folded by pack.py.) Run layer by layer
// these three are HBM-bound pointwise / short-K launches that move 275 MB tensors five times per block; chained they
// move them twice (residual in, block output out) plus the 64-channel t1 / t1' side tensors.
//
// How the chain works without an LDS round trip: pack.py permutes the weight rows of every 64-cout block so that the 16
// accumulator values an MFMA lane owns along cout are two runs of 8 CONSECUTIVE output channels of one pixel (dp_conv.hip,
// store_tile). Converted to the storage type, run h of a 64-cout block IS the B-operand fragment (pixel = lane & 15,
// 8 channels at K offset 8 * (lane >> 4)) of K step h of the next 1x1 convolution: conv2's accumulators feed conv3 and
// conv3's epilogue registers feed conv1' directly. The values are rounded to the storage type exactly where the
// layer-by-layer path stores them, and every accumulation runs in the same K order, so the fused launch is bit-identical
// to dp_conv2d_nhwc called three times (tests/test_gpu_kernels.py::test_bottleneck_tail_*).
//
// Structure (the streaming 1x1 kernel of dp_conv.hip, extended): all three weight matrices (72 + 32 + 32 KiB) are staged
// into LDS once per workgroup; every wave owns whole 32-pixel tiles (persistent, strided over the pixel range); its
// conv2 operand fragments (9 taps x 2 channel blocks per pixel tile) and its residual runs are plain 16-byte buffer loads
// straight into registers, refilled IN PLACE with the wave's next tile right after their last use; zero padding and the
// ragged tail are out-of-range buffer offsets (loads return 0, stores are dropped). One workgroup of 4 waves per CU.
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"
#include <stdlib.h>

#ifndef DP_EXP
#define DP_EXP 0
#endif
#ifndef DP_TAIL_NW
#define DP_TAIL_NW 8   // waves per workgroup (one workgroup per CU)
#endif
#ifndef DP_TAIL_TP
#define DP_TAIL_TP 1   // 16-pixel MFMA tiles per wave tile
#endif
#ifndef DP_STRIP_NW
#define DP_STRIP_NW 8      // strip walker: waves per workgroup (experiment builds: 4 = one wave per SIMD)
#endif
#ifndef DP_STRIP_ORDER
#define DP_STRIP_ORDER 0   // strip walker, job order: 0 = segments of a strip first, 1 = strips of a segment first
#endif
#ifndef DP_STRIP_DRAIN
#define DP_STRIP_DRAIN 0   // strip walker, experiment: drain the job's first loads before the row loop (see there; measured: no change)
#endif
#ifndef DP_STRIP_AHEAD
#define DP_STRIP_AHEAD 2   // strip walker: K steps the weight-fragment reads run ahead of the MFMAs (register sets = this + 1)
#endif

namespace {

struct TailArgs {
  const void* t1;
  const void* res;
  void* out;
  void* t1n;
  const void* w2;
  const void* w3;
  const void* w1n;
  const i32x4* ktab2;
  const float* b2;
  const float* b3;
  const float* b1n;
  int N, H, W, M;
  int kpad2, kpad3, kpad1n;
  int hi_off, wi_off;
  unsigned t1_bytes, out_bytes, t1n_bytes, res_bytes;
};

constexpr int kTailW2 = 18 * 64 * 64;                  // conv2: 18 K planes x 64 couts x 64 B
constexpr int kTailW3 = 2 * 256 * 64;                  // conv3: 2 planes x 256 couts
constexpr int kTailW1 = 8 * 64 * 64;                   // conv1': 8 planes x 64 couts
constexpr int kTailBias = (64 + 256 + 64) * 4;
constexpr int kTailLds = kTailW2 + kTailW3 + kTailW1 + kTailBias;

// One K step of the chain = 4 weight fragments (64 couts x 32 K) against the pixel fragments of the wave tile. The steps
// of a tile are numbered 0..33: 0..17 conv2 (K plane s = tap s >> 1, channel block s & 1), 18..25 conv3 (64-cout block b, plane s), 26..33 conv1' (plane q);
// the weight fragments of step k+1 are read from LDS into the other register set while the MFMAs of step k run.
// (C3 = K planes of conv3: 2, or 4 with the projection shortcut behind it - steps 18..33 are then conv3 + shortcut, no conv1')
template <int K, int C3 = 2>
__device__ __forceinline__ const unsigned char* tail_wfrag_addr(const unsigned char* w2_s, const unsigned char* w3_s, const unsigned char* w1_s) {
  if constexpr (K < 18) return w2_s + K * 4096;
  else if constexpr (K < 18 + 4 * C3) return w3_s + ((K - 18) % C3) * 16384 + ((K - 18) / C3) * 4096;
  else return w1_s + (K - 26) * 4096;
}

template <typename T, bool HAS_NEXT, int NW, int TP>
__global__ __launch_bounds__(NW * 64, NW / 4) void bottleneck_tail64_kernel(const TailArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int NSTEP = HAS_NEXT ? 34 : 26;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const w2_s = smem;
  unsigned char* const w3_s = smem + kTailW2;
  unsigned char* const w1_s = smem + kTailW2 + kTailW3;
  float* const b2_s = reinterpret_cast<float*>(smem + kTailW2 + kTailW3 + kTailW1);
  float* const b3_s = b2_s + 64;
  float* const b1_s = b3_s + 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- weights -> LDS, once: piece = (K plane, 16-row group), 1 KiB per wave instruction; the 16-byte chunk c of row r
  //      lands at slot c ^ swz(r) (swizzle applied to the per-lane SOURCE address, the LDS-DMA destination is lane-linear)
  {
    const int srow = lane >> 2;
    const int scc = (lane & 3) ^ swz(srow);
    const unsigned char* __restrict__ w2 = reinterpret_cast<const unsigned char*>(p.w2) + dp_wtile_off(srow, 0, scc, p.kpad2 * 2 / 64);   // tiled weight matrix
    for (int piece = wave; piece < 18 * 4; piece += NW) {
      const int pl = piece >> 2, rg = piece & 3;
      __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w2 + ((long long)rg * (p.kpad2 * 2 / 64) + pl) * 1024),
                                       DP_LDS_PTR(w2_s + pl * 4096 + rg * 1024), 16, 0, 0);
    }
    const unsigned char* __restrict__ w3 = reinterpret_cast<const unsigned char*>(p.w3) + dp_wtile_off(srow, 0, scc, p.kpad3 * 2 / 64);   // tiled weight matrix
    for (int piece = wave; piece < 2 * 16; piece += NW) {
      const int pl = piece >> 4, rg = piece & 15;
      __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w3 + ((long long)rg * (p.kpad3 * 2 / 64) + pl) * 1024),
                                       DP_LDS_PTR(w3_s + pl * 16384 + rg * 1024), 16, 0, 0);
    }
    if (HAS_NEXT) {
      const unsigned char* __restrict__ w1 = reinterpret_cast<const unsigned char*>(p.w1n) + dp_wtile_off(srow, 0, scc, p.kpad1n * 2 / 64);
      for (int piece = wave; piece < 8 * 4; piece += NW) {
        const int pl = piece >> 2, rg = piece & 3;
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w1 + ((long long)rg * (p.kpad1n * 2 / 64) + pl) * 1024),
                                         DP_LDS_PTR(w1_s + pl * 4096 + rg * 1024), 16, 0, 0);
      }
    }
    if (tid < 64) b2_s[tid] = p.b2[tid];
    if (tid < 256) b3_s[tid] = p.b3[tid];
    if (HAS_NEXT && tid < 64) b1_s[tid] = p.b1n[tid];
  }
  __syncthreads();  // (vmcnt(0) + barrier)

  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int rd = fr * 64 + ((fq ^ swz(fr)) << 4);   // fragment read offset inside a 16-row group of a plane
  const int n_wt = (p.M + 16 * TP - 1) / (16 * TP);
  const int wt_step = gridDim.x * NW;
  int wt = blockIdx.x * NW + wave;
  if (wt >= n_wt) return;

  // the 3 x 3 taps of conv2 (wave-uniform; pack.py enumerates them row-major: tap t = row t / 3, column t % 3): pixel
  // displacement and byte offset in the 128-byte-per-pixel t1 tensor. K is packed TAP-major (plane s = tap s >> 1, channel
  // block s & 1): the two 64-byte halves of a t1 pixel (one 128-byte line) are fetched by consecutive loads.
  const __attribute__((address_space(4))) i32x4* ktab_c = (const __attribute__((address_space(4))) i32x4*)p.ktab2;
  int tdy[3], tdx[3], toff[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    tdy[i] = ktab_c[i * 3 * 8][0] + p.hi_off;   // ktab holds one entry per 16-byte K chunk: 8 per tap (64 channels)
    tdx[i] = ktab_c[i * 8][1] + p.wi_off;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (tdy[t / 3] * p.W + tdx[t % 3]) * 128;
  // a wave's consecutive tiles are wt_step tiles apart: (row, column) of a pixel advance by a constant (dh, dw) + carry
  const int HW = p.H * p.W;
  const int step_px = wt_step * 16 * TP;
  const int dstep = step_px % HW, dh = dstep / p.W, dw = dstep - dh * p.W;

  const __amdgpu_buffer_rsrc_t rs_t1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t1), 0, p.t1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t1n = __builtin_amdgcn_make_buffer_rsrc(HAS_NEXT ? p.t1n : p.out, 0, HAS_NEXT ? p.t1n_bytes : 0u, 0x00020000);

  // per-lane view of a wave tile: pixel m (row ho, column wo) of each 16-pixel MFMA tile and, per tap, the byte offset of
  // this lane's 16-byte piece (chunk fq of channel block 0) in t1 - or an out-of-range offset when the tap falls outside
  // the image or the pixel is >= M (the load then returns zeros without touching memory).
  struct Geom { int m, ho, wo; };
  auto tap_offsets = [&](const Geom& g, int (&voff)[9]) __attribute__((always_inline)) {
    const bool live = g.m < p.M;
    bool rok[3], cok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rok[i] = live && (unsigned)(g.ho + tdy[i]) < (unsigned)p.H;
      cok[i] = (unsigned)(g.wo + tdx[i]) < (unsigned)p.W;
    }
    const int base = g.m * 128 + fq * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t) voff[t] = (rok[t / 3] && cok[t % 3]) ? base + toff[t] : (int)0x80000000;
  };
  auto advance = [&](Geom& g) __attribute__((always_inline)) {
    g.m += step_px;
    g.wo += dw;
    const int c = g.wo >= p.W ? 1 : 0;
    g.wo -= c ? p.W : 0;
    g.ho += dh + c;
    g.ho -= g.ho >= p.H ? p.H : 0;
  };

  u32x4 a[18][TP];      // conv2 operand fragments of the CURRENT tile (refilled in place with the next one)
  u32x4 r[TP][8];       // residual runs of the current tile
  Geom g[TP];
  int m_cur[TP];
  int voff[TP][9];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    g[j].m = (wt * TP + j) * 16 + fr;
    const int rem = g[j].m % HW;
    g[j].ho = rem / p.W;
    g[j].wo = rem - g[j].ho * p.W;
    tap_offsets(g[j], voff[j]);
#pragma unroll
    for (int s = 0; s < 18; ++s) a[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_t1, voff[j][s >> 1] + (s & 1) * 64, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) r[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, g[j].m * 512 + q * 64 + fq * 16, 0, 0);
  }

  for (; wt < n_wt; wt += wt_step) {
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      m_cur[j] = g[j].m;
      advance(g[j]);                  // past the end: every pixel >= M
      tap_offsets(g[j], voff[j]);
    }

    u32x4 wfA[4], wfB[4];
    u32x4 tf[2][TP];      // t2 (conv3's B fragments)
    u32x4 xf[8][TP];      // block output (conv1''s B fragments)
    f32x4 acc[4][TP];
#pragma unroll
    for (int i = 0; i < 4; ++i) wfA[i] = *reinterpret_cast<const u32x4*>(w2_s + i * 1024 + rd);

    static_for<0, NSTEP>([&](auto kk) {
      constexpr int K = decltype(kk)::value;
      u32x4 (&wf)[4] = (K & 1) ? wfB : wfA;
      u32x4 (&wn)[4] = (K & 1) ? wfA : wfB;
      // weight fragments of the next step fly while this step's MFMAs run
      if constexpr (K + 1 < NSTEP) {
        const unsigned char* nx = tail_wfrag_addr<K + 1>(w2_s, w3_s, w1_s) + rd;
#pragma unroll
        for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const u32x4*>(nx + i * 1024);
      }
      if constexpr (K == 0 || (K >= 18 && ((K - 18) & 1) == 0 && K < 26) || K == 26) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (K < 18) {
        // ---- conv2, K plane K: B = tap fragments in registers
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) Mma<T>::run(wf[i], a[K][j], acc[i][j]);
        if (!(DP_EXP & 1)) {
#pragma unroll
          for (int j = 0; j < TP; ++j) a[K][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_t1, voff[j][K >> 1] + (K & 1) * 64, 0, 0);
        }
        if constexpr (K == 17) {
          // t2 = relu(acc + b2), rounded to the storage type: run h of pixel tile j is conv3's B fragment of K plane h
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b2_s + h * 32 + fq * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b2_s + h * 32 + fq * 8 + 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = fmaxf(acc[2 * h][j][k] + b0[k], 0.f);
                v[4 + k] = fmaxf(acc[2 * h + 1][j][k] + b1[k], 0.f);
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) tf[h][j][k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
            }
          }
        }
      } else if constexpr (K < 26) {
        // ---- conv3, 64-cout block b, K plane s (+ bias + residual + ReLU -> out after the second plane)
        constexpr int b = (K - 18) >> 1, sp = (K - 18) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) Mma<T>::run(wf[i], tf[sp][j], acc[i][j]);
        if constexpr (sp == 1) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            constexpr int qb = 2 * b;
            const int q = qb + h;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b3_s + q * 32 + fq * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b3_s + q * 32 + fq * 8 + 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = acc[2 * h][j][k] + b0[k];
                v[4 + k] = acc[2 * h + 1][j][k] + b1[k];
              }
              const u32x4 rv = r[j][q];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[2 * k] += Elem<T>::unpack(rv[k] & 0xffffu);
                v[2 * k + 1] += Elem<T>::unpack(rv[k] >> 16);
              }
              if (!(DP_EXP & 2)) r[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, g[j].m * 512 + q * 64 + fq * 16, 0, 0);
              u32x4 pk;
#pragma unroll
              for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(fmaxf(v[2 * k], 0.f), fmaxf(v[2 * k + 1], 0.f));
              if (!(DP_EXP & 4) || pk[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, m_cur[j] * 512 + q * 64 + fq * 16, 0, 0);
              xf[q][j] = pk;
            }
          }
        }
      } else {
        // ---- conv1' of the next block: K = the 256 channels just produced (plane q = run q above)
        constexpr int q = K - 26;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) Mma<T>::run(wf[i], xf[q][j], acc[i][j]);
        if constexpr (q == 7) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1_s + h * 32 + fq * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b1_s + h * 32 + fq * 8 + 4);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = fmaxf(acc[2 * h][j][k] + b0[k], 0.f);
                v[4 + k] = fmaxf(acc[2 * h + 1][j][k] + b1[k], 0.f);
              }
              u32x4 pk;
#pragma unroll
              for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
              if (!(DP_EXP & 4) || pk[0] == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(pk, rs_t1n, m_cur[j] * 128 + h * 64 + fq * 16, 0, 0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
}

// =====================================================================================================
// Strip walker: the same chain, with every byte of t1 / residual / output moved in whole 128-byte lines.
//
// The tile kernel above loads conv2's operand as MFMA fragments (16 pixels x 64 bytes per instruction, 18 per tile) and the
// residual / output as 64-byte runs: 36 KB of vector-memory traffic per 16-pixel tile of which 18 KB re-read t1 nine times.
// A CU sustains only ~25 one-KiB vector-memory instructions per microsecond on such streams (PMC: the TA command / address
// FIFOs are full 30 % of the time, profiles/r2_*), so that traffic - not HBM - bounded the kernel at 3.5 TB/s.
//
// Here a wave walks DOWN a 16-pixel-wide column strip of one image, L rows per job:
//   * t1 is read once per row in its natural layout (18 pixels x 128 B = the strip plus one halo pixel each side, three
//     whole-line loads, 8 pixels each); the wave keeps the rows r-1, r, r+1 in registers and loads only row r+2 per step.
//     Zero padding is decided at the load (out-of-range buffer offsets return 0): no per-fragment masking at all.
//   * a 2.25 KiB per-wave LDS buffer turns a natural-layout row into the 6 MFMA B fragments of a kernel row (3 column taps x
//     2 channel blocks, the column shift is an LDS address offset), the residual's whole lines into the epilogue's runs and
//     the epilogue's runs back into whole lines for the stores. It is wave-private (LDS requests of one wave execute in
//     order): no barrier anywhere in the loop. Chunk c of pixel p sits at slot c ^ (p & 7): every access pattern used here
//     (line-shaped stores / loads, fragment-shaped loads / stores at any pixel shift) is bank-conflict free.
//   * per 16 pixels: 3 + 8 + 8 + 2 = 21 full-line vector-memory instructions instead of 36 fragment-shaped ones.
// K order, rounding points and arithmetic are those of the tile kernel: results stay bit-identical to the layer-by-layer path.
// =====================================================================================================
constexpr int kStripBuf = 18 * 128;                     // per-wave row buffer: 18 pixels x 64 channels
constexpr int kStripLds = kTailLds + 8 * kStripBuf;     // 159,232 B of the 163,840 B a workgroup may use

struct StripArgs {
  TailArgs t;
  int n_strips, n_seg, seg_rows, n_jobs;
};

#if DP_EXP & 64   // ablation (timing only): no MFMAs, one VALU op keeps the operands alive
#define STRIP_MMA(a, b, c) ((c)[0] += __builtin_bit_cast(float, (a)[0] ^ (b)[0]))
#else
#define STRIP_MMA(a, b, c) Mma<T>::run(a, b, c)
#endif
template <typename T, bool HAS_NEXT, bool HAS_SC>
__global__ __launch_bounds__(DP_STRIP_NW * 64, DP_STRIP_NW / 4) void bottleneck_strip64_kernel(const StripArgs sa) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(!(HAS_NEXT && HAS_SC), "the shortcut's weight planes live where conv1''s would");
  constexpr int NW = DP_STRIP_NW;
  constexpr int C3 = HAS_SC ? 4 : 2;             // K planes behind conv3's accumulators: t2 (2), then the block input (2)
  constexpr int K3END = 18 + 4 * C3;
  constexpr int NSTEP = HAS_NEXT ? 34 : K3END;
  constexpr int RB = HAS_SC ? 1 : 4;             // 128-byte line groups per pixel of the residual / shortcut-input row
  const TailArgs& p = sa.t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const w2_s = smem;
  unsigned char* const w3_s = smem + kTailW2;
  unsigned char* const w1_s = smem + kTailW2 + kTailW3;
  float* const b2_s = reinterpret_cast<float*>(smem + kTailW2 + kTailW3 + kTailW1);
  float* const b3_s = b2_s + 64;
  float* const b1_s = b3_s + 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const buf = smem + kTailLds + wave * kStripBuf;

  {  // weights -> LDS, once (same images as the tile kernel)
    const int srow = lane >> 2;
    const int scc = (lane & 3) ^ swz(srow);
    const unsigned char* __restrict__ w2 = reinterpret_cast<const unsigned char*>(p.w2) + dp_wtile_off(srow, 0, scc, p.kpad2 * 2 / 64);   // tiled weight matrix
    for (int piece = wave; piece < 18 * 4; piece += NW) {
      const int pl = piece >> 2, rg = piece & 3;
      __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w2 + ((long long)rg * (p.kpad2 * 2 / 64) + pl) * 1024),
                                       DP_LDS_PTR(w2_s + pl * 4096 + rg * 1024), 16, 0, 0);
    }
    const unsigned char* __restrict__ w3 = reinterpret_cast<const unsigned char*>(p.w3) + dp_wtile_off(srow, 0, scc, p.kpad3 * 2 / 64);   // tiled weight matrix
    for (int piece = wave; piece < C3 * 16; piece += NW) {     // (4 planes: the last two land in conv1''s 32 KiB)
      const int pl = piece >> 4, rg = piece & 15;
      __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w3 + ((long long)rg * (p.kpad3 * 2 / 64) + pl) * 1024),
                                       DP_LDS_PTR(w3_s + pl * 16384 + rg * 1024), 16, 0, 0);
    }
    if (HAS_NEXT) {
      const unsigned char* __restrict__ w1 = reinterpret_cast<const unsigned char*>(p.w1n) + dp_wtile_off(srow, 0, scc, p.kpad1n * 2 / 64);
      for (int piece = wave; piece < 8 * 4; piece += NW) {
        const int pl = piece >> 2, rg = piece & 3;
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(w1 + ((long long)rg * (p.kpad1n * 2 / 64) + pl) * 1024),
                                         DP_LDS_PTR(w1_s + pl * 4096 + rg * 1024), 16, 0, 0);
      }
    }
    if (tid < 64) b2_s[tid] = p.b2[tid];
    if (tid < 256) b3_s[tid] = p.b3[tid];
    if (HAS_NEXT && tid < 64) b1_s[tid] = p.b1n[tid];
  }
  __syncthreads();  // (vmcnt(0) + barrier) - the only workgroup barrier of the kernel

  const int fr = lane & 15, fq = lane >> 4;
  const int rd = fr * 64 + ((fq ^ swz(fr)) << 4);   // weight fragment read offset inside a 16-row group of a plane
  const int ps = lane >> 3, ci = lane & 7;          // line-shaped accesses: 8 lanes per pixel, 16-byte chunk ci of its 128 bytes
  // the three column taps of conv2 (pack.py enumerates the 3 x 3 taps row-major; tap-major K): this kernel needs the plain
  // 3 x 3, pad 1 geometry - column displacement -1, 0, +1 and row displacement -1, 0, +1 (checked on the host)
  const __amdgpu_buffer_rsrc_t rs_t1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t1), 0, p.t1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t1n = __builtin_amdgcn_make_buffer_rsrc(HAS_NEXT ? p.t1n : p.out, 0, HAS_NEXT ? p.t1n_bytes : 0u, 0x00020000);
  constexpr int OOB = (int)0x80000000;

  // LDS addresses inside the wave's buffer. Line shape: lane = (pixel slot ps, chunk ci), instruction i covers pixels 8i + ps.
  // Fragment shape: lane = (pixel fr, chunk 4 * half + fq). Chunk c of pixel px lives at px * 128 + ((c ^ (px & 7)) << 4).
  int line_a[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) line_a[i] = (8 * i + ps) * 128 + ((ci ^ ps) << 4);   // (8i + ps) & 7 == ps
  int frag_a[3][2];     // [column tap dx][channel block / run h]: pixel fr + dx
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int h = 0; h < 2; ++h) frag_a[dx][h] = (fr + dx) * 128 + (((h * 4 + fq) ^ ((fr + dx) & 7)) << 4);
  // 16-pixel tensors (residual, output, next t1) use buffer pixels 0..15: fragment shape at dx = 0, line shape i = 0, 1

  for (int job = blockIdx.x * NW + wave; job < sa.n_jobs; job += gridDim.x * NW) {
#if DP_STRIP_ORDER
    // neighbouring waves walk neighbouring strips of the same rows: a workgroup's eight waves read 8 x 16 pixels of a row side by side
    const int strip = job % sa.n_strips;
    const int jt = job / sa.n_strips;
    const int seg = jt % sa.n_seg;
    const int n = jt / sa.n_seg;
#else
    const int seg = job % sa.n_seg;
    const int jt = job / sa.n_seg;
    const int strip = jt % sa.n_strips;
    const int n = jt / sa.n_strips;
#endif
    const int r0 = seg * sa.seg_rows;
    const int r1 = min(r0 + sa.seg_rows, p.H);
    const int c0 = strip * 16;
    const int img = n * p.H;
    // per-lane column offsets (bytes), out of range -> OOB: t1 row loads (pixels c0 - 1 .. c0 + 16), 16-pixel line accesses
    int t1_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int px = 8 * i + ps, wo = c0 - 1 + px;
      t1_col[i] = (px < 18 && (unsigned)wo < (unsigned)p.W) ? wo * 128 + ci * 16 : OOB;
    }
    int px_col[2];      // pixel column of the 16-pixel line accesses (k = 0, 1), -1 when outside the image
#pragma unroll
    for (int k = 0; k < 2; ++k) px_col[k] = (c0 + 8 * k + ps < p.W) ? c0 + 8 * k + ps : -1;

    auto load_row = [&](int rr, u32x4 (&dst)[3]) __attribute__((always_inline)) {
      // rows outside the image are zero padding; rows outside [r0 - 1, r1] are never used by this job
      const bool rok = (unsigned)rr < (unsigned)p.H && rr <= r1;
      const int base = (img + rr) * p.W * 128;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t1, (rok && t1_col[i] != OOB) ? base + t1_col[i] : OOB, 0, 0);
    };
    auto res_load = [&](int rr, int b, int k) __attribute__((always_inline)) -> u32x4 {
      const bool ok = rr < r1 && px_col[k] >= 0;
      return __builtin_amdgcn_raw_buffer_load_b128(rs_res, ok ? ((img + rr) * p.W + px_col[k]) * (RB * 128) + b * 128 + ci * 16 : OOB, 0, 0);
    };

    u32x4 row[4][3];     // t1 rows r-1, r, r+1 (natural layout) and the row being prefetched
    u32x4 rres[RB][2];   // residual of row r: 64-cout block b, pixel octet k (whole 128-byte lines); HAS_SC: the block input's 64 channels
    load_row(r0 - 1, row[0]);
    load_row(r0, row[1]);
    load_row(r0 + 1, row[2]);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int k = 0; k < 2; ++k) rres[b][k] = res_load(r0, b, k);
#if DP_STRIP_DRAIN
    // The compiler's s_waitcnt at the row loop's header is the minimum over BOTH ways into it: behind the job's first loads above, the
    // residual lines are the youngest operations in flight (0 .. 7 behind them), so every ITERATION waited with vmcnt(2) / vmcnt(3) - for all
    // but the last two stores of the previous row - where the loop's own order leaves 10 - 18 operations behind those loads. With nothing
    // in flight at the loop's entry the header's count is the loop's own. (One full drain per job of seg_rows rows.)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
#endif

    for (int r = r0; r < r1; ++r) {
      if (!(DP_EXP & 1)) load_row(r + 2, row[3]);
      const int orow = (img + r) * p.W;      // first pixel of this output row

      constexpr int AH = DP_STRIP_AHEAD;
      u32x4 wfr[AH + 1][4]; // weight fragments: AH + 1 register sets, the reads run AH K steps ahead of the MFMAs
      u32x4 bfr[3][2];      // B fragments of the kernel row being multiplied: [dx][channel block]
      u32x4 tf[2];          // t2 (conv3's B fragments)
      u32x4 xf[8];          // block output (conv1''s B fragments)
      u32x4 rfr[RB][2];     // residual runs of the four 64-cout blocks (fragment shape); HAS_SC: the block input's two K planes
      f32x4 acc[4];
      static_for<0, AH>([&](auto ss) {
        constexpr int S = decltype(ss)::value;
        const unsigned char* nx = tail_wfrag_addr<S, C3>(w2_s, w3_s, w1_s) + rd;
#pragma unroll
        for (int i = 0; i < 4; ++i) wfr[S][i] = *reinterpret_cast<const u32x4*>(nx + i * 1024);
      });

      // natural-layout t1 row dy -> LDS -> the 6 fragments of kernel row dy
      auto stage_row = [&](const u32x4 (&src)[3]) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(buf + line_a[0]) = src[0];
        *reinterpret_cast<u32x4*>(buf + line_a[1]) = src[1];
        if (ps < 2) *reinterpret_cast<u32x4*>(buf + line_a[2]) = src[2];     // pixels 16, 17 (the buffer ends there)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) bfr[dx][cb] = *reinterpret_cast<const u32x4*>(buf + frag_a[dx][cb]);
      };
      // whole lines of the finished 64-cout block bb: LDS (fragment shape, written by the epilogue) -> line shape -> store
      auto flush_out = [&](int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const u32x4 ln = *reinterpret_cast<const u32x4*>(buf + line_a[k]);
          if (!(DP_EXP & 4) || ln[0] == 0x12345678u)
            __builtin_amdgcn_raw_buffer_store_b128(ln, rs_out, px_col[k] >= 0 ? (orow + px_col[k]) * 512 + bb * 128 + ci * 16 : OOB, 0, 0);
        }
      };
      stage_row(row[0]);

      static_for<0, NSTEP>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        u32x4 (&wf)[4] = wfr[K % (AH + 1)];
        if constexpr (K + AH < NSTEP) {   // weight fragments of step K + AH fly while the MFMAs of steps K .. K + AH - 1 run
          const unsigned char* nx = tail_wfrag_addr<K + AH, C3>(w2_s, w3_s, w1_s) + rd;
          if (!(DP_EXP & 32) || K < 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wfr[(K + AH) % (AH + 1)][i] = *reinterpret_cast<const u32x4*>(nx + i * 1024);
          }
        }
        if constexpr (K == 0 || (K >= 18 && K < K3END && (K - 18) % C3 == 0) || K == K3END) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (K < 18) {
          // ---- conv2, K plane K = (kernel row dy, column tap dx, channel block cb)
          constexpr int dy = K / 6, dx = (K % 6) >> 1, cb = K & 1;
#pragma unroll
          for (int i = 0; i < 4; ++i) STRIP_MMA(wf[i], bfr[dx][cb], acc[i]);
          if constexpr (K >= 1 && K <= RB) {
            // residual of 64-cout block K - 1 (the buffer is idle between two kernel rows): whole lines -> LDS -> the two
            // runs this lane adds in conv3's epilogue; the line registers are refilled with the next row's residual
            constexpr int b = K - 1;
            *reinterpret_cast<u32x4*>(buf + line_a[0]) = rres[b][0];
            *reinterpret_cast<u32x4*>(buf + line_a[1]) = rres[b][1];
            rfr[b][0] = *reinterpret_cast<const u32x4*>(buf + frag_a[0][0]);
            rfr[b][1] = *reinterpret_cast<const u32x4*>(buf + frag_a[0][1]);
            if (!(DP_EXP & 2)) {
              rres[b][0] = res_load(r + 1, b, 0);
              rres[b][1] = res_load(r + 1, b, 1);
            }
          }
          if constexpr (K % 6 == 5 && dy < 2) stage_row(row[dy + 1]);     // next kernel row (its reads land during the MFMAs above)
          if constexpr (K == 17) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(b2_s + h * 32 + fq * 8);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(b2_s + h * 32 + fq * 8 + 4);
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = fmaxf(acc[2 * h][k] + b0[k], 0.f);
                v[4 + k] = fmaxf(acc[2 * h + 1][k] + b1[k], 0.f);
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) tf[h][k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
            }
          }
        } else if constexpr (K < K3END) {
          // ---- conv3, 64-cout block b, K plane sp (planes 2, 3: the projection shortcut over the block input)
          constexpr int b = (K - 18) / C3, sp = (K - 18) % C3;
#pragma unroll
          for (int i = 0; i < 4; ++i) STRIP_MMA(wf[i], sp < 2 ? tf[sp & 1] : rfr[0][sp & 1], acc[i]);
          if constexpr (sp == 0 && b > 0) flush_out(b - 1);      // the previous block's lines, one step after they were written
          if constexpr (sp == C3 - 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              constexpr int qb = 2 * b;
              const int q = qb + h;
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(b3_s + q * 32 + fq * 8);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(b3_s + q * 32 + fq * 8 + 4);
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = acc[2 * h][k] + b0[k];
                v[4 + k] = acc[2 * h + 1][k] + b1[k];
              }
              if constexpr (!HAS_SC) {
                const u32x4 rv = rfr[b][h];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  v[2 * k] += Elem<T>::unpack(rv[k] & 0xffffu);
                  v[2 * k + 1] += Elem<T>::unpack(rv[k] >> 16);
                }
              }
              u32x4 pk;
#pragma unroll
              for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(fmaxf(v[2 * k], 0.f), fmaxf(v[2 * k + 1], 0.f));
              xf[q] = pk;
              *reinterpret_cast<u32x4*>(buf + frag_a[0][h]) = pk;      // runs -> LDS (-> whole lines in flush_out)
            }
            if constexpr (b == 3 && !HAS_NEXT) flush_out(3);
          }
        } else {
          // ---- conv1' of the next block: K = the 256 channels just produced (plane q = run q above)
          constexpr int q = K - 26;
#pragma unroll
          for (int i = 0; i < 4; ++i) STRIP_MMA(wf[i], xf[q], acc[i]);
          if constexpr (q == 0) flush_out(3);
          if constexpr (q == 7) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1_s + h * 32 + fq * 8);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(b1_s + h * 32 + fq * 8 + 4);
              float v[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[k] = fmaxf(acc[2 * h][k] + b0[k], 0.f);
                v[4 + k] = fmaxf(acc[2 * h + 1][k] + b1[k], 0.f);
              }
              u32x4 pk;
#pragma unroll
              for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
              *reinterpret_cast<u32x4*>(buf + frag_a[0][h]) = pk;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const u32x4 ln = *reinterpret_cast<const u32x4*>(buf + line_a[k]);
              if (!(DP_EXP & 4) || ln[0] == 0x12345678u)
                __builtin_amdgcn_raw_buffer_store_b128(ln, rs_t1n, px_col[k] >= 0 ? (orow + px_col[k]) * 128 + ci * 16 : OOB, 0, 0);
            }
          }
        }
        if (!(DP_EXP & 8)) __builtin_amdgcn_sched_barrier(0);
      });
      // the three live rows slide down by one
#pragma unroll
      for (int i = 0; i < 3; ++i) { row[0][i] = row[1][i]; row[1][i] = row[2][i]; row[2][i] = row[3][i]; }
    }
  }
}

static int tail_num_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}

template <typename T, bool HAS_NEXT>
int launch_tail(const TailArgs& a, hipStream_t stream) {
  constexpr int NW = DP_TAIL_NW, TP = DP_TAIL_TP;
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_tail64_kernel<T, HAS_NEXT, NW, TP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kTailLds);
    cus = tail_num_cus();
    attr_set = true;
  }
  const int n_wt = (a.M + 16 * TP - 1) / (16 * TP);
  int gx = cus;
  if (gx > (n_wt + NW - 1) / NW) gx = (n_wt + NW - 1) / NW;
  hipLaunchKernelGGL((bottleneck_tail64_kernel<T, HAS_NEXT, NW, TP>), dim3(gx), dim3(NW * 64), kTailLds, stream, a);
  return dp_check_launch("bottleneck_tail64_kernel");
}

template <typename T, bool HAS_NEXT, bool HAS_SC = false>
int launch_strip(const TailArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_strip64_kernel<T, HAS_NEXT, HAS_SC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kStripLds);
    cus = tail_num_cus();
    attr_set = true;
  }
  StripArgs sa;
  sa.t = a;
  sa.n_strips = (a.W + 15) / 16;
  // rows per job: about two jobs per wave of a full chip (each job re-reads one halo row above and below its rows), at
  // least 4 rows; the split does not touch the arithmetic of a pixel, so results do not depend on it
  const long long cols = (long long)a.N * sa.n_strips;
  long long want = (2ll * cus * DP_STRIP_NW + cols - 1) / cols;
  if (want < 1) want = 1;
  int seg_rows = (int)((a.H + want - 1) / want);
  if (seg_rows < 4) seg_rows = a.H < 4 ? a.H : 4;
  sa.seg_rows = seg_rows;
  sa.n_seg = (a.H + seg_rows - 1) / seg_rows;
  const long long jobs = cols * sa.n_seg;
  sa.n_jobs = (int)jobs;
  int gx = (int)((jobs + DP_STRIP_NW - 1) / DP_STRIP_NW);
  if (gx > cus) gx = cus;
  hipLaunchKernelGGL((bottleneck_strip64_kernel<T, HAS_NEXT, HAS_SC>), dim3(gx), dim3(DP_STRIP_NW * 64), kStripLds, stream, sa);
  return dp_check_launch("bottleneck_strip64_kernel");
}

}  // namespace

// 1 when dp_bottleneck_tail_nhwc has a fused kernel for these parameters (else the caller runs the three layers through
// dp_conv2d_nhwc); the checks are the ones dp_bottleneck_tail_nhwc enforces.
static const char* tail_unsupported(const dp_bottleneck_params* p) {
  if (p->dtype != DP_BF16 && p->dtype != DP_F16) return "16-bit storage only";
  if (p->Cmid != 64 || p->Cout != 256) return "only the 64 -> 64 -> 256 (res2) shape is fused";
  if (p->ntaps2 != 9 || p->Kpad2 != 576) return "conv2 must be 3x3 over 64 channels (Kpad 576)";
  if (p->k_order2 != 1) return "conv2 must be packed tap-major (K = tap * 64 + channel)";
  if (p->sc_in) {    // first block of the stage: the projection shortcut as K planes 2, 3 of conv3 (strip walker only)
    if (p->Csc != 64 || p->Kpad3 != 128) return "shortcut form: conv3 must be the dual-source matrix over 64 + 64 channels (Kpad 128)";
    if (p->residual || p->next_t1) return "shortcut form: no residual tensor and no next-conv1 stage (its weights' LDS holds the shortcut planes)";
    if (p->hi_off2 != -1 || p->wi_off2 != -1) return "shortcut form: conv2 must be the plain 3x3, pad 1";
  } else if (p->Kpad3 != 64) {
    return "conv3 must be 1x1 over 64 channels (Kpad 64)";
  }
  if (p->next_t1 && (p->Cmid_next != 64 || p->Kpad1n != 256)) return "next conv1 must be 256 -> 64 (Kpad 256)";
  const long long M = (long long)p->N * p->H * p->W;
  // 32-bit buffer offsets; pixels up to one grid stride of tiles past the end are addressed before the range check drops them
  if ((M + (1ll << 16)) * 512 >= (1ll << 31)) return "tensor too large for 32-bit buffer offsets (split the batch)";
  return nullptr;
}

extern "C" int dp_bottleneck_tail_supported(const dp_bottleneck_params* p) {
  if (!p) return 0;
  return tail_unsupported(p) == nullptr ? 1 : 0;
}

extern "C" int dp_bottleneck_tail_nhwc(const dp_bottleneck_params* p, dp_stream_t stream) {
  DP_REQUIRE(p != nullptr, "dp_bottleneck_tail_nhwc: null params");
  DP_REQUIRE(p->N >= 0 && p->H > 0 && p->W > 0, "dp_bottleneck_tail_nhwc: bad spatial shape");
  const char* why = tail_unsupported(p);
  if (why) return dp_fail(DP_ERR_UNSUPPORTED, "dp_bottleneck_tail_nhwc: %s", why);
  const long long M = (long long)p->N * p->H * p->W;
  if (M == 0) return DP_OK;
  DP_REQUIRE(p->t1 && (p->residual || p->sc_in) && p->out && p->w2 && p->w3 && p->ktab2 && p->b2 && p->b3, "dp_bottleneck_tail_nhwc: null pointer");
  DP_REQUIRE(!p->next_t1 || (p->w1n && p->b1n), "dp_bottleneck_tail_nhwc: next_t1 given without its weights");
  TailArgs a;
  a.t1 = p->t1; a.res = p->sc_in ? p->sc_in : p->residual; a.out = p->out; a.t1n = p->next_t1;
  a.w2 = p->w2; a.w3 = p->w3; a.w1n = p->w1n; a.ktab2 = reinterpret_cast<const i32x4*>(p->ktab2);
  a.b2 = p->b2; a.b3 = p->b3; a.b1n = p->b1n;
  a.N = p->N; a.H = p->H; a.W = p->W; a.M = (int)M;
  a.kpad2 = p->Kpad2; a.kpad3 = p->Kpad3; a.kpad1n = p->Kpad1n;
  a.hi_off = p->hi_off2; a.wi_off = p->wi_off2;
  a.t1_bytes = (unsigned)(M * 128); a.out_bytes = (unsigned)(M * 512); a.t1n_bytes = (unsigned)(M * 128);
  a.res_bytes = p->sc_in ? (unsigned)(M * 128) : a.out_bytes;
  hipStream_t s = as_stream(stream);
  if (p->sc_in) {
    DP_REQUIRE((long long)p->N * ((p->W + 15) / 16) * p->H < (1ll << 30), "dp_bottleneck_tail_nhwc: too many strip jobs");
    return p->dtype == DP_BF16 ? launch_strip<uint16_t, false, true>(a, s) : launch_strip<f16_t, false, true>(a, s);
  }
  // two kernels for the same arithmetic: the strip walker (whole-line memory traffic, default whenever conv2 is the plain
  // 3x3 / pad 1 it is written for) and the tile kernel (any tap offsets; policy key tail_kernel = 1 selects it for A/B runs)
  const bool strip = dp_policy().tail_kernel == 0 && p->hi_off2 == -1 && p->wi_off2 == -1 && (long long)p->N * ((p->W + 15) / 16) * p->H < (1ll << 30);
  if (strip) {
    if (p->dtype == DP_BF16) return p->next_t1 ? launch_strip<uint16_t, true>(a, s) : launch_strip<uint16_t, false>(a, s);
    return p->next_t1 ? launch_strip<f16_t, true>(a, s) : launch_strip<f16_t, false>(a, s);
  }
  if (p->dtype == DP_BF16) return p->next_t1 ? launch_tail<uint16_t, true>(a, s) : launch_tail<uint16_t, false>(a, s);
  return p->next_t1 ? launch_tail<f16_t, true>(a, s) : launch_tail<f16_t, false>(a, s);
}
