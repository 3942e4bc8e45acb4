// NHWC implicit-GEMM convolution on MFMA for gfx950 (MI355X).
//
//   out[m][co] = act( sum_k A[m][k] * Wt[co][k] + bias[co] (+ residual) ),  m = (n, ho, wo), k = (tap, c)
//
// Replaces every F.conv2d / nn.Linear / ConvTranspose2d call site of the reference's hot path
// (detectron2/layers/wrappers.py:105-111 and the callers listed in include/densepose_hip.h).
//
// One file, five kernels that share the operand conventions below (DESIGN.md §4.1 has the measurements):
//   conv_igemm_kernel<BN,NPL>   generic: any Cin multiple of 8 (stem, Cout <= 64, tiny widths), per-lane tap table
//   conv_ring_kernel<256x256>   8 waves, 4-slot LDS ring of 64-byte K planes, counted vmcnt: the layers that carry the FLOPs
//   conv_ring_kernel<128x128>   4 waves, same ring: small-M layers
//   conv_ring2_kernel<256x128>  8 waves, two workgroups per CU: launches whose 256x256 tiling ends in a nearly empty round
//   conv1x1_stream_kernel<KP>   HBM-bound pointwise layers: resident weight slice, per-wave persistent tiles, no LDS for pixels
//
// Shared conventions (wave64):
//   * the weight tile is the MFMA "A" operand (rows = couts) and the im2col pixel tile the "B" operand (cols = pixels);
//     a wave always owns 64 couts (four 16x16 tiles) x TP pixel tiles. pack.py permutes the weight rows inside every
//     64-cout block so that a lane's 16 accumulator rows are two runs of 8 CONSECUTIVE output channels: the epilogue
//     (store_tile) adds bias / residual, applies ReLU and stores 16-byte pieces straight from registers.
//   * operands travel global/L2 -> LDS by LDS-DMA (global_load_lds / buffer_load ... lds, 16 B per lane, 1 KiB per wave
//     instruction, lane-linear destination); zero padding / ragged M / K padding never predicate a load: the generic
//     kernel reads a 16-byte zero page, the ring kernels an out-of-range buffer offset (hardware returns zeros).
//   * LDS image: planes of [rows][64 B]; chunk c of row r sits at slot c ^ ((-(r>>2))&3) - applied to the per-lane SOURCE
//     address of the DMA - and the fragment ds_read_b128 is bank-conflict free.
//   * the K axis is table driven (ktab: {dy, dx, c0, valid | tap << 8} per 16-byte chunk), so 1x1 / 3x3 / dilated /
//     7x7-stem / 2x2 sub-pixel (deconv) / fully-connected layers all run through the same kernels; multi-tap layers are
//     packed channel-block major, taps inner, so consecutive K planes re-read the same pixels shifted by one tap (L2 hits).
//   * DP_BF16: v_mfma_f32_16x16x32_bf16, DP_F16: v_mfma_f32_16x16x32_f16 (fp32 accumulate). DP_F32 (parity mode):
//     v_mfma_f32_16x16x4_f32, bit-exact fp32 FMA chain; same LDS image in bytes, 4 MFMAs per fragment instead of 1.
//   * workgroup ids are remapped so that consecutive tiles (same pixel rows, neighbouring cout tiles) run on the same
//     XCD and share its L2.
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"
#include <type_traits>

namespace {

constexpr int kBM = 128;       // pixels per workgroup
constexpr int kKB = 128;       // bytes of K per step
constexpr int kThreads = 256;

struct ConvArgs {
  const void* in;
  const void* weight;
  const i32x4* ktab;
  const float* bias;
  const void* residual;
  void* out;
  int N, H, W, Cin, Ho, Wo, Cout, Kpad, stride, stride_w, hi_off, wi_off, relu, rshift, out_f32;
  long long osN, osH, osW, rsN, rsH, rsW;
  int M, tiles_n, n_ktiles, HoWo, n_tiles;
  unsigned in_bytes, w_bytes, res_bytes;   // buffer-resource extents (ring / streaming kernels)
  int ntaps, out_linear, res_linear;
  const int* n_dev;       // device-side count of live images (dp_conv_params.n_dev): tiles that start behind them exit at once
  const void* in2;        // second source of a pointwise layer (dp_conv_params.in2): K channels >= Cin come from it
  int H2, W2, Cin2, stride2;
  unsigned in2_bytes;
  const void* head_w;     // fused 1x1 head (RPN): plain [16][Cout] storage type, rows = head channels
  const float* head_b;    // [16]
  float* head_out;        // [M][16] fp32
  int split_k;            // > 1: the K planes are cut into split_k segments, workgroup (tile, blockIdx.y) accumulates segment blockIdx.y
  float* split_ws;        //      and writes its raw fp32 sums to split_ws[segment][M][Cout] (dp_conv_params.split_k)
  int n_groups;           // > 1: grouped launch (dp_conv_params.n_groups): tile = (pixel tile, group, cout tile); group g runs on
  const void* weight_g[4];   //  weight_g[g] / ktab_g[g] and stores at out_g[g]; tiles_n counts the cout tiles of ONE group
  const i32x4* ktab_g[4];
  void* out_g[4];
};

// 16 zero bytes in global memory: out-of-image / K-padding chunks are loaded from here, so every staging load is
// unconditional (a predicated load makes hipcc branch around it and wait vmcnt(0) per load: serialised round trips).
__device__ u32x4 g_zero16 = {0u, 0u, 0u, 0u};

// ---- register epilogue ----------------------------------------------------------------------------------------
// Every kernel below gives a wave 64 couts (four 16x16 MFMA tiles i = 0..3 along cout, weights as the A operand) x TP
// pixel tiles, so lane (fr = lane & 15, fq = lane >> 4) holds rows fq*4 .. fq*4+3 of each cout tile for pixel fr of each
// pixel tile. pack.py stores the weight rows of every 64-cout block PERMUTED - physical row i*16 + fq*4 + e carries
// logical cout (i>>1)*32 + fq*8 + (i&1)*4 + e - so those 16 accumulator rows are two runs of 8 CONSECUTIVE output
// channels of one pixel. Bias / residual / ReLU / convert happen in registers and each lane writes 16-byte pieces, the
// four lanes of a pixel covering 64 contiguous bytes of the NHWC row per instruction: no LDS staging tile, no barrier,
// and the operand ring stays free for the next tile.
template <typename T, int TP>
__device__ __forceinline__ void store_tile(const ConvArgs& p, void* out, const f32x4 (&acc)[4][TP], int m_wave, int n_wave, int fr, int fq) {
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  const bool f32_out = p.out_f32 || sizeof(T) == 4;
  const bool linear = p.out_linear && (!res || p.res_linear);
  float bias[2][8];
  bool live[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = n_wave + h * 32 + fq * 8;
    live[h] = c < p.Cout;   // Cout is a multiple of 8: a run of 8 channels is all in or all out
    const f32x4 b0 = live[h] ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 b1 = live[h] ? *reinterpret_cast<const f32x4*>(p.bias + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { bias[h][k] = b0[k]; bias[h][4 + k] = b1[k]; }
  }
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int m_raw = m_wave + j * 16 + fr;
    const bool row_ok = m_raw < p.M;
    const int m = row_ok ? m_raw : p.M - 1;   // loads stay unconditional (clamped row), only the stores are predicated
    long long ob, rb = 0;
    if (linear) {
      ob = (long long)m * p.osW;
      rb = (long long)m * p.rsW;
    } else {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      ob = n * p.osN + ho * p.osH + wo * p.osW;
      if (res) rb = n * p.rsN + (ho >> p.rshift) * p.rsH + (wo >> p.rshift) * p.rsW;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!live[h]) continue;
      const int c = n_wave + h * 32 + fq * 8;
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = acc[2 * h][j][k] + bias[h][k];
        v[4 + k] = acc[2 * h + 1][j][k] + bias[h][4 + k];
      }
      if (res) {
        float r[8];
        load8(res + rb + c, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += r[k];
      }
      if (p.relu) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      if (row_ok) {
        if (f32_out) store8(reinterpret_cast<float*>(out) + ob + c, v);
        else store8(reinterpret_cast<T*>(out) + ob + c, v);
      }
    }
  }
}

// split-K epilogue: the raw fp32 accumulators of one K segment -> ws[M][Cout] of that segment (no bias, no activation; the
// accumulator layout is store_tile's: two runs of 8 consecutive couts per lane and pixel)
template <int TP>
__device__ __forceinline__ void store_tile_partial(float* __restrict__ ws, int M, int Cout, const f32x4 (&acc)[4][TP], int m_wave, int n_wave,
                                                   int fr, int fq) {
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int m = m_wave + j * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = n_wave + h * 32 + fq * 8;
      if (c >= Cout) continue;
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = acc[2 * h][j][k]; v[4 + k] = acc[2 * h + 1][j][k]; }
      store8(ws + (long long)m * Cout + c, v);
    }
  }
}

// out[m][c] = act(bias[c] + ws[0][m][c] + ws[1][m][c] + ...), segments added in index order: the summation order of a pixel is
// fixed by the LAYER (its segment count), not by the batch or the tile it lands in
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long M, int Cout, const float* __restrict__ bias,
                                                             int relu, T* __restrict__ out, long long osW) {
  const int C8 = Cout >> 3;
  const long long total = M * C8, seg = M * Cout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    float v[8], t[8];
    load8(ws + m * Cout + c, v);
    for (int sg = 1; sg < S; ++sg) {
      load8(ws + sg * seg + m * Cout + c, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    load8(bias + c, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] += t[k];
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    store8(out + m * osW + c, v);
  }
}

// NPL = 64-byte K planes per pipeline stage: 2 (128-byte K step, 64 KiB LDS, 2 workgroups/CU) for compute-heavy layers,
// 1 (64-byte K step, 32 KiB LDS, 4 workgroups/CU) for the short-K, HBM-bound 1x1 layers where resident workgroups
// (bytes in flight per CU), not per-step efficiency, set the speed.
template <typename T, int BN, int NPL>
__global__ __launch_bounds__(kThreads, 4) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int A_PLANE = kBM * 64;
  constexpr int B_PLANE = BN * 64;
  constexpr int A_BUF = NPL * A_PLANE;
  constexpr int B_BUF = NPL * B_PLANE;
  constexpr int BUF = A_BUF + B_BUF;
  // staging: one global_load_lds_dwordx4 wave-instruction fills 16 rows x 64 B of one plane (1 KiB, lane-linear)
  constexpr int A_GROUPS = kBM / 16 / 4;  // 16-row groups per wave
  constexpr int B_GROUPS = BN / 16 / 4;
  // wave tiling: BN=128 -> 2(pixels) x 2(couts) waves of 64x64 ; BN=64 -> 4 x 1 waves of 32 x 64
  constexpr int WAVES_C = BN / 64;
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int TP = kBM / WAVES_P / 16;
  constexpr int TC = 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware tile id (bijective remap of blockIdx over 8 XCDs)
  int tile;
  {
    const int nwg = p.n_tiles, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const int mt = tile / p.tiles_n;
  const int nt = tile - mt * p.tiles_n;
  const int m0 = mt * kBM;
  const int n0 = nt * BN;
  if (p.n_dev != nullptr && m0 >= *p.n_dev * p.HoWo) return;   // (uniform: a scalar load) nothing live in this tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave % WAVES_P;
  const int wc = wave / WAVES_P;

  // ---- staging assignment. Lane l of a wave-instruction lands at LDS byte l*16 of the 1-KiB piece, i.e. physical
  //      (row = l>>2, slot = l&3); the slot holds logical chunk slot ^ swz(row), so the swizzle is applied to the
  //      per-lane SOURCE address (the LDS-DMA destination is lane-linear by construction).
  const int srow = lane >> 2;
  const int scc = (lane & 3) ^ swz(srow);  // logical 16-byte chunk inside the plane (0..3)

  int a_hi0[A_GROUPS], a_wi0[A_GROUPS], a_pix[A_GROUPS];
#pragma unroll
  for (int i = 0; i < A_GROUPS; ++i) {
    const int m = m0 + (wave * A_GROUPS + i) * 16 + srow;
    if (m < p.M) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride + p.hi_off;
      a_wi0[i] = wo * p.stride_w + p.wi_off;
      a_pix[i] = n * p.H * p.W;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_pix[i] = 0;
    }
  }
  const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
  // weights: 1 KiB tiles of 16 rows x 64 B (dp_wtile_off): this lane's chunk of row group (n0 / 16 + wave * B_GROUPS), plane 0
  const int w_planes = p.Kpad * ES / 64;
  const unsigned char* __restrict__ wgt =
      reinterpret_cast<const unsigned char*>(p.weight) + dp_wtile_off(n0 + wave * B_GROUPS * 16 + srow, 0, scc, w_planes);
  const T* __restrict__ zsrc = reinterpret_cast<const T*>(&g_zero16);
  unsigned char* const lds_a = smem + (wave * A_GROUPS) * 1024;           // + buf*BUF + plane*A_PLANE + i*1024
  unsigned char* const lds_b = smem + A_BUF + (wave * B_GROUPS) * 1024;   // + buf*BUF + plane*B_PLANE + i*1024

// out-of-image / K-padding chunks are fetched from a 16-byte zero page: every LDS-DMA is unconditional
#define DP_STAGE_TILE(KT_IDX, EN, BUF_IDX)                                                                        \
  {                                                                                                               \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                          \
      _Pragma("unroll") for (int i = 0; i < A_GROUPS; ++i) {                                                      \
        const int hi = a_hi0[i] + (EN)[pl][0], wi = a_wi0[i] + (EN)[pl][1];                                       \
        const bool ok = ((EN)[pl][3] & 1) && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;              \
        const long long off = (long long)(a_pix[i] + hi * p.W + wi) * p.Cin + (EN)[pl][2];                        \
        const T* src = ok ? (in + off) : zsrc;                                                                    \
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(src), DP_LDS_PTR(lds_a + (BUF_IDX) * BUF + pl * A_PLANE + i * 1024), 16, 0, 0); \
      }                                                                                                           \
      _Pragma("unroll") for (int i = 0; i < B_GROUPS; ++i) {                                                      \
        const unsigned char* wsrc = wgt + ((long long)i * w_planes + (KT_IDX) * NPL + pl) * 1024;                 \
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(wsrc), DP_LDS_PTR(lds_b + (BUF_IDX) * BUF + pl * B_PLANE + i * 1024), 16, 0, 0); \
      }                                                                                                           \
    }                                                                                                             \
  }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (row = lane&15 inside a 16-row tile, chunk = lane>>4)
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int rd_off = fr * 64 + ((fq ^ swz(fr)) << 4);
  const int a_row0 = (wp * TP * 16) * 64;     // pixel rows of this wave
  const int b_row0 = (wc * TC * 16) * 64;     // cout rows of this wave

  const int nk = p.n_ktiles * (2 / NPL);      // K steps (n_ktiles counts 128-byte steps)
  // tap-table entries of this lane's chunk(s) are fetched one K-step ahead of the loads that need them
  i32x4 en[NPL];
  {
    i32x4 e0[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) e0[pl] = p.ktab[pl * 4 + scc];
    DP_STAGE_TILE(0, e0, 0);
    const int k1 = nk > 1 ? 1 : 0;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) en[pl] = p.ktab[(k1 * NPL + pl) * 4 + scc];
  }
  __syncthreads();  // (drains the LDS-DMA: hipcc emits vmcnt(0) before the barrier)

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      DP_STAGE_TILE(kt + 1, en, cur ^ 1);
      const int kn = kt + 2 < nk ? kt + 2 : nk - 1;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) en[pl] = p.ktab[(kn * NPL + pl) * 4 + scc];
    }
    const unsigned char* sa = smem + cur * BUF + a_row0 + rd_off;
    const unsigned char* sb = smem + cur * BUF + A_BUF + b_row0 + rd_off;
#pragma unroll
    for (int ks = 0; ks < NPL; ++ks) {
      u32x4 fp[TP], fc[TC];
#pragma unroll
      for (int j = 0; j < TP; ++j) fp[j] = *reinterpret_cast<const u32x4*>(sa + ks * A_PLANE + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < TC; ++i) fc[i] = *reinterpret_cast<const u32x4*>(sb + ks * B_PLANE + i * 16 * 64);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) Mma<T>::run(fc[i], fp[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue straight from the accumulators ----
  store_tile<T, TP>(p, p.out, acc, m0 + wp * TP * 16, n0 + wc * 64, fr, fq);
}

#undef DP_STAGE_TILE

// =====================================================================================================
// LDS-ring kernels for the layers that carry the FLOPs (Cin*esize % 64 == 0, i.e. a 64-byte K plane lies inside
// one tap): K is consumed in 64-byte planes through a 4-slot LDS ring filled by global_load_lds; up to two planes stay
// in flight behind a COUNTED s_waitcnt vmcnt(N) + raw s_barrier (a plain __syncthreads() would drain them) while the
// MFMAs of the current plane run on fragments that were read from LDS during the previous plane's MFMAs (register
// double buffering). The tap-table entry of a plane is wave-uniform and fetched with scalar loads, so the only VMEM
// traffic of the loop is the LDS-DMA itself and the vmcnt arithmetic is exact.
//   WC = 4, TP = 8 : 256 pixels x 256 couts, 8 waves (64 couts x 128 pixels each), 4 x 32 KiB ring, 1 workgroup / CU
//   WC = 2, TP = 4 : 128 pixels x 128 couts, 4 waves (64 x 64 each),               4 x 16 KiB ring, 2 workgroups / CU
// =====================================================================================================
constexpr int kRing = 4;
#ifndef DP_RING_EXP
#define DP_RING_EXP 0     // diagnostic builds: 1 = no LDS-DMA issue in the steady loop, 2 = every piece reads one contiguous (cached) KiB
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
// `planes` whole K planes of this wave's LDS-DMA pieces (pw = 3 or 4 pieces per plane, wave-uniform) may stay in flight
__device__ __forceinline__ void wait_planes(int planes, int pw) {
  if (planes <= 0) wait_vmcnt<0>();
  else if (planes == 1) { if (pw == 4) wait_vmcnt<4>(); else wait_vmcnt<3>(); }
  else { if (pw == 4) wait_vmcnt<8>(); else wait_vmcnt<6>(); }
}

// DUAL: the pointwise layer has a second source (dp_conv_params.in2). A template parameter, not a runtime flag: the extra per-lane
// offsets, the second buffer resource and the per-plane source select cost the 224- and 256-row instances 21 - 25 % when they are
// merely PRESENT in the kernel (scalar registers 97 - 100 of 102), measured in round 3.
// SPLIT: split-K launch (dp_conv_params.split_k): grid.y = segment, K planes [seg * per, ...), raw fp32 sums to the workspace.
// A template parameter for the reason DUAL is one.
template <typename T, int WC, int TP, bool DUAL, bool SPLIT = false>
__global__ __launch_bounds__(WC * 2 * 64, 2) void conv_ring_kernel(const ConvArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int CH = 16 / ES;
  constexpr int TC = 4;
  constexpr int NW = WC * 2;              // waves
  constexpr int BM = 2 * TP * 16;         // pixels
  constexpr int BN = WC * 64;             // couts
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, SLOT = A_PLANE + B_PLANE;
  // staging: a plane has NA = BM/16 pixel pieces and BN/16 weight pieces of 16 rows (1 KiB, one wave instruction each).
  // Every wave stages two weight pieces and the pixel pieces {wave, wave + NW} that exist: tile heights that are not a
  // multiple of 16*NW rows (TP = 5, 6, 7) leave some waves with ONE pixel piece, so the per-plane piece count a wave
  // waits on (pw = 3 or 4) is a wave-uniform runtime value.
  constexpr int NA = BM / 16;
  static_assert(BN / 16 / NW == 2 && NA >= NW && NA <= 2 * NW, "two weight pieces and one or two pixel pieces per wave");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int tile;
  {
    const int nwg = p.n_tiles, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  // grouped launch (the chart predictor's four sub-pixel convolutions, chart.py:45-60): tile = (pixel tile, group, cout tile) - the
  // groups of a pixel tile are neighbours in the numbering, i.e. on one XCD, and read the same input rows
  const int tpg = p.tiles_n * (p.n_groups > 1 ? p.n_groups : 1);
  const int mt = tile / tpg;
  const int grp = (tile - mt * tpg) / p.tiles_n;
  const int nt = tile - mt * tpg - grp * p.tiles_n;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  if (p.n_dev != nullptr && m0 >= *p.n_dev * p.HoWo) return;   // (uniform: a scalar load) nothing live in this tile
  const void* const g_weight = p.n_groups > 1 ? p.weight_g[grp] : p.weight;
  const i32x4* const g_ktab = p.n_groups > 1 ? p.ktab_g[grp] : p.ktab;
  void* const g_out = p.n_groups > 1 ? p.out_g[grp] : p.out;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC;   // cout block (64)
  const int wp = wave / WC;   // pixel half

  const int srow = lane >> 2;
  const int scc = (lane & 3) ^ swz(srow);

  // constant address space: the only way to get s_load (a vector load here would sit in vmcnt and drain the ring)
  const __attribute__((address_space(4))) i32x4* ktab_c = (const __attribute__((address_space(4))) i32x4*)g_ktab;

  // per-lane descriptors of the two pixel rows this lane stages (rows 32*wave + 16*i + srow of the tile):
  // byte offset of (row, tap (0,0), this lane's chunk) - possibly "virtual" at the border - and a bit mask of the taps
  // that fall inside the image for this row. Per plane the staging then costs one add + one select per load:
  // out-of-image / K-padding chunks get an out-of-range buffer offset, for which the hardware returns zeros.
  const bool a2 = wave + NW < NA;          // this wave stages a second pixel piece
  const int pw = a2 ? 4 : 3;               // LDS-DMA pieces per plane of this wave
  int a_boff[2];
  unsigned a_okm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (wave + i * NW) * 16 + srow;   // pixel piece wave + i*NW
    a_okm[i] = 0u;
    a_boff[i] = 0;
    if (m < p.M && (i == 0 || a2)) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride + p.hi_off, wi0 = wo * p.stride_w + p.wi_off;
      a_boff[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + scc * CH) * ES;
      for (int t = 0; t < p.ntaps; ++t) {
        const i32x4 e = ktab_c[t * 4];  // planes 0..ntaps-1 enumerate the taps (channel-block-major packing)
        if ((unsigned)(hi0 + e[0]) < (unsigned)p.H && (unsigned)(wi0 + e[1]) < (unsigned)p.W) a_okm[i] |= 1u << t;
      }
    }
  }
  // split-K: this workgroup's K segment = planes [s_base, s_base + ns) (tap validity above is per TAP, from the table's head)
  int ns = p.n_ktiles * 2;  // number of 64-byte planes along K
  int s_base = 0;
  if constexpr (SPLIT) {
    const int per = (ns + p.split_k - 1) / p.split_k;
    s_base = blockIdx.y * per;
    ns = min(per, ns - s_base);
  }
  const __attribute__((address_space(4))) i32x4* ktab_k = ktab_c + s_base * 4;
  // weights: 1 KiB tiles of 16 rows x 64 B (dp_wtile_off): piece i of plane S = tile (n0 / 16 + wave * 2 + i, S)
  const int w_planes = p.Kpad * ES / 64;
  const int w_boff = (int)dp_wtile_off(n0 + wave * 32 + srow, 0, scc, w_planes) + s_base * 1024;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_weight), 0, p.w_bytes, 0x00020000);
  // Second source of a pointwise layer (the projection shortcut of a bottleneck's first block as extra K planes of its conv3,
  // resnet.py:189-205: out = relu(W3 t2 + Ws x[::s, ::s] + b)): K channels >= Cin are read from in2 at pixel (ho * stride2,
  // wo * stride2). The plane's source is wave-uniform (its channel offset comes from the scalar tap table).
  constexpr bool dual = DUAL;
  const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dual ? p.in2 : p.in), 0, dual ? p.in2_bytes : 0u, 0x00020000);
  // a_dboff = (offset in the second source) - (offset in the first): the per-plane pick is a_boff + (a_dboff & mask) with a wave-uniform
  // mask. Written as `src2 ? a_boff2[i] : a_boff[i]` hipcc merged the two arrays into ONE private array indexed by the uniform
  // condition - 20 bytes of scratch and three scratch_load_dword per two K planes between the MFMAs of the steady loop (round 3).
  int a_dboff[2] = {0, 0};
  if constexpr (dual) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (wave + i * NW) * 16 + srow;
      if (m < p.M && (i == 0 || a2)) {
        const int n = m / p.HoWo;
        const int rem = m - n * p.HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_dboff[i] = (((n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + scc * CH) * ES - a_boff[i];
      }
    }
  }
  unsigned char* const lds_sa = smem + wave * 1024;              // pixel piece `wave` (+ NW*1024: piece wave + NW)
  unsigned char* const lds_sb = smem + A_PLANE + wave * 2048;    // this wave's two weight pieces

#define DP_RING_STAGE(S_IDX, E)                                                                                    \
  {                                                                                                                \
    const bool src2_ = dual && (E)[2] >= p.Cin;                                                                    \
    const int tap_boff = src2_ ? ((E)[2] - p.Cin) * ES : (((E)[0] * p.W + (E)[1]) * p.Cin + (E)[2]) * ES;          \
    const unsigned tapbit = ((E)[3] & 1) ? (1u << ((E)[3] >> 8)) : 0u;                                             \
    const int slot_ = ((S_IDX) & (kRing - 1)) * SLOT;                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
      if (i == 0 || a2) {                                                                                          \
        const int off = (a_okm[i] & tapbit) ? (a_boff[i] + (a_dboff[i] & (src2_ ? -1 : 0)) + tap_boff) : (int)0x80000000; \
        if (src2_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, DP_LDS_PTR(lds_sa + slot_ + i * NW * 1024), 16, off, 0, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(lds_sa + slot_ + i * NW * 1024), 16, off, 0, 0, 0); \
      }                                                                                                            \
    }                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, DP_LDS_PTR(lds_sb + slot_ + i * 1024), 16,                    \
                                               w_boff + (i * w_planes + (S_IDX)) * 1024, 0, 0, 0);                 \
    }                                                                                                              \
  }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int rd_off = fr * 64 + ((fq ^ swz(fr)) << 4);
  const unsigned char* const rd_a = smem + (wp * TP * 16) * 64 + rd_off;
  const unsigned char* const rd_b = smem + A_PLANE + (wc * TC * 16) * 64 + rd_off;

#define DP_RING_READ(S_IDX, FP, FC)                                                                                \
  {                                                                                                                \
    const int slot_ = ((S_IDX) & (kRing - 1)) * SLOT;                                                              \
    _Pragma("unroll") for (int i = 0; i < TC; ++i) (FC)[i] = *reinterpret_cast<const u32x4*>(rd_b + slot_ + i * 16 * 64); \
    _Pragma("unroll") for (int j = 0; j < TP; ++j) (FP)[j] = *reinterpret_cast<const u32x4*>(rd_a + slot_ + j * 16 * 64); \
  }

// one K plane: make plane S+1 visible, refill the slot of plane S-1 with plane S+3, prefetch the fragments of plane S+1
// into the other register set, then run the MFMAs of plane S on the current set
#define DP_RING_STEP(S_IDX, FP_CUR, FC_CUR, FP_NXT, FC_NXT)                                                        \
  {                                                                                                                \
    if ((S_IDX) + 1 < ns) {                                                                                        \
      wait_planes((S_IDX) + 2 < ns ? 1 : 0, pw);   /* only plane S+2 may still be in flight */                     \
      __builtin_amdgcn_s_barrier();                                                                                \
      if ((S_IDX) + 3 < ns) {                                                                                      \
        DP_RING_STAGE((S_IDX) + 3, e_nx);                                                                          \
        e_nx = ktab_k[((S_IDX) + 4 < ns ? (S_IDX) + 4 : (S_IDX) + 3) * 4];                                         \
      }                                                                                                            \
      DP_RING_READ((S_IDX) + 1, FP_NXT, FC_NXT);                                                                   \
    }                                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TC; ++i)                                                                 \
      _Pragma("unroll") for (int j = 0; j < TP; ++j) Mma<T>::run((FC_CUR)[i], (FP_CUR)[j], acc[i][j]);             \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  }

  {
    const i32x4 t0 = ktab_k[0];
    DP_RING_STAGE(0, t0);
    if (ns > 1) { const i32x4 t1 = ktab_k[4]; DP_RING_STAGE(1, t1); }
    if (ns > 2) { const i32x4 t2 = ktab_k[8]; DP_RING_STAGE(2, t2); }
  }
  // tap entry (wave-uniform, scalar load) of the plane staged in the NEXT step: fetched one step ahead of its use
  i32x4 e_nx = ktab_k[(ns > 3 ? 3 : 0) * 4];

  u32x4 fpA[TP], fcA[TC], fpB[TP], fcB[TC];
  // plane 0 landed once at most planes 1 and 2 (4 LDS-DMAs per plane per wave) are outstanding
  wait_planes(ns > 2 ? 2 : (ns > 1 ? 1 : 0), pw);
  __builtin_amdgcn_s_barrier();
  DP_RING_READ(0, fpA, fcA);

  // Steady-state step (planes S+1 .. S+3 all exist): branch-free, with the four LDS-DMA pieces of plane S+3 and the TC+TP
  // fragment reads of plane S+1 issued one at a time between groups of MFMAs of plane S (order pinned by sched_barrier
  // fences). Both waves of a SIMD start their MFMAs right after the barrier and each wave's memory-instruction issue
  // slots fall under its partner's MFMAs instead of both waves idling the matrix pipe together.
  constexpr int N_MEM = 4 + TC + TP;            // memory instructions per step
  constexpr int N_PAIR = TC * TP;               // fragment pairs (MFMA groups) per step
  auto steady = [&](int S, u32x4 (&fp_cur)[TP], u32x4 (&fc_cur)[TC], u32x4 (&fp_nxt)[TP], u32x4 (&fc_nxt)[TC]) __attribute__((always_inline)) {
    if (a2) wait_vmcnt<4>(); else wait_vmcnt<3>();   // only plane S+2 may still be in flight
    __builtin_amdgcn_s_barrier();
    const i32x4 e = e_nx;
    const bool src2 = dual && e[2] >= p.Cin;
    const int tap_boff = src2 ? (e[2] - p.Cin) * ES : ((e[0] * p.W + e[1]) * p.Cin + e[2]) * ES;
    const unsigned tapbit = (e[3] & 1) ? (1u << (e[3] >> 8)) : 0u;
    const int dslot = ((S + 3) & (kRing - 1)) * SLOT;
    const int rslot = ((S + 1) & (kRing - 1)) * SLOT;
    static_for<0, N_PAIR>([&](auto pi) {
      constexpr int pr = decltype(pi)::value;
      Mma<T>::run(fc_cur[pr / TP], fp_cur[pr % TP], acc[pr / TP][pr % TP]);
      if constexpr (pr == 1) {
        // The scalar load of the NEXT step's tap entry is pinned behind the first MFMAs. Left to the scheduler it sometimes lands
        // in front of them, directly followed by the s_waitcnt lgkmcnt(0) that guards this step's fragment registers: the wave then
        // sits out a scalar-cache round trip after every barrier (round 3: the same source compiled to a 128x256 instance 23 %
        // slower after an unrelated template parameter was added; the ISA differed in exactly this placement).
        __builtin_amdgcn_sched_barrier(0);
        e_nx = ktab_k[min(S + 4, ns - 1) * 4];
        __builtin_amdgcn_sched_barrier(0);
      }
      // memory instructions spread evenly over the MFMA groups
      constexpr int m_lo = pr * N_MEM / N_PAIR, m_hi = (pr + 1) * N_MEM / N_PAIR;
      if constexpr (m_hi > m_lo) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<m_lo, m_hi>([&](auto mi) {
          constexpr int m = decltype(mi)::value;
          if constexpr (DP_RING_EXP & 1) {
            if constexpr (m >= 4) {     // diagnostic build: no LDS-DMA issue in the steady loop (results are garbage, timing only)
              if constexpr (m < 4 + TC) fc_nxt[m - 4] = *reinterpret_cast<const u32x4*>(rd_b + rslot + (m - 4) * 16 * 64);
              else fp_nxt[m - 4 - TC] = *reinterpret_cast<const u32x4*>(rd_a + rslot + (m - 4 - TC) * 16 * 64);
            }
          } else if constexpr (m < 2) {
            if (m == 0 || a2) {
              int off = (a_okm[m] & tapbit) ? (a_boff[m] + (a_dboff[m] & (src2 ? -1 : 0)) + tap_boff) : (int)0x80000000;
              if constexpr (DP_RING_EXP & 2) off = lane * 16;     // diagnostic: same instruction count, one contiguous KiB per piece
              if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, DP_LDS_PTR(lds_sa + dslot + m * NW * 1024), 16, off, 0, 0, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(lds_sa + dslot + m * NW * 1024), 16, off, 0, 0, 0);
            }
          } else if constexpr (m < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, DP_LDS_PTR(lds_sb + dslot + (m - 2) * 1024), 16,
                                                     (DP_RING_EXP & 2) ? lane * 16 + (m - 2) * 1024 : w_boff + ((m - 2) * w_planes + (S + 3)) * 1024, 0, 0, 0);
          } else if constexpr (m < 4 + TC) {
            fc_nxt[m - 4] = *reinterpret_cast<const u32x4*>(rd_b + rslot + (m - 4) * 16 * 64);
          } else {
            fp_nxt[m - 4 - TC] = *reinterpret_cast<const u32x4*>(rd_a + rslot + (m - 4 - TC) * 16 * 64);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  int s = 0;
  for (; s + 5 <= ns; s += 2) {   // both steps of the pair have planes s+1 .. s+4 ahead of them
    steady(s, fpA, fcA, fpB, fcB);
    steady(s + 1, fpB, fcB, fpA, fcA);
  }
  for (; s < ns; s += 2) {        // tail: the generic step (conditional staging / counted waits)
    DP_RING_STEP(s, fpA, fcA, fpB, fcB);
    if (s + 1 < ns) DP_RING_STEP(s + 1, fpB, fcB, fpA, fcA);
  }

  // ---- fused 1x1 head (the RPN's objectness + anchor-delta convolutions, rpn.py:168-171, on the 3x3 conv's ReLU output):
  // head[16][px] = Wh[16][256] . relu(acc + bias). The permuted accumulator layout makes run h of pixel tile j, rounded to
  // the storage type, the B fragment of K step h (channels wc*64 + h*32 ..) of that product (dp_bottleneck.hip uses the same
  // property), so every wave contracts ITS 64 channels with two MFMAs per pixel tile. The four channel blocks (waves wc =
  // 0..3 of a pixel half) are chained THROUGH the accumulator: wave wc starts from wave wc-1's partial sums (handed over in
  // LDS), so the head is accumulated over K planes 0..7 in order - bit-identical to the separate 1x1 launch it replaces,
  // whatever kernel that launch would use for another batch size.
  if constexpr (WC == 4 && sizeof(T) == 2) {
    if (p.head_out != nullptr) {
      const T* __restrict__ hw = reinterpret_cast<const T*>(p.head_w) + fr * p.Cout + n0 + wc * 64 + fq * 8;
      const u32x4 hw0 = *reinterpret_cast<const u32x4*>(hw), hw1 = *reinterpret_cast<const u32x4*>(hw + 32);
      float bias[2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* bp = p.bias + n0 + wc * 64 + h * 32 + fq * 8;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[h][k] = b0[k]; bias[h][4 + k] = b1[k]; }
      }
      u32x4 bf[2][TP];   // this wave's 64 channels of the hidden tensor, as the head product's B fragments
#pragma unroll
      for (int j = 0; j < TP; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = fmaxf(acc[2 * h][j][k] + bias[h][k], 0.f);
            v[4 + k] = fmaxf(acc[2 * h + 1][j][k] + bias[h][4 + k], 0.f);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) bf[h][j][k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
        }
      }
      // every wave has finished reading the operand ring before it is reused for the hand-over
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      f32x4* const red = reinterpret_cast<f32x4*>(smem) + (wp * TP) * 64 + lane;   // [pixel half][TP][64 lanes]
      const f32x4 hb = *reinterpret_cast<const f32x4*>(p.head_b + fq * 4);
#pragma unroll
      for (int stage = 0; stage < 4; ++stage) {
        if (wc == stage) {
#pragma unroll
          for (int j = 0; j < TP; ++j) {
            f32x4 part = stage == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : red[j * 64];
            Mma<T>::run(hw0, bf[0][j], part);
            Mma<T>::run(hw1, bf[1][j], part);
            if (stage < 3) {
              red[j * 64] = part;
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k) part[k] += hb[k];
              const int m = m0 + wp * TP * 16 + j * 16 + fr;
              if (m < p.M) *reinterpret_cast<f32x4*>(p.head_out + (long long)m * 16 + fq * 4) = part;
            }
          }
        }
        if (stage < 3) __syncthreads();
      }
      if (p.out == nullptr) return;
    }
  }

  // ---- epilogue straight from the accumulators ----
  if constexpr (SPLIT) store_tile_partial<TP>(p.split_ws + (long long)blockIdx.y * p.M * p.Cout, p.M, p.Cout, acc, m0 + wp * TP * 16, n0 + wc * 64, fr, fq);
  else store_tile<T, TP>(p, g_out, acc, m0 + wp * TP * 16, n0 + wc * 64, fr, fq);
}
#undef DP_RING_STAGE
#undef DP_RING_READ
#undef DP_RING_STEP

template <typename T, int WC, int TP, bool DUAL>
int launch_conv_ring_d(const ConvArgs& a, hipStream_t stream) {
  constexpr int BM = 2 * TP * 16, BN = WC * 64;
  constexpr int lds = kRing * (BM + BN) * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<T, WC, TP, DUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_ring_kernel<T, WC, TP, DUAL>), dim3(a.n_tiles), dim3(WC * 2 * 64), lds, stream, a);
  return dp_check_launch("conv_ring_kernel");
}
template <typename T, int WC, int TP>
int launch_conv_ring(const ConvArgs& a, hipStream_t stream) {
  return a.in2 ? launch_conv_ring_d<T, WC, TP, true>(a, stream) : launch_conv_ring_d<T, WC, TP, false>(a, stream);
}
// split-K: grid (tiles, segments) of the SPLIT instance, then the reduction + bias + activation pass (16-bit storage only)
template <typename T, int WC, int TP>
int launch_conv_ring_split(const ConvArgs& a, int n_seg, hipStream_t stream) {
  if constexpr (sizeof(T) != 2) {
    return dp_fail(DP_ERR_UNSUPPORTED, "dp_conv2d_nhwc: split_k needs 16-bit storage");
  } else {
    constexpr int BM = 2 * TP * 16, BN = WC * 64;
    constexpr int lds = kRing * (BM + BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<T, WC, TP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_ring_kernel<T, WC, TP, false, true>), dim3(a.n_tiles, n_seg), dim3(WC * 2 * 64), lds, stream, a);
    const long long items = (long long)a.M * (a.Cout / 8);
    int g = (int)((items + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(g), dim3(256), 0, stream, a.split_ws, n_seg, (long long)a.M, a.Cout, a.bias, a.relu,
                       reinterpret_cast<T*>(a.out), a.osW);
    return dp_check_launch("conv_ring_kernel (split-K)");
  }
}

template <typename T, int BN, int NPL>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
  constexpr int buf = NPL * (kBM * 64 + BN * 64);   // one operand stage
  constexpr int full = 2 * buf;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BN, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize, full);
    attr_set = true;
  }
  // single-step layers never touch the second stage
  const int steps = a.n_ktiles * (2 / NPL);
  const int lds = steps == 1 ? buf : full;
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN, NPL>), dim3(a.n_tiles), dim3(kThreads), lds, stream, a);
  return dp_check_launch("conv_igemm_kernel");
}

template <typename T, int BN>
int launch_conv_k(const ConvArgs& a, hipStream_t stream) {
  // short-K layers (K <= 8 steps of 128 B): 64-byte steps, 32 KiB LDS, 4 workgroups per CU
  return a.n_ktiles <= 8 ? launch_conv<T, BN, 1>(a, stream) : launch_conv<T, BN, 2>(a, stream);
}

// =====================================================================================================
// 256 pixels x 128 couts, 8 waves (64 x 64 each, 64 accumulator VGPRs), 3-slot ring of 24 KiB planes, <= 128 VGPRs:
// TWO workgroups (16 waves, 4 per SIMD) are resident per CU. One workgroup's prologue / epilogue (whose burst of
// output stores is HBM-write bound when every CU reaches it at the same time) overlaps the other's MFMA loop, and four
// waves per SIMD cover each other's LDS-DMA / ds_read latencies without a second fragment register set.
// =====================================================================================================
constexpr int kR2Slots = 3;

template <typename T>
__global__ __launch_bounds__(512, 4) void conv_ring2_kernel(const ConvArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int CH = 16 / ES;
  constexpr int TC = 4, TP = 4;
  constexpr int BM = 256, BN = 128;
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, SLOT = A_PLANE + B_PLANE;   // 24 KiB

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int tile;
  {
    const int nwg = p.n_tiles, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  // grouped launch (the chart predictor's four sub-pixel convolutions, chart.py:45-60): tile = (pixel tile, group, cout tile) - the
  // groups of a pixel tile are neighbours in the numbering, i.e. on one XCD, and read the same input rows
  const int tpg = p.tiles_n * (p.n_groups > 1 ? p.n_groups : 1);
  const int mt = tile / tpg;
  const int grp = (tile - mt * tpg) / p.tiles_n;
  const int nt = tile - mt * tpg - grp * p.tiles_n;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  if (p.n_dev != nullptr && m0 >= *p.n_dev * p.HoWo) return;   // (uniform: a scalar load) nothing live in this tile
  const void* const g_weight = p.n_groups > 1 ? p.weight_g[grp] : p.weight;
  const i32x4* const g_ktab = p.n_groups > 1 ? p.ktab_g[grp] : p.ktab;
  void* const g_out = p.n_groups > 1 ? p.out_g[grp] : p.out;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1;    // cout half (64)
  const int wp = wave >> 1;   // pixel quarter (64)

  const int srow = lane >> 2;
  const int scc = (lane & 3) ^ swz(srow);
  const __attribute__((address_space(4))) i32x4* ktab_c = (const __attribute__((address_space(4))) i32x4*)g_ktab;

  // staging: A plane = 16 pieces of 16 rows (2 per wave: rows 32*wave + 16*i + srow), B plane = 8 pieces (1 per wave)
  int a_boff[2];
  unsigned a_okm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wave * 32 + i * 16 + srow;
    a_okm[i] = 0u;
    a_boff[i] = 0;
    if (m < p.M) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride + p.hi_off, wi0 = wo * p.stride_w + p.wi_off;
      a_boff[i] = (((n * p.H + hi0) * p.W + wi0) * p.Cin + scc * CH) * ES;
      for (int t = 0; t < p.ntaps; ++t) {
        const i32x4 e = ktab_c[t * 4];
        if ((unsigned)(hi0 + e[0]) < (unsigned)p.H && (unsigned)(wi0 + e[1]) < (unsigned)p.W) a_okm[i] |= 1u << t;
      }
    }
  }
  const int w_planes = p.Kpad * ES / 64;
  const int w_boff = (int)dp_wtile_off(n0 + wave * 16 + srow, 0, scc, w_planes);   // tile (n0 / 16 + wave, plane 0) of the tiled weight matrix
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g_weight), 0, p.w_bytes, 0x00020000);
  unsigned char* const lds_sa = smem + wave * 2048;             // this wave's 2 KiB of an A plane
  unsigned char* const lds_sb = smem + A_PLANE + wave * 1024;   // this wave's 1 KiB of a B plane

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int rd_off = fr * 64 + ((fq ^ swz(fr)) << 4);
  const unsigned char* const rd_a = smem + (wp * TP * 16) * 64 + rd_off;
  const unsigned char* const rd_b = smem + A_PLANE + (wc * TC * 16) * 64 + rd_off;

  const int ns = p.n_ktiles * 2;

  // stage plane S into ring slot SLOT_IDX (3 LDS-DMA pieces per wave)
  auto stage = [&](int S, int slot_idx, const i32x4 e) __attribute__((always_inline)) {
    const int tap_boff = ((e[0] * p.W + e[1]) * p.Cin + e[2]) * ES;
    const unsigned tapbit = (e[3] & 1) ? (1u << (e[3] >> 8)) : 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = (a_okm[i] & tapbit) ? (a_boff[i] + tap_boff) : (int)0x80000000;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(lds_sa + slot_idx * SLOT + i * 1024), 16, off, 0, 0, 0);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, DP_LDS_PTR(lds_sb + slot_idx * SLOT), 16, w_boff + S * 1024, 0, 0, 0);
  };

  stage(0, 0, ktab_c[0]);
  if (ns > 1) stage(1, 1, ktab_c[4]);
  i32x4 e_nx = ktab_c[(ns > 2 ? 2 : 0) * 4];
  int slot_cur = 0;       // ring slot of plane s
  int slot_fill = 2;      // ring slot of plane s+2 (= slot of plane s-1)
  for (int s = 0; s < ns; ++s) {
    // plane s landed once only plane s+1's pieces (3 per wave) may still be in flight
    if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everybody's pieces of plane s are visible; everybody finished reading plane s-1
    if (s + 2 < ns) {
      stage(s + 2, slot_fill, e_nx);
      e_nx = ktab_c[min(s + 3, ns - 1) * 4];
    }
    const int so = slot_cur * SLOT;
    u32x4 fp[TP], fc[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) fc[i] = *reinterpret_cast<const u32x4*>(rd_b + so + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < TP; ++j) fp[j] = *reinterpret_cast<const u32x4*>(rd_a + so + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j) Mma<T>::run(fc[i], fp[j], acc[i][j]);
    slot_cur = slot_cur == kR2Slots - 1 ? 0 : slot_cur + 1;
    slot_fill = slot_fill == kR2Slots - 1 ? 0 : slot_fill + 1;
  }

  // ---- epilogue straight from the accumulators ----
  store_tile<T, TP>(p, g_out, acc, m0 + wp * TP * 16, n0 + wc * 64, fr, fq);
}

template <typename T>
int launch_conv_ring2(const ConvArgs& a, hipStream_t stream) {
  constexpr int lds = kR2Slots * (256 + 128) * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring2_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_ring2_kernel<T>), dim3(a.n_tiles), dim3(512), lds, stream, a);
  return dp_check_launch("conv_ring2_kernel");
}

}  // namespace

static int num_cus();

// =====================================================================================================
// Streaming 1x1 kernel for the HBM-bound pointwise layers (res2/res3/res4 conv3 + residual, block-0 shortcuts, the
// decoder predictor): 1 tap, stride 1, plain NHWC in/out, 16-bit storage, Cin * 2 B = KP planes of 64 B (KP = 2, 4, 8),
// Cout a multiple of 256. The generic / ring kernels re-stage the weight tile through LDS for every pixel tile and
// fetch the pixel tile once per cout tile, so the LDS fill traffic of these layers is 3-4x their HBM bytes, and each
// workgroup walks a chain of dependent round trips (stage -> barrier -> MFMA -> residual load -> store). Here
//   * the 256 x K weight slice of the workgroup is staged into LDS ONCE (K-plane images swizzled like the ring planes),
//   * every wave owns whole 32-pixel x 256-cout tiles (persistent, strided over the pixel range): its pixel fragments
//     are plain 16-byte global loads straight into the MFMA B-operand registers - no LDS, no barrier in the loop -,
//   * the registers of the pixel fragments and of the residual are refilled IN PLACE with the wave's next tile right
//     after their last use, so the next tile's loads fly during the current tile's MFMAs, epilogue and stores.
// Wave tile = 16 cout tiles x 2 pixel tiles = 128 accumulator registers; one workgroup (4 waves) per CU.
// =====================================================================================================
// NB = 64-cout blocks per workgroup slice: 4 (256 couts; K up to 8 planes = 128 KiB of weights) or 2 (128 couts; K up to 16
// planes: the 512 -> 128 conv1 layers of res3).
template <typename T, int KP, int NB>
__global__ __launch_bounds__(256, 1) void conv1x1_stream_kernel(const ConvArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only (fp32 parity mode stays on the generic kernels)");
  constexpr int TP = 2;                 // 32 pixels per wave tile
  constexpr int NC = NB * 64;           // couts of the slice
  constexpr int W_PLANE = NC * 64;      // bytes of one K plane of the weight slice
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const bias_s = reinterpret_cast<float*>(smem + KP * W_PLANE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * NC;

  // ---- weights of this cout slice -> LDS, once: piece = (plane kp, 16-row group rg), 1 KiB per wave-instruction
  {
    const int srow = lane >> 2;
    const int scc = (lane & 3) ^ swz(srow);
    const int w_planes = p.Kpad * 2 / 64;      // tiled weight matrix (dp_wtile_off): piece = tile (n0 / 16 + rg, kp)
    const unsigned char* __restrict__ wbase = reinterpret_cast<const unsigned char*>(p.weight) + dp_wtile_off(n0 + srow, 0, scc, w_planes);
    for (int piece = wave; piece < KP * (NC / 16); piece += 4) {
      const int kp = piece / (NC / 16), rg = piece % (NC / 16);
      const unsigned char* src = wbase + ((long long)rg * w_planes + kp) * 1024;
      __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(src), DP_LDS_PTR(smem + kp * W_PLANE + rg * 1024), 16, 0, 0);
    }
    if (tid < NC) bias_s[tid] = p.bias[n0 + tid];
  }
  __syncthreads();  // (vmcnt(0) + barrier)

  const int fr = lane & 15;
  const int fq = lane >> 4;
  const unsigned char* const rd_w = smem + fr * 64 + ((fq ^ swz(fr)) << 4);
  const int n_wt = (p.M + 31) >> 5;                  // 32-pixel wave tiles
  const int wt_step = gridDim.x * 4;
  int wt = blockIdx.x * 4 + wave;
  if (wt >= n_wt) return;

  // Buffer resources sized to the M real pixel rows: a row >= M (tail of the last tile, or the "next tile" of a wave that
  // has none) is out of range, so its loads return zeros without touching memory and its stores are dropped - no
  // clamping or predication, 32-bit offsets with the plane / run displacement in the instruction's immediate.
  const bool has_res = p.residual != nullptr;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const int in_pitch = p.Cin * 2, res_pitch = (int)p.rsW * 2, out_pitch = (int)p.osW * 2;
  // residual: plain NHWC of the output's shape (rshift = 0, res_linear) or the top-down map of the FPN, read through a nearest
  // x2 up-sampling (fpn.py:152: element (n, ho >> 1, wo >> 1)); res_bytes covers exactly its N images, so rows >= M are dropped
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(has_res ? p.residual : p.in), 0, has_res ? p.res_bytes : 0u, 0x00020000);
  const bool res_up = has_res && !p.res_linear;
  auto res_row_off = [&](int m) __attribute__((always_inline)) -> int {   // byte offset of pixel m's residual row
    if (!res_up) return m * res_pitch;
    if (m >= p.M) return 0x7ffffff0;   // past the end: out of range whatever the image count
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    return (int)(n * p.rsN + (ho >> p.rshift) * p.rsH + (wo >> p.rshift) * p.rsW) * 2;
  };
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)((long long)p.M * p.osW * 2), 0x00020000);
  const int col_b = (n0 + fq * 8) * 2;   // byte offset of this lane's first run inside a pixel's channel row

  // Round 6 (profiles/r6_stream_kernel.txt): ONE wave per SIMD has nobody to hide behind. Until round 5 a tile was: per (K plane, cout
  // block) four weight-fragment reads from LDS, a wait, eight MFMAs - an exposed LDS latency per group, 32 groups per tile, about as long as
  // the tile's MFMAs - then the epilogue of all 16 runs on its own. Now
  //   * the loop runs cout block by cout block (K planes inner): block b's accumulators are final after its KP planes, and its epilogue -
  //     bias, residual, ReLU, pack, four stores - rides in the MFMAs' shadow of block b + 1 (of the next tile's block 0 for the last
  //     block): two blocks of accumulators alive instead of all of them;
  //   * the weight fragments of group g + 1 are read while group g multiplies;
  //   * the pixel fragments of the NEXT tile land in a second register set during the whole current tile (they were refilled plane by plane,
  //     which only works when the planes are the outer loop).
  // Every accumulator still sees K planes 0 .. KP - 1 in order: the bits are those of the LDS-ring kernels as before.
  constexpr int UB = TP * 2;            // epilogue units of a cout block: (pixel half j, run pair h) = 8 couts x 16 pixels x 4 lanes each
  constexpr int OOBS = 0x7ffffff0;
  u32x4 aA[KP][TP], aB[KP][TP];         // pixel fragments (MFMA B operand) of the current / the next tile: pixel fr of half j, K chunk fq of plane kp
  u32x4 r[TP][NB * 2];                  // residual: the runs of 8 consecutive couts this lane stores per pixel
  auto load_pixels = [&](u32x4 (&dst)[KP][TP], int wtile, auto kpp) __attribute__((always_inline)) {
    constexpr int kp = decltype(kpp)::value;
#pragma unroll
    for (int j = 0; j < TP; ++j) dst[kp][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (wtile * 32 + j * 16 + fr) * in_pitch + fq * 16 + kp * 64, 0, 0);
  };
  static_for<0, KP>([&](auto kpp) { load_pixels(aA, wt, kpp); });
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int ro = (has_res ? res_row_off(wt * 32 + j * 16 + fr) : 0) + col_b;
#pragma unroll
    for (int q = 0; q < NB * 2; ++q)       // (the last block's runs are loaded by the first tile itself: see epi_unit's refill)
      r[j][q] = (has_res && q < (NB - 1) * 2) ? __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + q * 64, 0, 0) : u32x4{0u, 0u, 0u, 0u};
  }
  auto wfrag = [&](auto gg, u32x4 (&wf)[4]) __attribute__((always_inline)) {
    constexpr int g = decltype(gg)::value;
    constexpr int b = g / KP, kp = g % KP;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const u32x4*>(rd_w + kp * W_PLANE + (b * 64 + i * 16) * 64);
  };
  // epilogue unit (j, h) of cout block b out of `acc`: store at o_off[j] (out of range = nothing stored), then refill the residual registers
  // it has consumed with the values at r_off[j] - the same run of the tile whose epilogue uses them next
  auto epi_unit = [&](f32x4 (&acc)[4][TP], const int (&o_off)[TP], const int (&r_off)[TP], auto bb, auto uu) __attribute__((always_inline)) {
    constexpr int b = decltype(bb)::value, u = decltype(uu)::value;
    constexpr int j = u >> 1, h = u & 1, q = 2 * b + h;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + q * 32 + fq * 8);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + q * 32 + fq * 8 + 4);
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = acc[2 * h][j][k] + b0[k];
      v[4 + k] = acc[2 * h + 1][j][k] + b1[k];
    }
    if (has_res) {
      const u32x4 rv = r[j][q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += Elem<T>::unpack(rv[k] & 0xffffu);
        v[2 * k + 1] += Elem<T>::unpack(rv[k] >> 16);
      }
      r[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, r_off[j] + q * 64, 0, 0);
    }
    if (p.relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    u32x4 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(v[2 * k], v[2 * k + 1]);
    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, o_off[j] + q * 64, 0, 0);
  };
  f32x4 accP[2][4][TP];                 // the block being accumulated and the block whose epilogue is pending
  int o_prev[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) o_prev[j] = OOBS;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) accP[0][i][j] = accP[1][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one tile; PAR = its parity: which pixel set is current, and (for an odd block count) which accumulator set block 0 takes
  auto tile = [&](auto par_) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_)::value;
    u32x4 (&acur)[KP][TP] = PAR ? aB : aA;
    u32x4 (&anxt)[KP][TP] = PAR ? aA : aB;
    int o_cur[TP], r_cur[TP], r_nxt[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      const int m_cur = wt * 32 + j * 16 + fr, m_nxt = m_cur + wt_step * 32;      // (past the end: out of range, see above)
      o_cur[j] = m_cur < p.M ? m_cur * out_pitch + col_b : OOBS;
      r_cur[j] = (has_res ? res_row_off(m_cur) : 0) + col_b;
      r_nxt[j] = (has_res ? res_row_off(m_nxt) : 0) + col_b;
    }
    u32x4 wf[2][4];
    wfrag(std::integral_constant<int, 0>{}, wf[0]);
    static_for<0, NB * KP>([&](auto gg) {
      constexpr int g = decltype(gg)::value;
      constexpr int b = g / KP, kp = g % KP;
      constexpr int S = (b + PAR * (NB & 1)) & 1;          // accumulator set of this block; the pending block's is 1 - S
      if constexpr (g + 1 < NB * KP) wfrag(std::integral_constant<int, (g + 1 < NB * KP ? g + 1 : 0)>{}, wf[(g + 1) & 1]);
      if constexpr (b == 0) load_pixels(anxt, wt + wt_step, std::integral_constant<int, kp>{});      // the next tile's pixels, a plane per group
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kp == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TP; ++j) accP[S][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) Mma<T>::run(wf[g & 1][i], acur[kp][j], accP[S][i][j]);
      __builtin_amdgcn_sched_barrier(0);
      // the pending block's epilogue units of this group: block b - 1 of this tile, or the last block of the previous tile
      static_for<(kp * UB) / KP, ((kp + 1) * UB) / KP>([&](auto uu) {
        if constexpr (b == 0) epi_unit(accP[1 - S], o_prev, r_cur, std::integral_constant<int, NB - 1>{}, uu);
        else epi_unit(accP[1 - S], o_cur, r_nxt, std::integral_constant<int, (b > 0 ? b - 1 : 0)>{}, uu);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int j = 0; j < TP; ++j) o_prev[j] = o_cur[j];
  };
  bool odd = false;        // parity of the last tile run
  for (;;) {
    tile(std::integral_constant<int, 0>{});
    wt += wt_step;
    odd = false;
    if (wt >= n_wt) break;
    tile(std::integral_constant<int, 1>{});
    wt += wt_step;
    odd = true;
    if (wt >= n_wt) break;
  }
  // drain: the last tile's last block (nothing left to refill)
  {
    int r_none[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) r_none[j] = OOBS;
    constexpr int SL0 = (NB - 1) & 1, SL1 = (NB - 1 + (NB & 1)) & 1;
    if (!odd) static_for<0, UB>([&](auto uu) { epi_unit(accP[SL0], o_prev, r_none, std::integral_constant<int, NB - 1>{}, uu); });
    else static_for<0, UB>([&](auto uu) { epi_unit(accP[SL1], o_prev, r_none, std::integral_constant<int, NB - 1>{}, uu); });
  }
}

template <typename T, int KP, int NB = 4>
int launch_conv_stream(const ConvArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) != 2) {
    return dp_fail(DP_ERR_UNSUPPORTED, "conv1x1_stream_kernel: 16-bit storage only");
  } else {
    constexpr int NC = NB * 64;
    constexpr int lds = KP * NC * 64 + NC * 4;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_stream_kernel<T, KP, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set = true;
    }
    const int slices = a.Cout / NC;
    const int n_wt = (a.M + 31) / 32;
    int gx = (num_cus() + slices - 1) / slices;   // one workgroup per CU over all slices
    if (gx > (n_wt + 3) / 4) gx = (n_wt + 3) / 4;
    hipLaunchKernelGGL((conv1x1_stream_kernel<T, KP, NB>), dim3(gx, slices), dim3(256), lds, stream, a);
    return dp_check_launch("conv1x1_stream_kernel");
  }
}

// Kernel choice (measured on MI355X, profiles/): the LDS-ring kernels need a 64-byte K plane to lie inside one tap;
// the 256x256 ring tile is ~1.15x the 128x128 ring tile when both fill the chip, so the shape is picked by
// wave-quantisation efficiency (workgroups / (CUs x resident workgroups per CU), rounded up to whole rounds);
// short-K layers are HBM/latency bound and run on the generic kernel with 64-byte steps (4 workgroups per CU).
enum { DP_CONV_K64 = 0, DP_CONV_K128 = 1, DP_CONV_RING256 = 2, DP_CONV_RING128 = 3, DP_CONV_RING256x128 = 4, DP_CONV_STREAM = 5, DP_CONV_WSR = 6, DP_CONV_ROWS = 7, DP_CONV_ROWS2 = 8, DP_CONV_PWS = 9, DP_CONV_WSQ = 10 };

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    else
      n = 256;
  }
  return n;
}

static int choose_conv_kernel(const dp_conv_params* p, long long M) {
  const int es = p->dtype == DP_F32 ? 4 : 2;
  // ring kernels: a 64-byte plane inside one tap, <= 32 taps (per-row tap validity is a bit mask), tensors addressable
  // through 32-bit buffer offsets
  const bool ring_ok = (p->Cin * es) % 64 == 0 && p->ntaps >= 1 && p->ntaps <= 32 &&
                       (long long)p->N * p->H * p->W * p->Cin * es < (1ll << 31) && (long long)p->Cout_w * p->Kpad * es < (1ll << 31);
  const bool big_ok = ring_ok && p->Cout % 256 == 0 && p->Cout_w % 256 == 0;
  // streaming 1x1: pointwise, stride 1, plain NHWC tensors, 16-bit storage, K = 2 / 4 / 8 planes, Cout in 256-slices
  const int kb = p->Cin * es;
  const bool lin_out = p->osH == (long long)p->Wo * p->osW && p->osN == (long long)p->Ho * p->osH;
  const bool lin_res = !p->residual || (p->rshift == 0 && p->rsH == (long long)p->Wo * p->rsW && p->rsN == (long long)p->Ho * p->rsH);
  // ... or the FPN top-down map read through a nearest x2 up-sampling (rshift 1, a plain NHWC tensor of half the size)
  const bool up_res = p->residual && p->rshift == 1 && p->Ho % 2 == 0 && p->Wo % 2 == 0 && p->rsH == (long long)(p->Wo / 2) * p->rsW &&
                      p->rsN == (long long)(p->Ho / 2) * p->rsH && (long long)(p->N + 2) * p->rsN * 2 < (1ll << 31);
  // slice shapes: 256 couts with K = 2 / 4 / 8 planes of 64 B, or exactly 128 couts with K = 16 planes (the 512 -> 128 conv1 of res3)
  // ... or exactly 64 couts with K = 2 planes (res2.0 conv1 on the pooled stem output, resnet.py:192-193: 138 MB in / out at batch 8,
  // 40 us at 110 TFLOP/s on the generic kernel in round 3)
  const bool stream_shape = ((kb == 128 || kb == 256 || kb == 512) && p->Cout % 256 == 0 && p->Cout_w % 256 == 0) ||
                            (kb == 1024 && p->Cout == 128 && p->Cout_w == 128) || (kb == 128 && p->Cout == 64 && p->Cout_w == 128);
  const bool stream_ok = es == 2 && !p->n_dev && p->ntaps == 1 && p->stride == 1 && (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == 0 && p->wi_off == 0 && p->H == p->Ho && p->W == p->Wo &&
                         stream_shape && p->Kpad == p->Cin &&
                         !p->out_f32 && lin_out && (lin_res || up_res) && M >= 4096 &&
                         // 32-bit buffer offsets, rows up to one grid stride of tiles past the end are addressed
                         (M + (1ll << 16)) * 2 * (p->Cin > p->osW ? p->Cin : (p->osW > p->rsW ? p->osW : p->rsW)) < (1ll << 31);
  const DpPolicy& pol = dp_policy();
  if (p->Cout <= 64) {
    if (stream_ok && !p->in2 && pol.conv_big < 0 && pol.conv_stream != 0) return DP_CONV_STREAM;
    return DP_CONV_K64;
  }
  if (p->in2) {   // second source (K-concatenated pointwise layer): the LDS-ring kernels implement it (conv_big = 2: the 128x128 one)
    if (!ring_ok) return -1;
    if (big_ok && ((M + 127) / 128) * (p->Cout / 256) >= 128 && pol.conv_big != 2) return DP_CONV_RING256;
    return DP_CONV_RING128;
  }
  if (pol.conv_big >= 0) {   // test / calibration override - 0: generic only, 1: 256x256 ring whenever legal, 2: 128x128 ring whenever legal, 3: 256x128 ring, 5: streaming 1x1 whenever legal
    const int f = (int)pol.conv_big;
    if (f == 5 && stream_ok) return DP_CONV_STREAM;
    if (f == 1 && big_ok) return DP_CONV_RING256;
    if (f == 2 && ring_ok) return DP_CONV_RING128;
    if (f == 3 && ring_ok && p->Cout > 64) return DP_CONV_RING256x128;
    return DP_CONV_K128;
  }
  if (dp_conv_rows2_ok(p)) return DP_CONV_ROWS2;   // 3x3 with 512 input channels on maps whose width suits 32-pixel strips (DensePose head): dp_conv_rows.hip, third form
  if (dp_conv_rows_ok(p)) return DP_CONV_ROWS;     // 3x3 with 512 input channels (DensePose head, res5): row-streaming K-split weight-stationary kernel (dp_conv_rows.hip)
  if (dp_conv_wsq_ok(p)) return DP_CONV_WSQ;       // 3x3 256 -> 256: weights stationary in registers, one wave per SIMD on v_mfma_f32_32x32x16 (dp_conv_wq.hip)
  if (dp_conv_wsr_ok(p)) return DP_CONV_WSR;       // 3x3 C -> C, C = 128 / 256: weights stationary in registers (dp_conv_ws.hip)
  if (dp_conv_pws_ok(p)) return DP_CONV_PWS;       // 1x1 with K = 512 / 1024 / 2048: weights stationary in registers, pixels once through LDS (dp_conv_pw.hip)
  if (stream_ok && pol.conv_stream != 0) return DP_CONV_STREAM;
  if (!ring_ok) return DP_CONV_K128;
  if ((long long)p->Kpad * es < 1024) return DP_CONV_K128;   // short K: see above
  // Calibrated on the real layer shapes (tools/conv_sweep.py, MI355X): the 256x256 ring tile wins whenever it has at
  // least ~half a chip of tiles, even with a ragged last round (fewer resident workgroups run faster), except when the last
  // round is almost empty (p3-level 3x3: 525 tiles), where the 256x128 two-workgroup tile is ~7 % faster; small-M layers
  // (res5, p5, fully-connected: M <= 8400) need the 128x128 tile to occupy the chip at all.
#ifdef DP_EXPERIMENTS
  if (pol.conv_policy == 0) {  // round-1 "a" policy: wave-quantisation estimate only
    const double cus = (double)num_cus();
    const double ts = (double)((M + 127) / 128) * ((p->Cout + 127) / 128);
    const double rs = ts / (2.0 * cus);
    const double eff_s = rs / (double)(long long)(rs + 0.999999);
    if (big_ok) {
      const double tb = (double)((M + 255) / 256) * (p->Cout / 256);
      const double rb = tb / cus;
      const double eff_b = rb / (double)(long long)(rb + 0.999999);
      if (1.15 * eff_b > eff_s) return DP_CONV_RING256;
    }
    return DP_CONV_RING128;
  }
  if (big_ok && pol.conv_policy == 1) {   // round-1 interim policy
    const long long t256 = ((M + 255) / 256) * (p->Cout / 256);
    if (t256 >= 96 && (long long)p->Kpad * es >= 256 * 64) return DP_CONV_RING256;
    if (t256 >= 132) {
      const long long rem = t256 % num_cus();
      if (t256 < 4ll * num_cus() && rem >= 1 && rem <= num_cus() / 6) return DP_CONV_RING256x128;
      return DP_CONV_RING256;
    }
    return DP_CONV_RING128;
  }
#endif
  if (big_ok) {
    // the 256-cout kernel picks its own tile height (choose_ring256_tp); with 128-row tiles it beats the 128x128 kernel as
    // soon as it has about half a chip of them (res5 convs, fc2: 5-15 % faster at 132-252 tiles; p5-level 3x3 with 66: slower)
    const long long t128 = ((M + 127) / 128) * (p->Cout / 256);
    if (t128 >= 128) return DP_CONV_RING256;
  }
  // Cout not a multiple of 256 on MANY pixels (the chart predictor's sub-pixel convolutions on hundreds of ROIs): the 256x128 tile with
  // two workgroups per CU puts four waves on a SIMD behind the same per-plane latency and wins 6 - 8 % from ~230 k pixels on
  // (512 -> 80 on N x 28 x 28: N = 300 / 400 / 600 / 800: 279 / 365 / 534 / 710 us against 304 / 392 / 569 / 767; N = 64: 102 against 76).
  // Same K order as every ring tile: the choice may depend on the pixel count. conv_ring2_m: the line (0 = never).
  if (pol.conv_ring2_m > 0 && M >= pol.conv_ring2_m && p->Cout > 64) return DP_CONV_RING256x128;
  return DP_CONV_RING128;
}

// Tile height of the 256-cout ring kernel: 32 * TP pixels, TP = 4 .. 8 (template instances). A launch runs in rounds of
// one workgroup per CU. Measured on MI355X (tools/conv_micro.py with DP_CONV_TP) a tile costs about
// TP + 1.5 units (TP pixel tiles of MFMA work + a part that does not shrink: weight staging, barriers, prologue) and a
// partly filled last round costs about half of what the missing tiles would - so
//     time ~ (ceil(r) + r) / 2 * (TP + 1.5),   r = tiles(TP) / CUs.
// That one expression reproduces the measured ranking of every shape class of the model: many-round launches keep 256 rows
// (200x336-level 3x3: 0.672 ms vs 0.682 / 0.694 / 0.733 / 0.751 at 224 / 192 / 160 / 128 rows), launches that would leave
// CUs idle shrink their tiles (50x84-level 3x3: 132 tiles -> 210 of 160 rows, 62 -> 50 us; fc1: 274 -> 237 us), and a launch
// whose 256-row tiling ends in an almost empty round (100x168-level 3x3: 525 tiles = 2.05 rounds) takes 192 rows
// (0.206 ms on the 256x128 two-workgroup kernel -> 0.184 ms). The per-pixel arithmetic does not depend on the tile a pixel
// lands in, so results stay bit-identical across batch sizes.
static int choose_ring256_tp(const dp_conv_params* p, long long M) {
  const int f = (int)dp_policy().conv_tp;   // test / calibration override
  if (f >= 4 && f <= 8) return f;
  const long long tn = p->Cout / 256;
  const double cus = (double)num_cus();
  int best = 8;
  double best_cost = 1e30;
  for (int tp = 8; tp >= 4; --tp) {   // ties go to the taller tile
    const double r = (double)(((M + 32 * tp - 1) / (32 * tp)) * tn) / cus;
    const double cost = 0.5 * ((double)(long long)(r + 0.999999) + r) * (tp + 1.5);
    if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = tp; }
  }
  return best;
}

// dp_conv_params.split_k is honoured when the split instances of the LDS-ring kernels take the layer; otherwise the layer runs unsplit
// on whatever kernel it would get without the field (round 3 returned DP_ERR_UNSUPPORTED here: a batch whose fc1 input passes 2 GiB,
// 86 frames, turned from "generic kernel" into a hard error)
static bool split_legal(const dp_conv_params* p) {
  const int es = p->dtype == DP_F32 ? 4 : 2;
  const bool ring_ok = (p->Cin * es) % 64 == 0 && p->ntaps >= 1 && p->ntaps <= 32 &&
                       (long long)p->N * p->H * p->W * p->Cin * es < (1ll << 31) && (long long)p->Cout_w * p->Kpad * es < (1ll << 31);
  const bool lin_out = p->osH == (long long)p->Wo * p->osW && p->osN == (long long)p->Ho * p->osH;
  return p->split_k > 1 && es == 2 && ring_ok && p->split_ws && !p->residual && !p->head_out && !p->in2 && !p->post_res && !p->out_f32 &&
         !p->n_dev && lin_out && p->out && p->Cout % 128 == 0 && p->Kpad * es / 64 >= 2 * p->split_k;
}

static bool split_uses_256(const dp_conv_params* p, long long M) {
  const int es = p->dtype == DP_F32 ? 4 : 2;
  const int planes = p->Kpad * es / 64;
  const int per = (planes + p->split_k - 1) / p->split_k;
  const int n_seg = (planes + per - 1) / per;
  return p->Cout % 256 == 0 && p->Cout_w % 256 == 0 && ((M + 255) / 256) * (p->Cout / 256) * n_seg >= (3 * num_cus()) / 5;
}

extern "C" int dp_conv2d_kernel_class(const dp_conv_params* p) {
  if (!p) return -1;
  const long long M = (long long)p->N * p->Ho * p->Wo;
  if (split_legal(p)) return split_uses_256(p, M) ? DP_CONV_RING256 : DP_CONV_RING128;
  return choose_conv_kernel(p, M);
}

extern "C" int dp_conv2d_tile_rows(const dp_conv_params* p) {
  if (!p) return -1;
  const long long M = (long long)p->N * p->Ho * p->Wo;
  if (split_legal(p)) return split_uses_256(p, M) ? 256 : 128;
  switch (choose_conv_kernel(p, M)) {
    case DP_CONV_RING256: return 32 * choose_ring256_tp(p, M);
    case DP_CONV_RING256x128: return 256;
    case DP_CONV_STREAM: return 32;
    case DP_CONV_WSR: return 16;
    case DP_CONV_WSQ: return 16;
    case DP_CONV_ROWS: return 16;
    case DP_CONV_ROWS2: return 32;
    case DP_CONV_PWS: return p->Cin == 2048 ? 16 : 32;
    default: return 128;
  }
}

// evaluates EXPR with the storage type bound to T and returns its value
#define DP_BY_DTYPE(EXPR)                                  \
  switch (p->dtype) {                                      \
    case DP_F32: { using T = float; return EXPR; }         \
    case DP_BF16: { using T = uint16_t; return EXPR; }     \
    default: { using T = f16_t; return EXPR; }             \
  }

extern "C" int dp_conv2d_nhwc(const dp_conv_params* p, dp_stream_t stream) {
  DP_REQUIRE(p != nullptr, "dp_conv2d_nhwc: null params");
  DP_REQUIRE(p->dtype == DP_F32 || p->dtype == DP_BF16 || p->dtype == DP_F16, "dp_conv2d_nhwc: bad dtype %d", p->dtype);
  const int es = p->dtype == DP_F32 ? 4 : 2;
  DP_REQUIRE(p->N >= 0 && p->H > 0 && p->W > 0 && p->Ho > 0 && p->Wo > 0, "dp_conv2d_nhwc: bad spatial shape");
  const long long M = (long long)p->N * p->Ho * p->Wo;
  if (M == 0) return DP_OK;  // R = 0 detections is legal
  DP_REQUIRE(p->in && p->bias && ((p->weight && p->ktab && (p->out || p->head_out)) || p->n_groups > 1), "dp_conv2d_nhwc: null pointer");
  DP_REQUIRE(p->Cin > 0 && p->Cin % 8 == 0, "dp_conv2d_nhwc: Cin=%d must be a positive multiple of 8", p->Cin);
  DP_REQUIRE(p->Cout > 0 && p->Cout % 8 == 0 && p->Cout <= p->Cout_w, "dp_conv2d_nhwc: Cout=%d Cout_w=%d", p->Cout, p->Cout_w);
  DP_REQUIRE(p->Cout_w % 128 == 0, "dp_conv2d_nhwc: Cout_w=%d must be a multiple of 128", p->Cout_w);
  DP_REQUIRE(p->Kpad > 0 && (p->Kpad * es) % kKB == 0, "dp_conv2d_nhwc: Kpad=%d not a multiple of %d bytes", p->Kpad, kKB);
  DP_REQUIRE(p->stride >= 1 && p->stride_w >= 0, "dp_conv2d_nhwc: stride");
  DP_REQUIRE(M < (1ll << 31) && (long long)p->N * p->H * p->W * p->Cin < (1ll << 31), "dp_conv2d_nhwc: tensor too large for 32-bit pixel index");
  ConvArgs a;
  a.in = p->in; a.weight = p->weight; a.ktab = reinterpret_cast<const i32x4*>(p->ktab); a.bias = p->bias;
  a.residual = p->residual; a.out = p->out;
  a.N = p->N; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout; a.Kpad = p->Kpad;
  a.stride = p->stride; a.stride_w = p->stride_w > 0 ? p->stride_w : p->stride; a.hi_off = p->hi_off; a.wi_off = p->wi_off; a.relu = p->relu; a.rshift = p->rshift; a.out_f32 = p->out_f32;
  a.osN = p->osN; a.osH = p->osH; a.osW = p->osW; a.rsN = p->rsN; a.rsH = p->rsH; a.rsW = p->rsW;
  a.M = (int)M; a.HoWo = p->Ho * p->Wo; a.n_ktiles = p->Kpad * es / kKB;
  a.in_bytes = (unsigned)((long long)p->N * p->H * p->W * p->Cin * es);
  a.w_bytes = (unsigned)((long long)p->Cout_w * p->Kpad * es);
  a.res_bytes = p->residual ? (unsigned)((long long)p->N * p->rsN * es) : 0u;
  a.ntaps = p->ntaps;
  a.head_w = p->head_w; a.head_b = p->head_b; a.head_out = p->head_out;
  a.n_dev = p->n_dev;
  a.in2 = p->in2; a.H2 = p->H2; a.W2 = p->W2; a.Cin2 = p->Cin2; a.stride2 = p->stride2;
  a.in2_bytes = p->in2 ? (unsigned)((long long)p->N * p->H2 * p->W2 * p->Cin2 * es) : 0u;
  a.split_k = 0; a.split_ws = nullptr;
  a.n_groups = 0;
  for (int g = 0; g < 4; ++g) { a.weight_g[g] = nullptr; a.ktab_g[g] = nullptr; a.out_g[g] = nullptr; }
  if (p->n_groups > 1) {
    DP_REQUIRE(p->n_groups <= 4, "dp_conv2d_nhwc: n_groups=%d (at most 4)", p->n_groups);
    for (int g = 0; g < p->n_groups; ++g)
      DP_REQUIRE(p->weight_g[g] && p->ktab_g[g] && p->out_g[g], "dp_conv2d_nhwc: group %d of %d: null weight / ktab / out", g, p->n_groups);
    DP_REQUIRE(!p->in2 && !p->head_out && !p->post_res && p->post_mode == 0 && p->split_k <= 1 && !p->residual,
               "dp_conv2d_nhwc: a grouped launch takes no second source / fused head / post_res / split_k / residual");
  }
  if (p->in2) {
    DP_REQUIRE(p->ntaps == 1 && p->stride == 1 && p->hi_off == 0 && p->wi_off == 0 && p->H == p->Ho && p->W == p->Wo,
               "dp_conv2d_nhwc: a second source needs a pointwise stride-1 layer");
    DP_REQUIRE(p->Cin2 > 0 && (p->Cin * es) % 64 == 0 && (p->Cin2 * es) % 64 == 0 && p->Kpad == p->Cin + p->Cin2,
               "dp_conv2d_nhwc: second source: Cin=%d Cin2=%d Kpad=%d (both 64-byte multiples, Kpad = their sum)", p->Cin, p->Cin2, p->Kpad);
    DP_REQUIRE(p->stride2 >= 1 && p->H2 >= (p->Ho - 1) * p->stride2 + 1 && p->W2 >= (p->Wo - 1) * p->stride2 + 1 &&
               (long long)p->N * p->H2 * p->W2 * p->Cin2 * es < (1ll << 31), "dp_conv2d_nhwc: second source geometry");
    DP_REQUIRE(!p->head_out, "dp_conv2d_nhwc: second source and fused head are exclusive");
  }
  a.out_linear = (p->osH == (long long)p->Wo * p->osW && p->osN == (long long)p->Ho * p->osH) ? 1 : 0;
  a.res_linear = (p->residual && p->rshift == 0 && p->rsH == (long long)p->Wo * p->rsW && p->rsN == (long long)p->Ho * p->rsH) ? 1 : 0;
  hipStream_t s = as_stream(stream);
  if (split_legal(p)) {
    // Split-K (long-K layers with few pixel tiles: the box head's fc1, res5's 1x1 conv1, fpn_lateral5): the K planes are cut into split_k
    // segments, each (tile, segment) workgroup writes fp32 partial sums, one pass adds them in segment order (+ bias, activation). The
    // segment count belongs to the LAYER - the caller passes the same value whatever the batch - so a pixel's summation order is
    // fixed; what depends on the launch is only the tile shape: 256 x 256 once that gives about a chip of workgroups, else 128 x 128.
    const int planes = p->Kpad * es / 64;
    const int per = (planes + p->split_k - 1) / p->split_k;
    const int n_seg = (planes + per - 1) / per;      // every segment non-empty
    a.split_k = p->split_k; a.split_ws = reinterpret_cast<float*>(p->split_ws);
    const long long t256 = ((M + 255) / 256) * (p->Cout / 256);
    if (split_uses_256(p, M)) {
      a.tiles_n = p->Cout / 256;
      a.n_tiles = (int)t256;
      if (p->dtype == DP_BF16) return launch_conv_ring_split<uint16_t, 4, 8>(a, n_seg, s);
      return launch_conv_ring_split<f16_t, 4, 8>(a, n_seg, s);
    }
    a.tiles_n = (p->Cout + 127) / 128;
    a.n_tiles = (int)((M + 127) / 128) * a.tiles_n;
    if (p->dtype == DP_BF16) return launch_conv_ring_split<uint16_t, 2, 4>(a, n_seg, s);
    return launch_conv_ring_split<f16_t, 2, 4>(a, n_seg, s);
  }
  const int kc = choose_conv_kernel(p, M);
  if (kc < 0) return dp_fail(DP_ERR_UNSUPPORTED, "dp_conv2d_nhwc: a second source needs a shape the LDS-ring kernels take");
  if (p->n_groups > 1) {
    if (kc != DP_CONV_RING128 && kc != DP_CONV_RING256x128)
      return dp_fail(DP_ERR_UNSUPPORTED, "dp_conv2d_nhwc: a grouped launch runs on the 128-cout LDS-ring kernels (kernel classes 3 / 4) only; this layer is class %d", kc);
    a.n_groups = p->n_groups;
    for (int g = 0; g < p->n_groups; ++g) { a.weight_g[g] = p->weight_g[g]; a.ktab_g[g] = reinterpret_cast<const i32x4*>(p->ktab_g[g]); a.out_g[g] = p->out_g[g]; }
  }
  if (p->head_out) {
    // fused 1x1 head: the 256-cout ring kernel only (all channels of a pixel in one workgroup), 16-bit storage, ReLU hidden layer
    DP_REQUIRE(p->head_w && p->head_b, "dp_conv2d_nhwc: head_out given without head_w / head_b");
    if (!(kc == DP_CONV_RING256 && p->Cout == 256 && es == 2 && p->relu && !p->residual && !p->out_f32))
      return dp_fail(DP_ERR_UNSUPPORTED, "dp_conv2d_nhwc: the fused head needs the 256-cout ring kernel (Cout 256, 16-bit storage, ReLU, no residual)");
  }
  if (kc == DP_CONV_ROWS || kc == DP_CONV_ROWS2) return dp_conv_rows_launch(p, stream);
  if (kc == DP_CONV_WSR) return dp_conv_wsr_launch(p, stream);
  if (kc == DP_CONV_WSQ) return dp_conv_wsq_launch(p, stream);
  if (kc == DP_CONV_PWS) return dp_conv_pws_launch(p, stream);
  if (p->post_res || p->post_mode)
    return dp_fail(DP_ERR_UNSUPPORTED, "dp_conv2d_nhwc: post_res (a tensor added after the activation) is implemented by the weight-stationary 3x3 kernel only "
                                        "(256 -> 256 channels, ReLU, 16-bit storage; post_mode 2 needs even H and W)");
  if (kc == DP_CONV_STREAM) {
    a.tiles_n = p->Cout / 256;
    a.n_tiles = 0;
    const int kp = p->Cin * es / 64;
    if (kp == 16) { DP_BY_DTYPE((launch_conv_stream<T, 16, 2>(a, s))); }
    if (p->Cout == 64) { DP_BY_DTYPE((launch_conv_stream<T, 2, 1>(a, s))); }
    if (kp == 2) { DP_BY_DTYPE((launch_conv_stream<T, 2>(a, s))); }
    if (kp == 4) { DP_BY_DTYPE((launch_conv_stream<T, 4>(a, s))); }
    DP_BY_DTYPE((launch_conv_stream<T, 8>(a, s)));
  }
  if (kc == DP_CONV_RING256) {
    const int tp = choose_ring256_tp(p, M);
    a.tiles_n = p->Cout / 256;
    a.n_tiles = (int)((M + 32 * tp - 1) / (32 * tp)) * a.tiles_n;
    switch (tp) {
      case 4: { DP_BY_DTYPE((launch_conv_ring<T, 4, 4>(a, s))); }
      case 5: { DP_BY_DTYPE((launch_conv_ring<T, 4, 5>(a, s))); }
      case 6: { DP_BY_DTYPE((launch_conv_ring<T, 4, 6>(a, s))); }
      case 7: { DP_BY_DTYPE((launch_conv_ring<T, 4, 7>(a, s))); }
      default: { DP_BY_DTYPE((launch_conv_ring<T, 4, 8>(a, s))); }
    }
  }
  const int n_grp = p->n_groups > 1 ? p->n_groups : 1;
  if (kc == DP_CONV_RING256x128) {
    a.tiles_n = (p->Cout + 127) / 128;
    a.n_tiles = (int)((M + 255) / 256) * a.tiles_n * n_grp;
    DP_BY_DTYPE((launch_conv_ring2<T>(a, s)));
  }
  if (kc == DP_CONV_RING128) {
    a.tiles_n = (p->Cout + 127) / 128;
    a.n_tiles = (int)((M + 127) / 128) * a.tiles_n * n_grp;
    DP_BY_DTYPE((launch_conv_ring<T, 2, 4>(a, s)));
  }
  const int tiles_m = (int)((M + kBM - 1) / kBM);
  if (kc == DP_CONV_K64) {
    a.tiles_n = (p->Cout + 63) / 64;
    a.n_tiles = tiles_m * a.tiles_n;
    DP_BY_DTYPE((launch_conv_k<T, 64>(a, s)));
  }
  a.tiles_n = (p->Cout + 127) / 128;
  a.n_tiles = tiles_m * a.tiles_n;
  DP_BY_DTYPE((launch_conv_k<T, 128>(a, s)));
}
