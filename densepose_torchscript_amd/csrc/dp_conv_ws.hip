// Weight-stationary 3x3 convolutions (16-bit storage): the weights of a cout slice live in the REGISTER FILE of a CU for the
// whole launch, input rows stream through an LDS ring. Behind dp_conv2d_nhwc (dp_conv.hip, kernel class 6).
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"
#include <stdlib.h>

#ifndef DP_EXP
#define DP_EXP 0      // diagnostic builds: 1 no fragment reads, 4 no MFMAs, 8 no row fetches, 16 in-kernel phase stamps
#endif
#ifndef DP_WSR_LATE
#define DP_WSR_LATE 1 // two-part form: the storing waves run their epilogue right after their matrix loop and read the hand-over first thing (0: round-4 order)
#endif

namespace {
int ws_num_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}
}  // namespace

// =====================================================================================================
// C -> C channels, 3x3, pad 1, stride 1, for C = 128 (res3 conv2) and C = 256 (res4 conv2, the FPN output convolutions and the
// decoder's 3x3 convolutions; /root/reference/detectron2/modeling/backbone/resnet.py:195-197, fpn.py:134-135,
// densepose/modeling/roi_heads/roi_head.py decoder scale heads).
//
// On the LDS-ring kernels these layers re-stage 128 - 256 weight rows and their pixel rows per 64-byte K plane, wait, synchronise:
// 36 - 72 plane steps whose cost is latency, not arithmetic. But a 16-cout x 1152-K slice of the weights is
// 144 VGPRs of a wave, so here nothing about the weights ever moves after the prologue.
//
// Every wave keeps 16 output channels x 1152 K values (36 K steps, 144 VGPRs) for the whole launch. With C = 128 that is the
// whole K axis (8 waves = 128 couts per workgroup); with C = 256 two waves share a cout group: the part-0 wave contracts channel
// blocks 0..3 and hands its accumulators (through LDS) to the part-1 wave, which continues with blocks 4..7 - the K order of the
// packed matrix, so the result is bit-identical to the ring kernels. The chain is pipelined: in iteration i part h works on step
// i - h. A workgroup then covers 64 couts and the grid is (cout slice) x (pixel group); the slices of a pixel group share an XCD.
//   * work = steps (RP output rows of a 16-pixel column strip), numbered strip by strip, top to bottom; a workgroup owns a
//     contiguous range and NEVER drains between columns: ring rows are addressed by a virtual row number that advances by RP
//     per step and by 2 more at a column start (the new column's first two rows are fresh); the bookkeeping is incremental
//     (no division or modulo in the loop: the first version spent 338 scalar instructions per wave and step on them);
//   * input rows travel global -> LDS by LDS-DMA (whole lines, one step ahead, no registers, no ds_write). The ring stores a
//     pixel with a 32-byte pad (pitch 2C + 32): chunk c of pixel p lands on 16-byte slot (2p + c) mod 16, fragment reads at
//     every column tap are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0) AND the address is affine, so column tap and channel
//     block are instruction offsets: one ds_read_b128 per fragment, no address arithmetic;
//   * the weight rows are read un-permuted (A row r of wave group g = logical cout 16 g + r), so the four lanes of a pixel
//     hold 16 consecutive output channels and the epilogue stores 8 bytes per lane straight from the accumulators: no output
//     staging buffer, ONE barrier per step;
//   * waves w and w + 4 share a SIMD and the older one wins the matrix pipe whenever both are ready. Everything that is not a
//     fragment read or an MFMA sits at the TOP of an iteration, before the wave has LDS reads in flight (an LDS-DMA issued behind
//     outstanding ds_reads of its wave waits for them): row fetches, the hand-over read, and - with one K part - the PREVIOUS step's
//     stores (the epilogue is deferred by one iteration). With two K parts the roles follow the age of the waves: older = part 1
//     (hand-over read first thing, a loop of nothing but reads and MFMAs, then ITS step's epilogue and stores in the ~1800 cycles it
//     would otherwise wait at the barrier - round 5), younger = part 0 (all row fetches, MFMAs, hand-over write).
// In-kernel phase stamps (-DDP_EXP=16) on the 200x336 level (profiles/r5_wsr_experiments.txt): 3456 MFMA cycles per SIMD and step out of
// ~5200. The matrix loops themselves run at 17 (older wave) and 22 (younger, mostly alone) cycles per MFMA; what is lost is the ~1000
// cycles after the barrier in which neither wave of a SIMD multiplies yet (step bookkeeping, the younger wave's row fetches).
// A persistent launch with a static split over exactly as many workgroups as CUs cannot rebalance when it shares the chip with
// another stream: the host says so (dp_conv_params.shared_chip) and those launches are split over twice as many workgroups
// (DESIGN.md section 4.1c).
// =====================================================================================================
namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WsrArgs {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  int N, H, W, relu, kpad, cout;
  int n_strips, spc, n_slices, n_pg;
  int S;                       // steps per cout slice
  int over;                    // workgroups per CU slot (1 = one persistent workgroup per CU)
  unsigned in_bytes, out_bytes;
  const void* post;            // tensor added after the activation (dp_conv_params.post_res): POST = 1 output geometry, 2 = half size, bilinear x2
  int Hl, Wl;                  // POST = 2: geometry of the half-size map
  unsigned post_bytes;
  unsigned long long* dbg;     // diagnostic builds (-DDP_EXP=16): per-wave phase cycle sums
};

template <int N>
__device__ __forceinline__ void wsr_wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else { static_assert(N == 8, "vmcnt immediate"); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
}

// step bookkeeping without divisions: a step's image, first column, first output row and the ring row (virtual row modulo the
// ring size) of its input row r - 1; `first` = none of its input rows is in the ring yet
struct WsrStep {
  int n, c0, r, um, first;
};

// ATen's bilinear x2 source index / weight (align_corners = False): src = max((o + 0.5) * 0.5 - 0.5, 0) - the formula of dp_ops.hip
__device__ __forceinline__ void wsr_bil_src(int o, int n, int& i0, int& i1, float& l) {
  float s = ((float)o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}

// POST: 0 = plain, 1 = out = act(..) + post[same pixel], 2 = out = act(..) + bilinear_x2(post) (the decoder's level sum).
// The post values of a step are loaded at the top of the iteration that COMPUTES the step and consumed by its (deferred)
// epilogue one iteration later: a whole MFMA phase hides their latency.
template <typename T, int C, int RP, bool RELU, int POST>
__global__ __launch_bounds__(512, 2) void conv3x3_wsr_kernel(const WsrArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(C == 128 || C == 256, "channel counts the register budget covers");
  constexpr int KS = C / 128;                 // K parts = waves chained through the accumulator
  constexpr int NG = 8 / KS;                  // cout groups (16 couts) per workgroup
  constexpr int CS = NG * 16;                 // couts per workgroup
  constexpr int PIX = 2 * C;                  // bytes per pixel in global memory
  constexpr int PP = PIX + 32;                // ... and in the ring: the 32-byte pad makes pixel p's chunk c land on 16-byte
                                              // slot (2p + c) mod 16 - fragment reads at every column tap are conflict free and
                                              // the address stays affine (column tap and channel block are instruction offsets)
  constexpr int PPR = (18 * PP + 1023) / 1024;   // DMA pieces per ring row
  constexpr int ROWB = PPR * 1024;
  constexpr int NQ = RP + 2;                  // input rows of a step
  constexpr int NSLOT = (KS + 1) * RP + 4;    // rows of KS steps in use + RP being fetched + 2 + 2 more at a column start
  constexpr int NF = 4 * NQ * 3;              // fragments per step and wave: channel block x input row x column tap
  constexpr int HAND = NSLOT * ROWB;          // accumulator hand-over buffers [group][link][parity][row][lane]
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // cout group and K part. Waves w and w + 4 share a SIMD and the older one (w < 4) wins the matrix pipe whenever both are
  // ready. With two K parts the older wave takes part 1 (hand-over read + the previous step's stores at the top of its
  // iteration) and the younger wave part 0 (every row fetch at the top of its iteration, hand-over write at the end).
  const int g = wave % NG, h = KS == 1 ? 0 : 1 - wave / NG;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  const int s_begin = (int)((long long)p.S * pg / p.n_pg), s_end = (int)((long long)p.S * (pg + 1) / p.n_pg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;

  // ---- this wave's 16 couts x its K part, in registers for the whole launch (A row fr = logical cout cbase + fr)
  const int cbase = slice * CS + g * 16;
  u32x4 wfr[36];
  {
    const int L = cbase + fr, l64 = L & 63;
    const int rem = l64 & 31;
    const int phys = (L & ~63) + (((l64 >> 5) * 2 + ((rem >> 2) & 1)) * 16) + (rem >> 3) * 4 + (rem & 3);   // pack.py's row permutation
    // tiled weight matrix (dp_wtile_off): K step s of part h = plane h * 36 + s, this lane's chunk fq of row phys
    const unsigned char* __restrict__ w = reinterpret_cast<const unsigned char*>(p.w) + dp_wtile_off(phys, h * 36, fq, p.kpad * 2 / 64);
#pragma unroll
    for (int s = 0; s < 36; ++s) wfr[s] = *reinterpret_cast<const u32x4*>(w + s * 1024);
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + cbase + fq * 4);

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_post = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(POST ? p.post : p.in), 0, POST ? p.post_bytes : 0u, 0x00020000);

  const int frag_lane = fr * PP + fq * 16 + h * 256;     // lane part of a fragment address: pixel fr, chunk h * 16 + fq
  const int n_cols = p.n_strips * 16;

  auto advance = [&](WsrStep& st) __attribute__((always_inline)) {
    st.r += RP; st.um += RP; st.first = 0;
    if (st.r >= p.H) {
      st.r = 0; st.um += 2; st.first = 1; st.c0 += 16;
      if (st.c0 >= n_cols) { st.c0 = 0; st.n += 1; }
    }
    if (st.um >= NSLOT) st.um -= NSLOT;
  };
  // Fetching the rows of a step that are not in the ring yet: whole 1 KiB pieces of ring rows; the lanes that fall on a pixel's
  // pad or outside the image read zeros (out-of-range buffer offset). In the steady state (RP new rows) piece j of this wave is
  // always the same (row, piece-in-row), so its per-lane part is computed once: staged pixel and byte offset inside the row.
  // With two K parts the part-0 waves issue every fetch and the part-1 waves every store.
  // Two-part form (round 5, profiles/r5_wsr_experiments.txt): the storing wave - the older wave of its SIMD - reaches the step barrier ~1800
  // cycles before its partner, so ITS epilogue is not deferred: bias, activation, post term and stores run right after its matrix loop, in
  // time it would spend waiting, and its next step starts with the hand-over read instead of 430 cycles of stores.
  constexpr bool LATE = KS > 1 && DP_WSR_LATE;
  constexpr int ND = KS == 1 ? 8 : NG;         // fetching waves
  constexpr int NJ = (RP * PPR + ND - 1) / ND; // pieces per fetching wave and step
  const bool dma_wave = h == 0;
  const int dw = KS == 1 ? wave : g;           // index among the fetching waves
  int dj_px[NJ], dj_off[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pc = dw + ND * j, pr = pc % PPR;
    const int o = pr * 1024 + lane * 16;
    const int px = o / PP, wb = o - px * PP;
    const bool lane_ok = px < 18 && wb < PIX;
    dj_px[j] = lane_ok ? px : 0x10000;
    dj_off[j] = px * PIX + wb;
  }
  // piece j of the steady-state fetch of step st (is_lo / is_hi / is_base: issue_setup)
  int is_lo = 0, is_hi = 0, is_base = 0;
  auto issue_setup = [&](const WsrStep& st) __attribute__((always_inline)) {
    is_lo = st.c0 == 0 ? 1 : 0;                        // staged pixel 0 is column c0 - 1
    is_hi = min(17, p.W - st.c0);                      // column c0 - 1 + px < W
    is_base = ((st.n * p.H + st.r + 1) * p.W + st.c0 - 1) * PIX;    // input row q = 2 (image row r + 1)
  };
  auto issue_piece = [&](const WsrStep& st, auto jj) __attribute__((always_inline)) {
    constexpr int j = decltype(jj)::value;
    const int pc = dw + ND * j;
    if (pc < RP * PPR) {
      const int qq = pc / PPR, pr = pc - qq * PPR;
      const int cnt = (unsigned)(st.r + 1 + qq) < (unsigned)p.H ? is_hi - is_lo + 1 : 0;
      const bool ok = (unsigned)(dj_px[j] - is_lo) < (unsigned)cnt;
      const int off = ok ? is_base + qq * (p.W * PIX) + dj_off[j] : OOB;
      int slot = st.um + 2 + qq;
      if (slot >= NSLOT) slot -= NSLOT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + slot * ROWB + pr * 1024), 16, off, 0, 0, 0);
    }
  };
  // ... and the general form (all NQ rows: the first step of a workgroup and of every column)
  auto issue_all_rows = [&](const WsrStep& st) __attribute__((always_inline)) {
    for (int pc = wave; pc < NQ * PPR; pc += 8) {
      const int q = pc / PPR, pr = pc - q * PPR;
      const int o = pr * 1024 + lane * 16;
      const int px = o / PP, wb = o - px * PP;
      const int row = st.r - 1 + q, col = st.c0 - 1 + px;
      const bool ok = px < 18 && wb < PIX && (unsigned)row < (unsigned)p.H && (unsigned)col < (unsigned)p.W;
      const int off = ok ? ((st.n * p.H + row) * p.W + col) * PIX + wb : OOB;
      int slot = st.um + q;
      if (slot >= NSLOT) slot -= NSLOT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + slot * ROWB + pr * 1024), 16, off, 0, 0, 0);
    }
  };
  // step states: st_i = the step whose rows are fetched this iteration (i + 1), st_w[h] = the step wave part h computes (i - h)
  WsrStep st_i, st_w[KS];
  {
    const int colid = s_begin / p.spc, k = s_begin - colid * p.spc;
    st_i.n = colid / p.n_strips;
    st_i.c0 = (colid - st_i.n * p.n_strips) * 16;
    st_i.r = k * RP;
    st_i.um = 0;
    st_i.first = 1;
  }
  issue_all_rows(st_i);
#pragma unroll
  for (int k = 0; k < KS; ++k) st_w[k] = st_i;
  advance(st_i);
  wsr_wait_vm<0>();
  __builtin_amdgcn_s_barrier();

  // The epilogue of a step is deferred to the top of the wave's NEXT iteration (its accumulators, output offset and row count
  // wait in registers): with two K parts it then runs under the latency of the hand-over read, and the end of an iteration -
  // where the younger wave of every SIMD is the last one computing - has nothing left but the hand-over write.
  f32x4 eacc[RP];
  int e_off = 0, e_rows = 0, e_col = 0;
  bool have_prev = false;
  const int opix = p.cout * 2;
  // post tensor values of the step whose epilogue is pending. POST = 1: the 4 channels of this lane's pixel in each output row.
  // POST = 2: the three half-size rows b, b + 1, b + 2 (b = first source row of the step's first output row; clamped) that the
  // RP <= 3 output rows interpolate between, at this lane's two source columns.
  constexpr int NPV = POST == 1 ? RP : (POST == 2 ? 6 : 1);
  static_assert(POST != 2 || RP <= 3, "three half-size rows cover at most three output rows");
  constexpr int NPL = POST == 1 ? RP : (POST == 2 ? 6 : 0);   // post loads per step of a storing wave
  u32x2 pv[NPV];
  int e_r = 0, e_b = 0;          // POST = 2: first output row of the pending step, its base half-size row
  float p_lx = 0.f;              // POST = 2: this lane's horizontal weight
  auto post_issue = [&](const WsrStep& st) __attribute__((always_inline)) {
    if constexpr (POST == 1) {
      const int col = st.c0 + fr;
      const int off = ((st.n * p.H + st.r) * p.W + col) * opix + (cbase + fq * 4) * 2;
#pragma unroll
      for (int t = 0; t < RP; ++t)
        pv[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, (st.r + t < p.H && col < p.W) ? off + t * (p.W * opix) : OOB, 0, 0);
    } else if constexpr (POST == 2) {
      int x0, x1, y0, y1;
      float ly;
      wsr_bil_src(min(st.c0 + fr, p.W - 1), p.Wl, x0, x1, p_lx);
      wsr_bil_src(st.r, p.Hl, y0, y1, ly);
      e_r = st.r; e_b = y0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int row = min(y0 + k, p.Hl - 1);
        const int off = ((st.n * p.Hl + row) * p.Wl) * opix + (cbase + fq * 4) * 2;
        pv[2 * k] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, off + x0 * opix, 0, 0);
        pv[2 * k + 1] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, off + x1 * opix, 0, 0);
      }
    }
  };
  // POST = 2: the horizontal half of the interpolation once per half-size row (three rows serve the step's output rows)
  float hrow[POST == 2 ? 3 : 1][4];
  auto epi_prepare = [&]() __attribute__((always_inline)) {
    if constexpr (POST == 2) {
      const float hx = 1.f - p_lx;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int sh = (e & 1) * 16;
          const float fa = Elem<T>::unpack((pv[2 * k][e >> 1] >> sh) & 0xffffu), fb = Elem<T>::unpack((pv[2 * k + 1][e >> 1] >> sh) & 0xffffu);
          hrow[k][e] = hx * fa + p_lx * fb;
        }
    }
  };
  auto epi_row = [&](auto tt) __attribute__((always_inline)) {
    constexpr int t = decltype(tt)::value;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = eacc[t][e] + bias[e];
      if constexpr (RELU) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (POST == 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v[2 * e] += Elem<T>::unpack(pv[t][e] & 0xffffu);
        v[2 * e + 1] += Elem<T>::unpack(pv[t][e] >> 16);
      }
    } else if constexpr (POST == 2) {
      int y0, y1;
      float ly;
      wsr_bil_src(e_r + t, p.Hl, y0, y1, ly);     // wave-uniform
      const int i = y0 - e_b, j = y1 - e_b;        // 0 / 1 and 0 / 1 / 2: rows of hrow
      const float hy = 1.f - ly;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float top = i == 0 ? hrow[0][e] : hrow[1][e];
        const float bot = j == 0 ? hrow[0][e] : (j == 1 ? hrow[1][e] : hrow[2][e]);
        v[e] += hy * top + ly * bot;               // ATen's order: hy * (hx * a + lx * b) + ly * (hx * c + lx * d)
      }
    }
    const bool ok = t < e_rows && e_col < p.W;
    const u32x2 pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])};
    __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, ok ? e_off + t * (p.W * opix) : OOB, 0, 0);
  };

  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(k) if constexpr (DP_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  unsigned long long tl = (DP_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int i = 0; i < nst + KS - 1; ++i) {
    const bool fetch = i + 1 < nst && !(DP_EXP & 8);
    const bool fetch_inl = fetch && !st_i.first;
    const bool fetch_all = fetch && st_i.first;
    if (fetch_all) issue_all_rows(st_i);       // every wave takes part
    if (fetch_inl) issue_setup(st_i);
    asm volatile("" ::: "memory");
    DP_STAMP(0)
    const int sw = i - h;
    const bool had_prev = have_prev;
    bool stored = had_prev;        // this iteration issues RP stores (behind its row fetches)
    f32x4 acc[RP];
    if constexpr (LATE) {          // the hand-over read goes out before any bookkeeping: its latency runs under it
      if (h > 0 && sw >= 0 && sw < nst) {
        const unsigned char* hb = smem + HAND + (((g * (KS - 1) + h - 1) * 2 + ((i - 1) & 1)) * RP) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < RP; ++t) acc[t] = *reinterpret_cast<const f32x4*>(hb + t * 1024);
      }
    }
    if (sw >= 0 && sw < nst) {
      WsrStep st = st_w[0];
      if constexpr (KS > 1) { if (h == 1) st = st_w[1]; }
      int va[NQ];       // per input row: ring row base + lane part
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        int slot = st.um + q;
        if (slot >= NSLOT) slot -= NSLOT;
        va[q] = slot * ROWB + frag_lane;
      }
      DP_STAMP(6)

      // Everything that is not a fragment read or an MFMA happens HERE, before the first LDS read of the iteration is in flight:
      // an LDS-DMA issued behind outstanding ds_reads of its wave waits for them (measured: ~150 cycles of wave time per piece
      // between the MFMAs against ~70 up front; 0.567 -> 0.531 ms on the 200x336 level). Order: fetches, then the hand-over
      // read, then the previous step's stores under its latency - so that at the end of the iteration "all but the last RP
      // vector-memory operations" still means "every fetch".
      if (fetch_inl && dma_wave) static_for<0, NJ>([&](auto jj) { issue_piece(st_i, jj); });
      asm volatile("" ::: "memory");      // the fetches stay older than the stores below
      DP_STAMP(7)
      if constexpr (KS > 1) {
        if (h > 0) {
          if constexpr (!LATE) {
            const unsigned char* hb = smem + HAND + (((g * (KS - 1) + h - 1) * 2 + ((i - 1) & 1)) * RP) * 1024 + lane * 16;
#pragma unroll
            for (int t = 0; t < RP; ++t) acc[t] = *reinterpret_cast<const f32x4*>(hb + t * 1024);
          }
        } else {
#pragma unroll
          for (int t = 0; t < RP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int t = 0; t < RP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      DP_STAMP(8)
      if (had_prev) { epi_prepare(); static_for<0, RP>([&](auto tt) { epi_row(tt); }); }
      DP_STAMP(9)
      if constexpr (POST != 0) {          // the storing waves fetch THIS step's post values into the registers the epilogue above
        asm volatile("" ::: "memory");    // has just consumed; they are used one iteration later (a whole MFMA phase of latency cover)
        if (KS == 1 || h == KS - 1) post_issue(st);
      }
      __builtin_amdgcn_sched_barrier(0);

      // fragment f = (channel block cbl, input row q, column tap dx): input row r - 1 + q feeds output row r + t with kernel row q - t
      auto frag = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
        constexpr int f = decltype(ff)::value;
        constexpr int cbl = f / (3 * NQ), q = (f % (3 * NQ)) / 3, dx = f % 3;
        return *reinterpret_cast<const u32x4*>(smem + va[q] + (dx * PP + cbl * 64));
      };
      constexpr int AHEAD = 6;              // fragments in flight ahead of their MFMAs (4 and 8 measure the same within 2 %)
      u32x4 bf[AHEAD + 1];
      static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag(ff); });
      __builtin_amdgcn_sched_barrier(0);
      DP_STAMP(10)
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        constexpr int cbl = f / (3 * NQ), q = (f % (3 * NQ)) / 3, dx = f % 3;
        // order pinned: the scheduler otherwise sinks every read to one MFMA before its use (lgkmcnt(1) chains)
        if constexpr (f + AHEAD < NF && !(DP_EXP & 1)) bf[(f + AHEAD) % (AHEAD + 1)] = frag(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, RP>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          if constexpr (q - t >= 0 && q - t <= 2 && !((DP_EXP & 4) && f >= AHEAD + 1)) Mma<T>::run(wfr[cbl * 9 + (q - t) * 3 + dx], bf[f % (AHEAD + 1)], acc[t]);
          if constexpr ((DP_EXP & 4) && t == 0 && f >= AHEAD + 1) acc[0][0] += __builtin_bit_cast(float, bf[f % (AHEAD + 1)][0]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      have_prev = false;
      DP_STAMP(1)
      if (KS > 1 && h < KS - 1) {
        unsigned char* hb = smem + HAND + (((g * (KS - 1) + h) * 2 + (i & 1)) * RP) * 1024 + lane * 16;
#pragma unroll
        for (int t = 0; t < RP; ++t) *reinterpret_cast<f32x4*>(hb + t * 1024) = acc[t];
      } else {
        // 4 consecutive channels of pixel fr of each output row per lane, the four lanes of a pixel 32 contiguous bytes
#pragma unroll
        for (int t = 0; t < RP; ++t) eacc[t] = acc[t];
        e_col = st.c0 + fr;
        e_off = ((st.n * p.H + st.r) * p.W + e_col) * opix + (cbase + fq * 4) * 2;
        e_rows = p.H - st.r;
        if constexpr (LATE) {      // this wave reaches the step barrier long before its partner: bias, activation, post term and stores now
          epi_prepare();
          static_for<0, RP>([&](auto tt) { epi_row(tt); });
          stored = true;
        } else {
          have_prev = true;
        }
      }
    } else {
      if (fetch_inl && dma_wave) static_for<0, NJ>([&](auto jj) { issue_piece(st_i, jj); });
      asm volatile("" ::: "memory");
      if (had_prev) { epi_prepare(); static_for<0, RP>([&](auto tt) { epi_row(tt); }); }
      have_prev = false;
    }
    DP_STAMP(2)
    // the rows fetched in this iteration have landed (they are older than this iteration's stores; with two K parts a wave
    // either fetches or stores)
    // (the youngest vector-memory operations of a storing wave are its RP stores and, behind them, its NPL post loads)
    constexpr bool computes_post = POST != 0;
    const int npl = (computes_post && sw >= 0 && sw < nst && (KS == 1 || h == KS - 1)) ? NPL : 0;
    // (`stored`: RP stores were issued in this iteration - at its top from the deferred epilogue, or after the matrix loop by a storing wave of the
    // LATE form; either way behind the fetches, and the post loads sit between the fetches and the loop)
    if (dma_wave || fetch_all) {
      if (stored) { if (npl) wsr_wait_vm<RP + NPL>(); else wsr_wait_vm<RP>(); }
      else { if (npl) wsr_wait_vm<NPL>(); else wsr_wait_vm<0>(); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DP_STAMP(3)
    __builtin_amdgcn_s_barrier();
    DP_STAMP(4)
    if constexpr (KS > 1) st_w[1] = st_w[0];
    st_w[0] = st_i;
    advance(st_i);
  }
  if (have_prev) { epi_prepare(); static_for<0, RP>([&](auto tt) { epi_row(tt); }); }
  if constexpr (DP_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 5; ++k) p.dbg[(blockIdx.x * 8 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 8 + wave) * 16 + 5] = nst;
#pragma unroll
      for (int k = 6; k < 11; ++k) p.dbg[(blockIdx.x * 8 + wave) * 16 + k] = ph[k];
    }
  }
#undef DP_STAMP
}

template <typename T, int C, int RP, bool RELU, int POST>
int launch_wsr_r(WsrArgs a, hipStream_t stream) {
  constexpr int KS = C / 128, NG = 8 / KS, CS = NG * 16;
  constexpr int ROWB = (18 * (2 * C + 32) + 1023) / 1024 * 1024, NSLOT = (KS + 1) * RP + 4;
  constexpr int lds = NSLOT * ROWB + (KS > 1 ? NG * (KS - 1) * 2 * RP * 1024 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wsr_kernel<T, C, RP, RELU, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    cus = ws_num_cus();
    attr_set = true;
  }
  a.n_strips = (a.W + 15) / 16;
  a.spc = (a.H + RP - 1) / RP;
  a.S = a.N * a.n_strips * a.spc;
  a.n_slices = a.cout / CS;
  int groups = cus / (8 * a.n_slices);
  if (a.over < 0) { groups += a.over; a.over = 1; }    // shared_chip = 2: -over groups of 8 * n_slices CUs are left to the other stream
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8 * (a.over > 1 ? a.over : 1);     // over-decomposition: more, shorter workgroups than CUs
  a.dbg = nullptr;
#if DP_EXP & 16
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_pg * a.n_slices;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 16 * 8 * 4096);
  a.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 16 * 8 * nblk, stream);
#endif
  hipLaunchKernelGGL((conv3x3_wsr_kernel<T, C, RP, RELU, POST>), dim3(a.n_pg * a.n_slices), dim3(512), lds, stream, a);
#if DP_EXP & 16
  {
    static int shown = 0;
    if (shown++ == 4) {   // a warm launch
      (void)hipStreamSynchronize(stream);
      unsigned long long* hbuf = (unsigned long long*)malloc(sizeof(unsigned long long) * 128 * nblk);
      (void)hipMemcpy(hbuf, dbg, sizeof(unsigned long long) * 128 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) {
        double sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, n = 0;
        for (int b = 0; b < nblk; ++b) { for (int k = 0; k < 11; ++k) if (k != 5) sum[k] += (double)hbuf[(b * 8 + w) * 16 + k]; n += (double)hbuf[(b * 8 + w) * 16 + 5]; }
        fprintf(stderr, "wsr wave %d: per step cycles: top %.0f  [ring addresses %.0f  fetch pieces %.0f  hand-over read %.0f  deferred stores %.0f  post loads + first reads %.0f]  loop %.0f  after the loop %.0f  vmcnt/lgkm wait %.0f  barrier %.0f  (steps/wg %.1f)\n", w,
                sum[0] / n, sum[6] / n, sum[7] / n, sum[8] / n, sum[9] / n, sum[10] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, n / nblk);
      }
      free(hbuf);
    }
  }
#endif
  return dp_check_launch("conv3x3_wsr_kernel");
}

template <typename T, int C, int RP>
int launch_wsr(const WsrArgs& a, hipStream_t stream, int post_mode) {
  if constexpr (C == 256) {      // the decoder's level sum (256 channels, ReLU heads): the only caller of the post modes
    if (post_mode == 1 && a.relu) return launch_wsr_r<T, C, RP, true, 1>(a, stream);
    if (post_mode == 2 && a.relu) return launch_wsr_r<T, C, RP, true, 2>(a, stream);
  }
  if (post_mode != 0) return dp_fail(DP_ERR_UNSUPPORTED, "conv3x3_wsr_kernel: post_res needs 256 channels and ReLU");
  return a.relu ? launch_wsr_r<T, C, RP, true, 0>(a, stream) : launch_wsr_r<T, C, RP, false, 0>(a, stream);
}

constexpr int kWsrRP128 = 4, kWsrRP256 = 3;

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): is this launch one of the C -> C 3x3 / pad 1 / stride 1 layers the kernel is written for?
bool dp_conv_wsr_ok(const dp_conv_params* p) {
  const int mode = (int)dp_policy().conv_ws;    // policy override: 0 keeps these layers on the ring kernels, 4 only those that share the chip
  if (mode == 0 || (mode == 4 && p->shared_chip)) return false;
  const long long M = (long long)p->N * p->H * p->W;
  const bool shape = (p->Cin == 128 && p->Cout == 128 && p->Cout_w == 128) || (p->Cin == 256 && p->Cout == 256 && p->Cout_w == 256);
  const int rp = p->Cin == 128 ? kWsrRP128 : kWsrRP256;
  // fewest output pixels of a launch the kernel takes (policy key ws_min_m: calibration override). Round 2 drew the line at 2048; measured again at
  // batch 1 (one 25 x 42 / 13 x 21 map: the p5 / p6 levels of a single frame) the ring kernel needs 27 / 26 us for its 72 K planes, this
  // kernel 12 / 10 us - same bits either way, so the line only moves time
  // (launches with a post tensor keep the old line: whether the decoder's level sum is folded into the convolutions - a per-geometry
  // choice that moves rounding points - is decided by asking this function, and that choice stays what the parity tests pinned)
  const long long min_m = p->post_res ? 2048 : dp_policy().ws_min_m;
  return (p->dtype == DP_BF16 || p->dtype == DP_F16) && !p->n_dev && shape && p->ntaps == 9 && p->Kpad == 9 * p->Cin && p->stride == 1 &&
         (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == -1 && p->wi_off == -1 && p->H == p->Ho && p->W == p->Wo &&
         !p->residual && !p->out_f32 && !p->head_out && p->out && p->osW == p->Cout && p->osH == (long long)p->W * p->Cout &&
         p->osN == (long long)p->H * p->W * p->Cout && p->H >= 2 * rp && M >= min_m && M * 2 * p->Cin < (1ll << 31) &&
         (((uintptr_t)p->in | (uintptr_t)p->out | (uintptr_t)p->weight | (uintptr_t)p->post_res) & 15) == 0 &&
         (p->post_res == nullptr ? p->post_mode == 0
                                 : (p->Cin == 256 && p->relu && (p->post_mode == 1 || (p->post_mode == 2 && p->H % 2 == 0 && p->W % 2 == 0))));
}

int dp_conv_wsr_launch(const dp_conv_params* p, dp_stream_t stream) {
  WsrArgs a;
  a.in = p->in; a.w = p->weight; a.bias = p->bias; a.out = p->out;
  a.N = p->N; a.H = p->H; a.W = p->W; a.relu = p->relu; a.kpad = p->Kpad; a.cout = p->Cout;
  a.n_strips = a.spc = a.n_slices = a.n_pg = a.S = 0;
  a.dbg = nullptr;
  a.post = p->post_res;
  a.Hl = p->H / 2; a.Wl = p->W / 2;
  a.post_bytes = p->post_res ? (unsigned)((long long)p->N * (p->post_mode == 2 ? (long long)a.Hl * a.Wl : (long long)p->H * p->W) * p->Cout * 2) : 0u;
  {
    // A launch that has the chip to itself: one persistent workgroup per CU. One that runs beside another stream's large launches
    // (dp_conv_params.shared_chip = 1): two workgroups per CU slot, each with half the steps - a static split over exactly as many
    // workgroups as CUs cannot rebalance when some CUs are taken, the dispatcher can with workgroups to spare; the price is a
    // second weight prologue per CU. One that runs beside a chain of small, latency-bound launches (shared_chip = 2: the decoder
    // beside the proposal top-k / NMS chain): one workgroup per CU on all but one group of CUs, which stay free for that chain -
    // its single-wave workgroups otherwise wait for a workgroup of this launch to END before they get a CU (every CU's LDS and
    // registers are taken), which stretched the chain 2 - 3x (profiles/r3_timeline_*.txt). A/B knobs for all three cases.
    const DpPolicy& pol = dp_policy();
    if (p->shared_chip == 2) a.over = -(int)pol.ws_reserve;
    else a.over = (int)(p->shared_chip ? pol.ws_over_shared : pol.ws_over_alone);
    if (a.over == 0) a.over = 1;
  }
  a.in_bytes = (unsigned)((long long)p->N * p->H * p->W * p->Cin * 2);
  a.out_bytes = (unsigned)((long long)p->N * p->H * p->W * p->Cout * 2);
  hipStream_t s = as_stream(stream);
  const int pm = p->post_res ? p->post_mode : 0;
  if (p->Cin == 128) return p->dtype == DP_BF16 ? launch_wsr<uint16_t, 128, kWsrRP128>(a, s, pm) : launch_wsr<f16_t, 128, kWsrRP128>(a, s, pm);
  return p->dtype == DP_BF16 ? launch_wsr<uint16_t, 256, kWsrRP256>(a, s, pm) : launch_wsr<f16_t, 256, kWsrRP256>(a, s, pm);
}
