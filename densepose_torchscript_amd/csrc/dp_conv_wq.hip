// Weight-stationary 3x3 convolution, 256 -> 256 channels, on v_mfma_f32_32x32x16: ONE wave per SIMD, each with a
// 32-cout x 1152-K slice of the weights in its 512 registers. Behind dp_conv2d_nhwc (dp_conv.hip, kernel class 10).
// Layers: res4 conv2 (/root/reference/detectron2/modeling/backbone/resnet.py:195-197), the FPN output convolutions (fpn.py:134-135,
// fpn.py:157), the DensePose decoder's scale heads (densepose/modeling/roi_heads/roi_head.py:48-68).
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"
#include <stdlib.h>

#ifndef DP_WSQ_EXP
#define DP_WSQ_EXP 0      // diagnostic builds: 1 no fragment reads, 4 no MFMAs, 8 no row fetches, 16 in-kernel phase stamps
#endif

// =====================================================================================================
// Why this form (round 6; profiles/r5_wsr_experiments.txt has the measurements that led here). conv3x3_wsr_kernel<256> (dp_conv_ws.hip)
// issues ~1300 instructions per SIMD and step for 216 v_mfma_f32_16x16x32 (3456 pipe cycles): six instructions per MFMA where four
// keep the pipe full - two waves per SIMD each pay their own step bookkeeping, every 16 x 16 x 32 block of FLOPs costs one MFMA
// issue, and a 16-cycle MFMA leaves room for two other instructions. Here
//   * the instruction is v_mfma_f32_32x32x16: twice the FLOPs per issue, 32 pipe cycles of which 24 are free issue time;
//   * a workgroup is FOUR waves, one per SIMD (launch bounds 256: 512 registers per lane). Wave (g, h) holds the weights of couts
//     [32 g, 32 g + 32) of the workgroup's 64-cout slice for the channels [128 h, 128 h + 128) of all nine taps: 72 K steps of
//     16 = 288 registers. One set of step bookkeeping per SIMD instead of two;
//   * a step is FOUR output rows of a 16-pixel column strip. The 32 columns of an MFMA are two rows two apart: column n is
//     pixel (row t + 2 (n >> 4), column n & 15) of output pair t in {0, 1}. The B fragment F(q) - input rows q and q + 2 of the
//     six a step reads - then serves pair 0 with kernel row q and pair 1 with kernel row q - 1: four fragment reads per six MFMAs
//     (rows one apart would need five). 144 MFMAs = 4608 pipe cycles and 96 fragment reads per wave and step;
//   * both K halves work on the SAME step (the chained form of dp_conv_ws.hip would need 16 ring rows: 160 KiB). At the end of
//     a step each wave hands the half of its accumulators that belongs to its partner through LDS (parity double-buffered, 32 KiB)
//     and finishes the other half itself: out = (bias + sum over channels 0..127) + (sum over channels 128..255), the bias being
//     the initial accumulator value of the h = 0 wave. fp32 addition commutes, so both owners compute the same expression;
//   * everything that is not an MFMA rides in the MFMAs' shadow: the ten LDS-DMA pieces of the row a wave fetches for the next
//     step, the exchange reads, sum / ReLU / pack / store of the PREVIOUS step - placed by hand between the MFMAs of the
//     unrolled loop, all unconditional (a step that has nothing to fetch or store issues them with out-of-range offsets);
//   * ring pixels have a 16-byte pad (pitch 528): granule (p + c) mod 16 for chunk c of pixel p, which is conflict-free for
//     the 32x32x16 B fragment (ds_read_b128 serves lanes {0-3, 12-15, 20-27} together: pixels 0-3, 12-15 of one row and 4-11 of
//     the other).
// The summation order (K halves, 16-channel K steps, the 32x32x16 instruction's own order) is this kernel's own: a layer runs
// here for EVERY batch size or never - dp_conv_wsq_ok looks at the per-image geometry and the channel counts only, and a batch
// whose tensors pass the 32-bit offset range is cut into image chunks inside dp_conv_wsq_launch.
// =====================================================================================================
#ifndef DP_WSQ_AHEAD
#define DP_WSQ_AHEAD 6    // fragments in flight ahead of their MFMAs
#endif
#ifndef DP_WSQ_ROWS_VIA_REGS
#define DP_WSQ_ROWS_VIA_REGS 0   // steady-state row fetches: buffer_load into registers + ds_write_b128 later in the step (0: LDS-DMA)
#endif
#ifndef DP_WSQ_DMA_GAP
#define DP_WSQ_DMA_GAP 2  // fragments between two LDS-DMA pieces of a wave (2 / 4 / 7 measure the same; every piece must be OLDER than the step's stores: see the vmcnt at its end)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));

template <typename T>
struct Mma32;
template <>
struct Mma32<uint16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma32<f16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

int wq_num_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}

struct WsqArgs {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  int N, H, W, cout;
  int n_strips, spc, n_slices, n_pg;
  int S;                       // steps per cout slice
  unsigned in_bytes, out_bytes;
  const void* post;            // dp_conv_params.post_res
  int Hl, Wl;                  // POST = 2: geometry of the half-size map
  unsigned post_bytes;
  const int* n_dev;            // dp_conv_params.n_dev: only the first *n_dev - n0 of the N images of this launch hold data
  int n0;
  unsigned long long* dbg;
};

struct WsqStep {
  int n, c0, r, um, first;
};

constexpr int kWqRP = 4;                     // output rows per step
constexpr int kWqPix = 512;                  // bytes per pixel in global memory (256 channels, 16-bit)
constexpr int kWqPP = 528;                   // ... and in the ring
constexpr int kWqPPR = 10;                   // 1 KiB DMA pieces per ring row (18 pixels x 528 B = 9504 B)
constexpr int kWqRowB = kWqPPR * 1024;
constexpr int kWqSlots = 2 * kWqRP + 4;      // six rows in use + four being fetched + two more at a column start
constexpr int kWqXch = kWqSlots * kWqRowB;   // exchange buffers [wave][parity][pair][2][lane] of 16 bytes
constexpr int kWqLds = kWqXch + 4 * 2 * 4 * 1024;
constexpr int kWqAgprFrags = 64;              // weight fragments kept in accumulation registers (of 72)
constexpr unsigned kLaneInv = 0x80000000u;   // lane of a DMA piece that reads nothing (pad, column outside the image)
constexpr unsigned kRowInv = 0x7fffc000u;    // row base of a row outside the image: + any lane offset stays >= the tensor size

__device__ __forceinline__ void wq_bil_src(int o, int n, int& i0, int& i1, float& l) {   // ATen's bilinear x2 source (dp_ops.hip)
  float s = ((float)o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}

// POST: 0 = plain, 1 = out = act(..) + post[same pixel], 2 = out = act(..) + bilinear_x2(post) (the decoder's level sum).
template <typename T, bool RELU, int POST>
__global__ __launch_bounds__(256, 1) void conv3x3_wsq_kernel(const WsqArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int RP = kWqRP, PIX = kWqPix, PP = kWqPP, PPR = kWqPPR, ROWB = kWqRowB, NSLOT = kWqSlots, XCH = kWqXch;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, hi = lane >> 5;
  const int rs = n32 >> 4, pxl = n32 & 15;          // this lane's MFMA column: row select (0 / 1 = two rows further down) and pixel of the strip
  const int g = wave & 1, h = wave >> 1;            // cout group of 32, K half
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  int S = p.S;
  if (p.n_dev) S = min(max(*p.n_dev - p.n0, 0), p.N) * p.n_strips * p.spc;      // the live images only (uniform: a scalar load)
  const int s_begin = (int)((long long)S * pg / p.n_pg), s_end = (int)((long long)S * (pg + 1) / p.n_pg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;

  // ---- weights: A row m = j + 8 i + 4 hi' (the C/D layout's row of register 4 i + j in lane half hi') holds the cout
  //      16 ((i >> 1) ^ h) + 8 hi' + 4 (i & 1) + j of the group: registers 0..7 of every lane then are 8 consecutive couts of the
  //      half the wave finishes itself (couts 16 h + 8 hi + 0..7 of the group), registers 8..15 the 8 it hands to its partner
  const int cgrp = slice * 64 + g * 32;
  u32x4 wq[72];
  {
    const int m = n32, mi = m >> 3, mh = (m >> 2) & 1, mj = m & 3;
    const int L = cgrp + 16 * ((mi >> 1) ^ h) + 8 * mh + 4 * (mi & 1) + mj, l64 = L & 63;
    const int rem = l64 & 31;
    const int phys = (L & ~63) + (((l64 >> 5) * 2 + ((rem >> 2) & 1)) * 16) + (rem >> 3) * 4 + (rem & 3);   // pack.py's row permutation
    // K step 2 s + kk of this wave = chunks 2 kk, 2 kk + 1 of plane h * 36 + s (s = channel block x tap); this lane holds chunk 2 kk + hi
    const unsigned char* __restrict__ w = reinterpret_cast<const unsigned char*>(p.w) + dp_wtile_off(phys, h * 36, hi, 72);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      wq[2 * s] = *reinterpret_cast<const u32x4*>(w + s * 1024);
      wq[2 * s + 1] = *reinterpret_cast<const u32x4*>(w + s * 1024 + 32);
    }
    // Pin the register class of every fragment once: 64 of them fill the accumulation registers (the MFMA takes its A operand from
    // there), 8 stay in vector registers. Left to itself the allocator treats the 288 registers as vector-register values spilled
    // to accumulation registers and reloads ~40 fragments per step with four copies each in front of their MFMAs.
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      if (s < kWqAgprFrags) asm volatile("" : "+a"(wq[s]));
      else asm volatile("" : "+v"(wq[s]));
    }
  }
  // initial accumulator values: the bias in the h = 0 wave (registers 4 i + j = cout 16 (i >> 1) + 8 hi + 4 (i & 1) + j), zero in the other
  f32x16 binit;
  {
    const float* bp = p.bias + cgrp + 8 * hi;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 16), b3 = *reinterpret_cast<const f32x4*>(bp + 20);
    const float m = h == 0 ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      binit[e] = b0[e] * m; binit[4 + e] = b1[e] * m; binit[8 + e] = b2[e] * m; binit[12 + e] = b3[e] * m;
    }
  }

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_post = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(POST ? p.post : p.in), 0, POST ? p.post_bytes : 0u, 0x00020000);

  const int frag_lane = pxl * PP + h * 256 + hi * 16;     // lane part of a fragment address
  const int n_cols = p.n_strips * 16;
  const int opix = p.cout * 2;
  const int ocb = (cgrp + 16 * h + 8 * hi) * 2;             // byte offset of this lane's 8 output channels inside a pixel

  auto advance = [&](WsqStep& st) __attribute__((always_inline)) {
    st.r += RP; st.um += RP; st.first = 0;
    if (st.r >= p.H) {
      st.r = 0; st.um += 2; st.first = 1; st.c0 += 16;
      if (st.c0 >= n_cols) { st.c0 = 0; st.n += 1; }
    }
    if (st.um >= NSLOT) st.um -= NSLOT;
  };

  // ---- row fetches: ring row = 18 pixels (columns c0 - 1 .. c0 + 16) at pitch 528 = 594 granules of 16 bytes in 10 pieces of 64.
  //      dl[pr] = this lane's source offset inside the row for piece pr, or kLaneInv (pad granule, column outside the image);
  //      it depends on the column strip only and is recomputed when the strip changes.
  unsigned dl[PPR];
  auto lanes_for_strip = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int pr = 0; pr < PPR; ++pr) {
      const int gi = pr * 64 + lane, px = gi / 33, sub = gi - px * 33;
      const bool ok = px < 18 && sub < 32 && (unsigned)(c0 - 1 + px) < (unsigned)p.W;
      dl[pr] = ok ? (unsigned)(px * PIX + sub * 16) : kLaneInv;
    }
  };
  // source offset of pixel column c0 - 1 of input row q (image row r - 1 + q) of step st, or kRowInv
  auto row_base = [&](const WsqStep& st, int q, bool live) __attribute__((always_inline)) -> unsigned {
    const int row = st.r - 1 + q;
    return (live && (unsigned)row < (unsigned)p.H) ? (unsigned)(((st.n * p.H + row) * p.W + st.c0 - 1) * PIX) : kRowInv;
  };
  auto slot_of = [&](const WsqStep& st, int q) __attribute__((always_inline)) -> int {
    int s = st.um + q;
    if (s >= NSLOT) s -= NSLOT;
    return s;
  };
  auto fetch_piece = [&](int slot_bytes, unsigned rbase, auto prr) __attribute__((always_inline)) {
    constexpr int pr = decltype(prr)::value;
    if constexpr (!(DP_WSQ_EXP & 8))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + slot_bytes + pr * 1024), 16, (int)(dl[pr] + rbase), 0, 0, 0);
  };
  // the two extra rows (q = 0, 1) of a step that starts a column: wave w fetches pieces 5 (w >> 1) .. + 5 of row q = w & 1
  auto fetch_head_rows = [&](const WsqStep& st, bool live) __attribute__((always_inline)) {
    const int q = wave & 1;
    const int sb = slot_of(st, q) * ROWB;
    const unsigned rb = row_base(st, q, live);
    if (wave < 2) static_for<0, 5>([&](auto prr) { fetch_piece(sb, rb, prr); });
    else static_for<5, 10>([&](auto prr) { fetch_piece(sb, rb, prr); });
  };

  // ---- step states: st_c = the step computed in this iteration, st_n = the next one (its rows are fetched now)
  WsqStep st_c, st_n;
  {
    const int colid = s_begin / p.spc, k = s_begin - colid * p.spc;
    st_c.n = colid / p.n_strips;
    st_c.c0 = (colid - st_c.n * p.n_strips) * 16;
    st_c.r = k * RP;
    st_c.um = 0;
    st_c.first = 1;
  }
  lanes_for_strip(st_c.c0);
  fetch_head_rows(st_c, true);
  {
    const int sb = slot_of(st_c, 2 + wave) * ROWB;
    const unsigned rb = row_base(st_c, 2 + wave, true);
    static_for<0, PPR>([&](auto prr) { fetch_piece(sb, rb, prr); });
  }
  st_n = st_c;
  advance(st_n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- pending epilogue (the step computed in the previous iteration): this wave's half of its accumulators, where they go
  f32x4 eacc[2][2];
  int e_off = 0, e_rows = 0;     // e_rows: rows left below the lane's first output row (0 = nothing to store: column outside the image)
#pragma unroll
  for (int t = 0; t < 2; ++t) { eacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; eacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // post tensor values of the pending step. POST = 1: the lane's 8 channels at its pixel of each pair. POST = 2: the four
  // half-size neighbours of its pixel of each pair + the two weights.
  constexpr int NPV = POST == 1 ? 2 : (POST == 2 ? 8 : 1);
  u32x4 pv[NPV];
  float p_lx = 0.f, p_ly[2] = {0.f, 0.f};
  auto post_issue = [&](const WsqStep& st) __attribute__((always_inline)) {
    const int col = st.c0 + pxl;
    if constexpr (POST == 1) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = st.r + t + 2 * rs;
        const bool ok = row < p.H && col < p.W;
        pv[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_post, ok ? ((st.n * p.H + row) * p.W + col) * opix + ocb : (int)kLaneInv, 0, 0));
      }
    } else if constexpr (POST == 2) {
      int x0, x1;
      wq_bil_src(min(col, p.W - 1), p.Wl, x0, x1, p_lx);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int y0, y1;
        wq_bil_src(min(st.r + t + 2 * rs, p.H - 1), p.Hl, y0, y1, p_ly[t]);
        const int o0 = ((st.n * p.Hl + y0) * p.Wl) * opix + ocb, o1 = ((st.n * p.Hl + y1) * p.Wl) * opix + ocb;
        pv[4 * t + 0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_post, o0 + x0 * opix, 0, 0));
        pv[4 * t + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_post, o0 + x1 * opix, 0, 0));
        pv[4 * t + 2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_post, o1 + x0 * opix, 0, 0));
        pv[4 * t + 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_post, o1 + x1 * opix, 0, 0));
      }
    }
  };
  // epilogue pieces of the pending step, pair t: partner's sums (read from its exchange buffer), + own, activation, post term, store
  f32x4 eo[2][2];
  auto epi_read = [&](int par, auto tt) __attribute__((always_inline)) {
    constexpr int t = decltype(tt)::value;
    const unsigned char* xb = smem + XCH + ((((wave ^ 2) * 2 + par) * 2 + t) * 2) * 1024 + lane * 16;
    eo[t][0] = *reinterpret_cast<const f32x4*>(xb);
    eo[t][1] = *reinterpret_cast<const f32x4*>(xb + 1024);
  };
  auto epi_store = [&](auto tt) __attribute__((always_inline)) {
    constexpr int t = decltype(tt)::value;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = eacc[t][e >> 2][e & 3] + eo[t][e >> 2][e & 3];
      if constexpr (RELU) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (POST == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] += Elem<T>::unpack(pv[t][e] & 0xffffu);
        v[2 * e + 1] += Elem<T>::unpack(pv[t][e] >> 16);
      }
    } else if constexpr (POST == 2) {
      const float hx = 1.f - p_lx, hy = 1.f - p_ly[t];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sh = (e & 1) * 16;
        const float a = Elem<T>::unpack((pv[4 * t + 0][e >> 1] >> sh) & 0xffffu), bb = Elem<T>::unpack((pv[4 * t + 1][e >> 1] >> sh) & 0xffffu);
        const float c = Elem<T>::unpack((pv[4 * t + 2][e >> 1] >> sh) & 0xffffu), d = Elem<T>::unpack((pv[4 * t + 3][e >> 1] >> sh) & 0xffffu);
        v[e] += hy * (hx * a + p_lx * bb) + p_ly[t] * (hx * c + p_lx * d);     // ATen's order
      }
    }
    const u32x4 pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7])};
    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, t < e_rows ? e_off + t * (p.W * opix) : (int)kLaneInv, 0, 0);
  };

  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(k) if constexpr (DP_WSQ_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  unsigned long long tl = (DP_WSQ_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long t_loop0 = tl, rt0 = (DP_WSQ_EXP & 16) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // ---- loop state, one step ahead of its use: everything below is computed INSIDE the previous step's matrix loop (in the MFMAs'
  //      shadow), so that a step starts with its first fragment reads right behind the barrier.
  //      va[q]: fragment F(q) of the step computed now = ring rows q (lanes 0..15 of each half) and q + 2 (lanes 16..31);
  //      nsb / nrb: ring slot (bytes) and source row base of the row this wave fetches for the next step; n_first: that step starts a column
  const unsigned rs2 = 2 * rs;
  auto frag_bases = [&](const WsqStep& st, int (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned s = (unsigned)(st.um + q) + rs2;
      v[q] = (int)(min(s, s - (unsigned)NSLOT) * ROWB) + frag_lane;      // s < NSLOT: s - NSLOT wraps to a huge value
    }
  };
  int va[4], va_n[4];
  frag_bases(st_c, va);
  int nsb = slot_of(st_n, 2 + wave) * ROWB, nsb_n = 0;
  unsigned nrb = row_base(st_n, 2 + wave, nst > 1), nrb_n = 0;
  bool n_first = nst > 1 && st_n.first;
  constexpr int NF = 96, AHEAD = DP_WSQ_AHEAD, DMA_GAP = DP_WSQ_ROWS_VIA_REGS ? 4 : DP_WSQ_DMA_GAP, ST0 = 40;
  static_assert(DP_WSQ_ROWS_VIA_REGS || 2 + DMA_GAP * (kWqPPR - 1) < 24, "the last LDS-DMA piece of a step is issued before its first store / exchange read: vmcnt(stores + post loads) then covers every piece");
  u32x4 rowbuf[PPR];

  for (int i = 0; i < nst; ++i) {
    const int par = i & 1;
    f32x16 acc[2] = {binit, binit};
    // fragment f = (channel block cbl of 32, input-row pair q, column tap dx, K step kk of 16): B[k][n] = channel 32 cbl + 16 kk + 8 hi + j
    // of pixel (row q + 2 rs, column pxl + dx) of the ring
    auto frag = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
      constexpr int f = decltype(ff)::value;
      constexpr int cbl = f / 24, q = (f % 24) / 6, dx = (f % 6) / 2, kk = f % 2;
      return *reinterpret_cast<const u32x4*>(smem + va[q] + (dx * PP + cbl * 64 + kk * 32));
    };
    u32x4 bf[AHEAD + 1];
    static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag(ff); });
    __builtin_amdgcn_sched_barrier(0);
    DP_STAMP(0)
    static_for<0, NF>([&](auto ff) {
      constexpr int f = decltype(ff)::value;
      constexpr int cbl = f / 24, q = (f % 24) / 6, dx = (f % 6) / 2, kk = f % 2;
      if constexpr (f + AHEAD < NF && !(DP_WSQ_EXP & 1)) bf[(f + AHEAD) % (AHEAD + 1)] = frag(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DP_WSQ_EXP & 32) && f % 24 == 0 && f > 0) { DP_STAMP(7 + f / 24) }
      // ---- side work in the MFMAs' shadow
      if constexpr (f == 0) {               // the next step starts a column: lane offsets of its strip, its two head rows
        if (n_first) {
          lanes_for_strip(st_n.c0);
          fetch_head_rows(st_n, true);
        }
      }
      // the row this wave fetches for the next step, ten 1 KiB pieces. An LDS-DMA piece issued behind outstanding ds_reads of its wave waits
      // for them (a matrix loop always has ~6 in flight: ~45 cycles of stalled MFMA issue per piece, profiles/r6_wsq_experiments.txt), so the
      // pieces go through registers: buffer loads early in the step, ds_write_b128 once they have arrived
      if constexpr (DP_WSQ_ROWS_VIA_REGS) {
        if constexpr (f >= 2 && f < 2 + 2 * PPR && (f & 1) == 0) {
          constexpr int pr = (f - 2) / 2;
          if constexpr (!(DP_WSQ_EXP & 8)) rowbuf[pr] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(dl[pr] + nrb), 0, 0));
        }
        if constexpr (f >= ST0 && f < ST0 + DMA_GAP * PPR && (f - ST0) % DMA_GAP == 0) {
          constexpr int pr = (f - ST0) / DMA_GAP;
          if constexpr (!(DP_WSQ_EXP & 8)) *reinterpret_cast<u32x4*>(smem + nsb + pr * 1024 + lane * 16) = rowbuf[pr];
        }
      } else {
        if constexpr (f >= 2 && f < 2 + DMA_GAP * PPR && (f - 2) % DMA_GAP == 0) fetch_piece(nsb, nrb, std::integral_constant<int, (f >= 2 && f < 2 + DMA_GAP * PPR ? (f - 2) / DMA_GAP : 0)>{});
      }
      if constexpr (f == 24) epi_read(par ^ 1, std::integral_constant<int, 0>{});
      if constexpr (f == 26) epi_read(par ^ 1, std::integral_constant<int, 1>{});
      if constexpr (f == 45) epi_store(std::integral_constant<int, 0>{});
      if constexpr (f == 59) epi_store(std::integral_constant<int, 1>{});
      if constexpr (f == 67 && POST != 0) post_issue(st_c);
      if constexpr (f == 66) {              // where this step's outputs go (its epilogue runs inside the next step)
        const int col = st_c.c0 + pxl, row = st_c.r + 2 * rs;
        e_off = ((st_c.n * p.H + row) * p.W + col) * opix + ocb;
        e_rows = col < p.W ? p.H - row : 0;
      }
      if constexpr (f == 70) { st_c = st_n; advance(st_n); }
      if constexpr (f == 72) frag_bases(st_c, va_n);
      if constexpr (f == 74) {
        nsb_n = slot_of(st_n, 2 + wave) * ROWB;
        nrb_n = row_base(st_n, 2 + wave, i + 2 < nst);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- pair t uses F(q) with kernel row q - t
      static_for<0, 2>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        if constexpr (q - t >= 0 && q - t <= 2) {
          if constexpr (!(DP_WSQ_EXP & 4)) Mma32<T>::run(wq[(cbl * 9 + (q - t) * 3 + dx) * 2 + kk], bf[f % (AHEAD + 1)], acc[t]);
          else acc[t][0] += __builtin_bit_cast(float, bf[f % (AHEAD + 1)][0]) * __builtin_bit_cast(float, wq[(cbl * 9 + (q - t) * 3 + dx) * 2 + kk][0]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DP_WSQ_EXP & 32) { DP_STAMP(11) } else { DP_STAMP(1) }
    // ---- hand registers 8..15 of both pairs to the partner, keep 0..7 for the epilogue that runs inside the next step
    {
      unsigned char* xb = smem + XCH + (((wave * 2 + par) * 2) * 2) * 1024 + lane * 16;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        *reinterpret_cast<f32x4*>(xb + (t * 2) * 1024) = f32x4{acc[t][8], acc[t][9], acc[t][10], acc[t][11]};
        *reinterpret_cast<f32x4*>(xb + (t * 2 + 1) * 1024) = f32x4{acc[t][12], acc[t][13], acc[t][14], acc[t][15]};
        eacc[t][0] = f32x4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
        eacc[t][1] = f32x4{acc[t][4], acc[t][5], acc[t][6], acc[t][7]};
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) va[q] = va_n[q];
    nsb = nsb_n; nrb = nrb_n;
    n_first = i + 2 < nst && st_n.first;
    DP_STAMP(2)
    // the rows fetched in this iteration have landed: every fetch is older than this iteration's two stores (and post loads)
    // (rows through registers: their ds_writes have waited for the loads - and with them for every older vector-memory operation, the
    // head rows' LDS-DMA pieces included)
    if constexpr (!DP_WSQ_ROWS_VIA_REGS) {
      if constexpr (POST == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (POST == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DP_STAMP(3)
    __builtin_amdgcn_s_barrier();
    DP_STAMP(4)
  }
  // ---- drain: the last step's epilogue
  static_for<0, 2>([&](auto tt) { epi_read((nst - 1) & 1, tt); });
  static_for<0, 2>([&](auto tt) { epi_store(tt); });
  if constexpr (DP_WSQ_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 5; ++k) p.dbg[(blockIdx.x * 4 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 5] = nst;
#pragma unroll
      for (int k = 8; k < 12; ++k) p.dbg[(blockIdx.x * 4 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 6] = __builtin_amdgcn_s_memtime() - t_loop0;
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
  }
#undef DP_STAMP
}


// =====================================================================================================
// The same structure on v_mfma_f32_16x16x32 (round 6, second form; profiles/r6_wsq_experiments.txt section 3). The chip holds ~2.0 GHz under
// this instruction and ~1.78 under 32x32x16, which cost conv3x3_wsq_kernel what its structure gained. One wave per SIMD, 512 registers:
//   * wave (g, h) holds couts [32 g, 32 g + 32) x channels [128 h, 128 h + 128) x 9 taps as 2 cout halves x 36 K planes of A fragments;
//   * a step is four output rows of a 16-pixel strip; B fragment (channel block, input row q, column tap) = 16 pixels x 32 channels, read ONCE
//     and used by every output row it feeds (kernel rows q - t in 0..2) and both cout halves: 72 fragment reads for 288 MFMAs
//     (conv3x3_wsr_kernel: 120 for 216); input rows visited in the order 2, 0, 3, 5, 1, 4 so that the rows that feed one output row (two
//     MFMAs per fragment) never sit next to each other in the stream - six fragments ahead then are at least 24 MFMAs ahead;
//   * the wave finishes the cout half c' = 0 of its four rows itself and hands c' = 1 to its partner; its A rows are permuted so that c' = 0
//     IS the half it keeps (A row r of half c' = cout 16 (c' ^ h) + r of the group): lane (pixel fr, quarter fq) stores couts 16 h + 4 fq + 0..3,
//     8 bytes per row, the four lanes of a pixel 32 contiguous bytes;
//   * ring pixels at pitch 544 (32-byte pad: chunk c of pixel p on 16-byte slot (2 p + c) mod 16, conflict-free for this fragment shape).
// =====================================================================================================
constexpr int kW1PP = 544;

template <typename T, bool RELU, int POST>
__global__ __launch_bounds__(256, 1) void conv3x3_ws1_kernel(const WsqArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int RP = kWqRP, PIX = kWqPix, PP = kW1PP, PPR = kWqPPR, ROWB = kWqRowB, NSLOT = kWqSlots, XCH = kWqXch;
  static_assert(18 * PP <= ROWB, "ring row");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int g = wave & 1, h = wave >> 1;            // cout group of 32, K half
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  int S = p.S;
  if (p.n_dev) S = min(max(*p.n_dev - p.n0, 0), p.N) * p.n_strips * p.spc;      // the live images only (uniform: a scalar load)
  const int s_begin = (int)((long long)S * pg / p.n_pg), s_end = (int)((long long)S * (pg + 1) / p.n_pg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;

  const int cgrp = slice * 64 + g * 32;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_post = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(POST ? p.post : p.in), 0, POST ? p.post_bytes : 0u, 0x00020000);

  const int frag_lane = fr * PP + h * 256 + fq * 16;
  const int n_cols = p.n_strips * 16;
  const int opix = p.cout * 2;
  const int ocb = (cgrp + 16 * h + 4 * fq) * 2;             // byte offset of this lane's 4 output channels inside a pixel

  auto advance = [&](WsqStep& st) __attribute__((always_inline)) {
    st.r += RP; st.um += RP; st.first = 0;
    if (st.r >= p.H) {
      st.r = 0; st.um += 2; st.first = 1; st.c0 += 16;
      if (st.c0 >= n_cols) { st.c0 = 0; st.n += 1; }
    }
    if (st.um >= NSLOT) st.um -= NSLOT;
  };
  unsigned dl[PPR];
  auto lanes_for_strip = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int pr = 0; pr < PPR; ++pr) {
      const int gi = pr * 64 + lane, px = gi / 34, sub = gi - px * 34;
      const bool ok = px < 18 && sub < 32 && (unsigned)(c0 - 1 + px) < (unsigned)p.W;
      dl[pr] = ok ? (unsigned)(px * PIX + sub * 16) : kLaneInv;
    }
  };
  auto row_base = [&](const WsqStep& st, int q, bool live) __attribute__((always_inline)) -> unsigned {
    const int row = st.r - 1 + q;
    return (live && (unsigned)row < (unsigned)p.H) ? (unsigned)(((st.n * p.H + row) * p.W + st.c0 - 1) * PIX) : kRowInv;
  };
  auto slot_of = [&](const WsqStep& st, int q) __attribute__((always_inline)) -> int {
    int s = st.um + q;
    if (s >= NSLOT) s -= NSLOT;
    return s;
  };
  auto fetch_piece = [&](int slot_bytes, unsigned rbase, auto prr) __attribute__((always_inline)) {
    constexpr int pr = decltype(prr)::value;
    if constexpr (!(DP_WSQ_EXP & 8))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + slot_bytes + pr * 1024), 16, (int)(dl[pr] + rbase), 0, 0, 0);
  };
  auto fetch_head_rows = [&](const WsqStep& st, bool live) __attribute__((always_inline)) {
    const int q = wave & 1;
    const int sb = slot_of(st, q) * ROWB;
    const unsigned rb = row_base(st, q, live);
    if (wave < 2) static_for<0, 5>([&](auto prr) { fetch_piece(sb, rb, prr); });
    else static_for<5, 10>([&](auto prr) { fetch_piece(sb, rb, prr); });
  };

  WsqStep st_c, st_n;
  {
    const int colid = s_begin / p.spc, k = s_begin - colid * p.spc;
    st_c.n = colid / p.n_strips;
    st_c.c0 = (colid - st_c.n * p.n_strips) * 16;
    st_c.r = k * RP;
    st_c.um = 0;
    st_c.first = 1;
  }
  lanes_for_strip(st_c.c0);
  fetch_head_rows(st_c, true);
  {
    const int sb = slot_of(st_c, 2 + wave) * ROWB;
    const unsigned rb = row_base(st_c, 2 + wave, true);
    static_for<0, PPR>([&](auto prr) { fetch_piece(sb, rb, prr); });
  }
  // ---- the weights AFTER the first step's row fetches are in flight (both come out of L2; the rows land while the 72 fragment loads run)
  u32x4 wq[72];       // [cout half c'][plane s = channel block x tap]
  {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int L = cgrp + 16 * (c ^ h) + fr, l64 = L & 63;
      const int rem = l64 & 31;
      const int phys = (L & ~63) + (((l64 >> 5) * 2 + ((rem >> 2) & 1)) * 16) + (rem >> 3) * 4 + (rem & 3);   // pack.py's row permutation
      const unsigned char* __restrict__ w = reinterpret_cast<const unsigned char*>(p.w) + dp_wtile_off(phys, h * 36, fq, 72);
#pragma unroll
      for (int s = 0; s < 36; ++s) wq[c * 36 + s] = *reinterpret_cast<const u32x4*>(w + s * 1024);
    }
#pragma unroll
    for (int s = 0; s < 72; ++s) {      // register classes pinned once (see conv3x3_wsq_kernel)
      if (s < kWqAgprFrags) asm volatile("" : "+a"(wq[s]));
      else asm volatile("" : "+v"(wq[s]));
    }
  }
  // initial accumulator values: the bias in the h = 0 wave (half c', register e = cout 16 c' + 4 fq + e of the group), zero in the other
  f32x4 binit[2];
  {
    const float m = h == 0 ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) binit[c] = *reinterpret_cast<const f32x4*>(p.bias + cgrp + 16 * c + 4 * fq) * m;
  }

  st_n = st_c;
  advance(st_n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- pending epilogue (the step computed in the previous iteration): the kept cout half of the four rows, where they go
  f32x4 eacc[4];
  int e_off = 0, e_rows = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) eacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // post tensor values of the pending step. POST = 1: the lane's 4 channels at its pixel of each row. POST = 2: the FOUR half-size rows
  // b' .. b' + 3, b' = r / 2 - 1 (clamped to the map), at the lane's two source columns. A step's first output row r is a multiple of 4,
  // so output row r + t always interpolates between the rows (0, 1), (1, 2), (1, 2), (2, 3) of those four (ATen's align_corners = False
  // source index: (o + 0.5) / 2 - 0.5, clamped at 0; at r = 0 the clamped row -1 IS row 0 and its weight is ly = 0 or the same row's).
  constexpr int NPV = POST == 1 ? 4 : (POST == 2 ? 8 : 1);
  constexpr int NPL = POST == 1 ? 4 : (POST == 2 ? 8 : 0);
  u32x2w pv[NPV];
#pragma unroll
  for (int k = 0; k < NPV; ++k) pv[k] = u32x2w{0u, 0u};
  float p_lx = 0.f, p_ly[4] = {0.f, 0.f, 0.f, 0.f};
  auto post_issue = [&](const WsqStep& st) __attribute__((always_inline)) {
    const int col = st.c0 + fr;
    if constexpr (POST == 1) {
      const int off = ((st.n * p.H + st.r) * p.W + col) * opix + ocb;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        pv[t] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, (st.r + t < p.H && col < p.W) ? off + t * (p.W * opix) : (int)kLaneInv, 0, 0);
    } else if constexpr (POST == 2) {
      int x0, x1;
      wq_bil_src(min(col, p.W - 1), p.Wl, x0, x1, p_lx);
      const int bq = (st.r >> 1) - 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = min(max(bq + k, 0), p.Hl - 1);
        const int off = ((st.n * p.Hl + row) * p.Wl) * opix + ocb;
        pv[2 * k] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, off + x0 * opix, 0, 0);
        pv[2 * k + 1] = __builtin_amdgcn_raw_buffer_load_b64(rs_post, off + x1 * opix, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int y0, y1;
        wq_bil_src(st.r + t, p.Hl, y0, y1, p_ly[t]);       // wave-uniform; only the weight is used
      }
    }
  };
  // POST = 2: the horizontal half of the interpolation once per half-size row
  float hrow[POST == 2 ? 4 : 1][4];
  auto epi_prepare = [&]() __attribute__((always_inline)) {
    if constexpr (POST == 2) {
      const float hx = 1.f - p_lx;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int sh = (e & 1) * 16;
          const float fa = Elem<T>::unpack((pv[2 * k][e >> 1] >> sh) & 0xffffu), fb = Elem<T>::unpack((pv[2 * k + 1][e >> 1] >> sh) & 0xffffu);
          hrow[k][e] = hx * fa + p_lx * fb;
        }
    }
  };
  f32x4 eo[4];
  auto epi_read = [&](int par, auto tt) __attribute__((always_inline)) {
    constexpr int t = decltype(tt)::value;
    eo[t] = *reinterpret_cast<const f32x4*>(smem + XCH + ((((wave ^ 2) * 2 + par) * 4 + t)) * 1024 + lane * 16);
  };
  auto epi_store = [&](auto tt) __attribute__((always_inline)) {
    constexpr int t = decltype(tt)::value;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = eacc[t][e] + eo[t][e];
      if constexpr (RELU) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (POST == 1) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v[2 * e] += Elem<T>::unpack(pv[t][e] & 0xffffu);
        v[2 * e + 1] += Elem<T>::unpack(pv[t][e] >> 16);
      }
    } else if constexpr (POST == 2) {
      constexpr int i = t == 0 ? 0 : (t == 3 ? 2 : 1), j = i + 1;     // rows of hrow this output row interpolates between
      const float hy = 1.f - p_ly[t];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += hy * hrow[i][e] + p_ly[t] * hrow[j][e];      // ATen's order: hy * (hx * a + lx * b) + ly * (hx * c + lx * d)
    }
    const u32x2w pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])};
    __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, t < e_rows ? e_off + t * (p.W * opix) : (int)kLaneInv, 0, 0);
  };

  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(k) if constexpr (DP_WSQ_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  unsigned long long tl = (DP_WSQ_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long t_loop0 = tl, rt0 = (DP_WSQ_EXP & 16) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // ---- loop state, one step ahead of its use (computed inside the previous step's matrix loop)
  auto frag_bases = [&](const WsqStep& st, int (&v)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = slot_of(st, q) * ROWB + frag_lane;
  };
  int va[6], va_n[6];
  frag_bases(st_c, va);
  int nsb = slot_of(st_n, 2 + wave) * ROWB, nsb_n = 0;
  unsigned nrb = row_base(st_n, 2 + wave, nst > 1), nrb_n = 0;
  bool n_first = nst > 1 && st_n.first;
  constexpr int NF = 72, AHEAD = DP_WSQ_AHEAD, DMA_GAP = 3;
  static_assert(1 + DMA_GAP * (kWqPPR - 1) < 32, "the last LDS-DMA piece of a step is issued before its first store: vmcnt(stores + post loads) then covers every piece");
  // fragment f = (channel block cbl of 32, position in the row order, column tap dx)
  constexpr int kRowOrder[6] = {2, 0, 3, 5, 1, 4};

  for (int i = 0; i < nst; ++i) {
    const int par = i & 1;
    f32x4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t][0] = binit[0]; acc[t][1] = binit[1]; }
    auto frag = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
      constexpr int f = decltype(ff)::value;
      constexpr int cbl = f / 18, q = kRowOrder[(f % 18) / 3], dx = f % 3;
      return *reinterpret_cast<const u32x4*>(smem + va[q] + (dx * PP + cbl * 64));
    };
    u32x4 bf[AHEAD + 1];
    static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag(ff); });
    __builtin_amdgcn_sched_barrier(0);
    DP_STAMP(0)
    static_for<0, NF>([&](auto ff) {
      constexpr int f = decltype(ff)::value;
      constexpr int cbl = f / 18, q = kRowOrder[(f % 18) / 3], dx = f % 3;
      if constexpr (f + AHEAD < NF && !(DP_WSQ_EXP & 1)) bf[(f + AHEAD) % (AHEAD + 1)] = frag(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DP_WSQ_EXP & 32) && f % 18 == 0 && f > 0) { DP_STAMP(7 + f / 18) }
      // ---- side work in the MFMAs' shadow
      if constexpr (f == 0) {               // the next step starts a column: lane offsets of its strip, its two head rows
        if (n_first) {
          lanes_for_strip(st_n.c0);
          fetch_head_rows(st_n, true);
        }
      }
      if constexpr (f >= 1 && f < 1 + DMA_GAP * PPR && (f - 1) % DMA_GAP == 0) fetch_piece(nsb, nrb, std::integral_constant<int, (f >= 1 && f < 1 + DMA_GAP * PPR ? (f - 1) / DMA_GAP : 0)>{});
      if constexpr (f == 32) { epi_read(par ^ 1, std::integral_constant<int, 0>{}); epi_read(par ^ 1, std::integral_constant<int, 1>{}); }
      if constexpr (f == 33) { epi_read(par ^ 1, std::integral_constant<int, 2>{}); epi_read(par ^ 1, std::integral_constant<int, 3>{}); }
      if constexpr (f == 36) epi_prepare();
      if constexpr (f == 38) epi_store(std::integral_constant<int, 0>{});
      if constexpr (f == 42) epi_store(std::integral_constant<int, 1>{});
      if constexpr (f == 46) epi_store(std::integral_constant<int, 2>{});
      if constexpr (f == 50) epi_store(std::integral_constant<int, 3>{});
      if constexpr (f == 53 && POST != 0) post_issue(st_c);
      if constexpr (f == 54) {              // where this step's outputs go (its epilogue runs inside the next step)
        const int col = st_c.c0 + fr;
        e_off = ((st_c.n * p.H + st_c.r) * p.W + col) * opix + ocb;
        e_rows = col < p.W ? p.H - st_c.r : 0;
      }
      if constexpr (f == 56) { st_c = st_n; advance(st_n); }
      if constexpr (f == 58) frag_bases(st_c, va_n);
      if constexpr (f == 60) {
        nsb_n = slot_of(st_n, 2 + wave) * ROWB;
        nrb_n = row_base(st_n, 2 + wave, i + 2 < nst);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- input row q feeds output row t with kernel row q - t, both cout halves
      static_for<0, 4>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        if constexpr (q - t >= 0 && q - t <= 2) {
          static_for<0, 2>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (!(DP_WSQ_EXP & 4)) Mma<T>::run(wq[c * 36 + cbl * 9 + (q - t) * 3 + dx], bf[f % (AHEAD + 1)], acc[t][c]);
            else acc[t][c][0] += __builtin_bit_cast(float, bf[f % (AHEAD + 1)][0]) * __builtin_bit_cast(float, wq[c * 36 + cbl * 9 + (q - t) * 3 + dx][0]);
          });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DP_WSQ_EXP & 32) { DP_STAMP(11) } else { DP_STAMP(1) }
    // ---- hand cout half 1 of the four rows to the partner, keep half 0 for the epilogue that runs inside the next step
    {
      unsigned char* xb = smem + XCH + ((wave * 2 + par) * 4) * 1024 + lane * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        *reinterpret_cast<f32x4*>(xb + t * 1024) = acc[t][1];
        eacc[t] = acc[t][0];
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) va[q] = va_n[q];
    nsb = nsb_n; nrb = nrb_n;
    n_first = i + 2 < nst && st_n.first;
    DP_STAMP(2)
    // the rows fetched in this iteration have landed: every fetch is older than this iteration's four stores (and post loads)
    if constexpr (NPL == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NPL == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DP_STAMP(3)
    __builtin_amdgcn_s_barrier();
    DP_STAMP(4)
  }
  // ---- drain: the last step's epilogue
  static_for<0, 4>([&](auto tt) { epi_read((nst - 1) & 1, tt); });
  epi_prepare();
  static_for<0, 4>([&](auto tt) { epi_store(tt); });
  if constexpr (DP_WSQ_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 5; ++k) p.dbg[(blockIdx.x * 4 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 5] = nst;
#pragma unroll
      for (int k = 8; k < 12; ++k) p.dbg[(blockIdx.x * 4 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 6] = __builtin_amdgcn_s_memtime() - t_loop0;
      p.dbg[(blockIdx.x * 4 + wave) * 16 + 7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
  }
#undef DP_STAMP
}

template <typename T, bool RELU, int POST, int SH>
int launch_wsq_r(WsqArgs a, int over, hipStream_t stream) {
  static_assert(kWqLds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    if constexpr (SH == 32) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wsq_kernel<T, RELU, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, kWqLds);
    else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws1_kernel<T, RELU, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, kWqLds);
    cus = wq_num_cus();
    attr_set = true;
  }
  a.n_strips = (a.W + 15) / 16;
  a.spc = (a.H + kWqRP - 1) / kWqRP;
  a.S = a.N * a.n_strips * a.spc;
  a.n_slices = a.cout / 64;
  int groups = cus / (8 * a.n_slices);
  if (over < 0) { groups += over; over = 1; }    // shared_chip = 2: -over groups of 8 * n_slices CUs are left to the other stream
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8 * (over > 1 ? over : 1);
  a.dbg = nullptr;
#if DP_WSQ_EXP & 16
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_pg * a.n_slices;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 16 * 4 * 4096);
  a.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 16 * 4 * nblk, stream);
#endif
  if constexpr (SH == 32) hipLaunchKernelGGL((conv3x3_wsq_kernel<T, RELU, POST>), dim3(a.n_pg * a.n_slices), dim3(256), kWqLds, stream, a);
  else hipLaunchKernelGGL((conv3x3_ws1_kernel<T, RELU, POST>), dim3(a.n_pg * a.n_slices), dim3(256), kWqLds, stream, a);
#if DP_WSQ_EXP & 16
  {
    static int shown = 0;
    if (shown++ == 4) {   // a warm launch
      (void)hipStreamSynchronize(stream);
      unsigned long long* hbuf = (unsigned long long*)malloc(sizeof(unsigned long long) * 64 * nblk);
      (void)hipMemcpy(hbuf, dbg, sizeof(unsigned long long) * 64 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w) {
        double sum[5] = {0, 0, 0, 0, 0}, n = 0, cyc = 0, rt = 0;
        for (int b = 0; b < nblk; ++b) {
          for (int k = 0; k < 5; ++k) sum[k] += (double)hbuf[(b * 4 + w) * 16 + k];
          n += (double)hbuf[(b * 4 + w) * 16 + 5]; cyc += (double)hbuf[(b * 4 + w) * 16 + 6]; rt += (double)hbuf[(b * 4 + w) * 16 + 7];
        }
        if (DP_WSQ_EXP & 32) {
          double q4[4] = {0, 0, 0, 0};
          for (int b = 0; b < nblk; ++b) for (int k = 0; k < 4; ++k) q4[k] += (double)hbuf[(b * 4 + w) * 16 + 8 + k];
          fprintf(stderr, "wsq wave %d: loop quarters (1152 pipe cycles each): %.0f %.0f %.0f %.0f\n", w, q4[0] / n, q4[1] / n, q4[2] / n, q4[3] / n);
        }
        fprintf(stderr, "wsq wave %d: per step cycles: top %.0f  loop %.0f  hand-over %.0f  waits %.0f  barrier %.0f  (steps/wg %.1f, in-kernel clock %.2f GHz)\n", w,
                sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, n / nblk, cyc / (rt * 10.0));
      }
      free(hbuf);
    }
  }
#endif
  return dp_check_launch("conv3x3_wsq_kernel");
}

template <typename T, int SH>
int launch_wsq(const WsqArgs& a, int over, hipStream_t stream, int relu, int post_mode) {
  if (post_mode == 1 && relu) return launch_wsq_r<T, true, 1, SH>(a, over, stream);
  if (post_mode == 2 && relu) return launch_wsq_r<T, true, 2, SH>(a, over, stream);
  if (post_mode != 0) return dp_fail(DP_ERR_UNSUPPORTED, "conv3x3_wsq_kernel: post_res needs ReLU");
  return relu ? launch_wsq_r<T, true, 0, SH>(a, over, stream) : launch_wsq_r<T, false, 0, SH>(a, over, stream);
}

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): is this launch one of the 256 -> 256 3x3 / pad 1 / stride 1 layers the kernel is written for?
// The answer depends on the layer and the per-image geometry only - never on N.
bool dp_conv_wsq_ok(const dp_conv_params* p) {
  const DpPolicy& pol = dp_policy();
  if (pol.conv_wsq == 0 || p->ring_order) return false;
  const bool shape = p->Cin == 256 && p->Cout == 256 && p->Cout_w == 256;
  return (p->dtype == DP_BF16 || p->dtype == DP_F16) && shape && p->ntaps == 9 && p->Kpad == 9 * p->Cin && p->stride == 1 &&
         (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == -1 && p->wi_off == -1 && p->H == p->Ho && p->W == p->Wo &&
         !p->residual && !p->out_f32 && !p->head_out && !p->in2 && p->n_groups <= 1 && p->out && p->osW == p->Cout &&
         p->osH == (long long)p->W * p->Cout && p->osN == (long long)p->H * p->W * p->Cout &&
         (long long)p->H * p->W >= pol.wsq_min_hw && (long long)p->H * p->W * 512 <= 0x3fff0000ll &&
         (((uintptr_t)p->in | (uintptr_t)p->out | (uintptr_t)p->weight | (uintptr_t)p->post_res | (uintptr_t)p->bias) & 15) == 0 &&
         (p->post_res == nullptr ? p->post_mode == 0 : (p->relu && (p->post_mode == 1 || (p->post_mode == 2 && p->H % 2 == 0 && p->W % 2 == 0))));
}

int dp_conv_wsq_launch(const dp_conv_params* p, dp_stream_t stream) {
  const DpPolicy& pol = dp_policy();
  int over;
  if (p->shared_chip == 2) over = -(int)pol.ws_reserve;
  else over = (int)(p->shared_chip ? pol.ws_over_shared : pol.ws_over_alone);
  if (over == 0) over = 1;
  hipStream_t s = as_stream(stream);
  const int pm = p->post_res ? p->post_mode : 0;
  // 32-bit buffer offsets: a batch whose tensors pass the range goes image chunk by image chunk (images are independent; the
  // per-pixel arithmetic does not know about the chunking)
  const long long per_img = (long long)p->H * p->W * 512;
  long long lim = pol.rows_chunk_bytes < 0x7fff0000ll ? pol.rows_chunk_bytes : 0x7fff0000ll;
  int per = (int)(lim / per_img);
  if (per < 1) per = 1;
  for (int n0 = 0; n0 < p->N; n0 += per) {
    const int n = p->N - n0 < per ? p->N - n0 : per;
    WsqArgs a;
    a.in = reinterpret_cast<const unsigned char*>(p->in) + (long long)n0 * per_img;
    a.w = p->weight; a.bias = p->bias;
    a.out = reinterpret_cast<unsigned char*>(p->out) + (long long)n0 * per_img;
    a.N = n; a.H = p->H; a.W = p->W; a.cout = p->Cout;
    a.n_strips = a.spc = a.n_slices = a.n_pg = a.S = 0;
    a.dbg = nullptr;
    a.n_dev = p->n_dev; a.n0 = n0;
    a.Hl = p->H / 2; a.Wl = p->W / 2;
    const long long post_img = pm == 2 ? (long long)a.Hl * a.Wl * 512 : per_img;
    a.post = p->post_res ? reinterpret_cast<const unsigned char*>(p->post_res) + (long long)n0 * post_img : nullptr;
    a.post_bytes = p->post_res ? (unsigned)(n * post_img) : 0u;
    a.in_bytes = (unsigned)(n * per_img);
    a.out_bytes = (unsigned)(n * per_img);
    int rc;
    if (pol.wsq_shape == 32) rc = p->dtype == DP_BF16 ? launch_wsq<uint16_t, 32>(a, over, s, p->relu, pm) : launch_wsq<f16_t, 32>(a, over, s, p->relu, pm);
    else rc = p->dtype == DP_BF16 ? launch_wsq<uint16_t, 16>(a, over, s, p->relu, pm) : launch_wsq<f16_t, 16>(a, over, s, p->relu, pm);
    if (rc != DP_OK) return rc;
  }
  return DP_OK;
}
