// Weight-stationary POINTWISE convolutions with a long K axis (16-bit storage): kernel class 9 behind dp_conv2d_nhwc (dp_conv.hip).
//
// Layers: the 1x1 convolutions of the ResNet trunk and the FPN whose K is 512 / 1024 / 2048 channels on few pixels
// (/root/reference/detectron2/modeling/backbone/resnet.py:192-193 conv1 of res4 / res5, :199-205 conv3 (+ residual, ReLU) of res5,
// fpn.py:140-157 lateral convolutions (+ nearest x2 of the top-down map), roi_heads/box_head.py:71-73 fc2 as a 1x1 layer).
// On the LDS-ring kernels such a layer re-stages BOTH operands per tile: a 160 x 256 tile of res4's conv1 (K = 1024) pulls 512 KiB of
// weights and 320 KiB of pixels through one CU's 64 B/clk fill path, 3.3 tiles per CU - 2.7 MB per CU where the layer's operands are
// 0.8 MB per CU when nothing is fetched twice (profiles/r4_layers.txt: 33 us = neither the MFMA nor the HBM roof).
//
// Here a workgroup keeps a CW-cout x ALL-K slice of the weights (256 KiB = the 8 waves' 128 weight VGPRs per lane) in the register file
// for the whole launch and streams ITS pixels through LDS exactly once:
//   * K = SK x 512 channels. Wave (ks, cs) owns K slice ks (512 channels = 16 MFMA K steps) of cout sub-slice cs (32 couts = 2 MFMA
//     row tiles): 32 A fragments, loaded once from the 1 KiB weight tiles. SK = 1: CW = 256 couts, SK = 2: 128, SK = 4: 64.
//   * a step is P = 32 consecutive pixels (SK = 4: 16 - a stage is P x K x 2 B and two of them must fit). Pixels are independent
//     (1 tap, stride 1), so M = N H W is one flat axis; grid = (cout slice) x (pixel group), the slices of a pixel group on one XCD
//     (they read the same pixels: the second reader hits that XCD's L2), each workgroup owns a contiguous range of steps.
//   * a stage holds the step's pixel rows split by K slice: [ks][pixel][1 KiB], filled by LDS-DMA one step ahead (one piece = one
//     pixel's slice: 1 KiB contiguous on both sides). Chunk X (16 bytes) of pixel q sits at position X ^ (q & 15) - the XOR is applied
//     to the per-lane SOURCE offset of the DMA - which makes the B-fragment ds_read_b128 (16 pixels x 4 chunks) bank-conflict free
//     at a pitch of exactly 1 KiB: SK = 2 needs all 160 KiB (2 stages x 64 KiB + the exchange buffers), a padded pitch would not fit.
//   * SK > 1: the SK waves of a cout sub-slice hold partial sums over their K slices. Each OUTPUT block (16 pixels x 16 couts) has one
//     owner among them; the others write their partial block to an exchange buffer (double-buffered by step parity), and after the
//     step's barrier the owner adds the partials IN K-SLICE ORDER - a fixed order, so a pixel's bits do not depend on the batch, the
//     step or the workgroup it lands in - then bias, residual, ReLU, one rounding, 16-byte (SK = 4: 8-byte) stores from registers.
//   * the epilogue of step i runs at the top of iteration i + 1 (WSR's deferral, dp_conv_ws.hip): LDS-DMA issue first (before the wave
//     has ds_reads in flight), then the exchange reads and stores under the DMA's latency, then a loop of nothing but reads and MFMAs.
// One barrier per step. Summation order: one MFMA chain per K slice in channel order, slices added in order - not the ring kernels'
// single chain, so a layer runs here for EVERY batch size or never (the choice looks at the channel counts only).
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"
#include <stdlib.h>

#ifndef DP_PWS_OPT
#define DP_PWS_OPT 0     // schedule experiments (results exact): 2 ten fragments ahead
#endif
#ifndef DP_PWS_EXP
#define DP_PWS_EXP 0     // diagnostic builds (tools/build_variant.sh): 1 no LDS-DMA in the loop, 4 no MFMAs, 8 no fragment reads (timing only,
#endif                   // results garbage), 16 in-kernel phase stamps (results exact)

namespace {

int pws_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

struct PwsArgs {
  const void* in;
  const void* w;
  const float* bias;
  const void* res;
  void* out;
  int M, cout, relu, kpad;
  int n_slices, n_pg, S;
  unsigned in_bytes, out_bytes, res_bytes;
  int res_up;               // residual read through a nearest x2 up-sampling (fpn.py:152): element (n, ho >> 1, wo >> 1)
  int HoWo, Wo;
  int rsN, rsH, rsW;        // residual strides in elements (res_up)
  unsigned long long* dbg;  // diagnostic builds (-DDP_PWS_EXP=16): per-wave phase cycle sums
};

template <typename T, int SK>
struct PwsCfg {
  static constexpr int NCS = 8 / SK;                 // cout sub-slices (32 couts) per workgroup
  static constexpr int CW = NCS * 32;                // couts per workgroup
  static constexpr int PT = SK == 4 ? 1 : 2;         // pixel tiles per step
  static constexpr int P = 16 * PT;
  static constexpr int SLICE_B = P * 1024;           // one K slice of a stage
  static constexpr int STAGE_B = SK * SLICE_B;
  static constexpr int NBLK = PT * 2;                // output blocks (pixel tile, cout tile) per wave
  static constexpr int XCH_B = SK == 1 ? 0 : NCS * NBLK * (SK - 1) * 1024;   // exchange buffer of one parity
  static constexpr int NPW = P * SK / 8;             // DMA pieces per wave and step
  static constexpr int WSCR = STAGE_B;               // weight prologue scratch: 8 x 8 KiB behind stage 0
  // SKEW = the two halves of the workgroup half a step apart (two barriers per step); SK = 1 then needs a third stage (both halves
  // read the same slice, half a step apart)
  static constexpr int nstage(bool skew) { return (skew && SK == 1) ? 3 : 2; }
  static constexpr int lds(bool skew) {
    const int v = nstage(skew) * STAGE_B + 2 * XCH_B;
    return v < WSCR + 8 * 8192 ? WSCR + 8 * 8192 : v;
  }
};

template <typename T, int SK, bool HAS_RES, bool SKEW>
__global__ __launch_bounds__(512, 2) void conv1x1_pws_kernel(const PwsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(SK == 1 || SK == 2 || SK == 4, "K = 512 / 1024 / 2048 channels");
  using Cfg = PwsCfg<T, SK>;
  constexpr int NCS = Cfg::NCS, CW = Cfg::CW, PT = Cfg::PT, P = Cfg::P, SLICE_B = Cfg::SLICE_B, STAGE_B = Cfg::STAGE_B, NSTAGE = Cfg::nstage(SKEW);
  constexpr int NBLK = Cfg::NBLK, XCH_B = Cfg::XCH_B, NPW = Cfg::NPW;
  constexpr int KB = SK * 1024;                      // bytes of a pixel's channel row
  constexpr int XCH0 = NSTAGE * STAGE_B;
  constexpr int OOB = (int)0x80000000;
  static_assert(Cfg::lds(SKEW) <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(k) if constexpr (DP_PWS_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  unsigned long long tl = (DP_PWS_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cs = wave % NCS, ks = wave / NCS;
  // SKEW: waves w and w + 4 share a SIMD, and the two halves of the workgroup run half a step apart: while waves 0 .. 3 ("X") are in
  // their memory phase, waves 4 .. 7 ("Y") are in their matrix phase, and the other way round - two barriers per step. With SK > 1 the
  // halves hold different K slices (X the lower ones), so every stage slice is filled and read by waves of ONE half.
  const bool isY = SKEW && wave >= 4;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));
  const int s_begin = (int)((long long)p.S * pg / p.n_pg), s_end = (int)((long long)p.S * (pg + 1) / p.n_pg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  constexpr bool has_res = HAS_RES;
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(has_res ? p.res : p.in), 0, has_res ? p.res_bytes : 0u, 0x00020000);

  // ---- pixel rows of step s -> stage s % NSTAGE. Piece = (K slice kp, pixel q) = 1 KiB; wave w issues pieces [w, w + 1) * NPW, so half
  //      X (waves 0 .. 3) fills the lower K slices - with SK > 1 exactly the slices its own waves read
  auto issue_piece = [&](int s, auto jj) __attribute__((always_inline)) {
    constexpr int j = decltype(jj)::value;
    const int m0 = s < nst ? (s_begin + s) * P : p.M;      // past the workgroup's range: every lane out of range, nothing moves
    const int pc = wave * NPW + j;
    const int kp = pc / P, q = pc % P;                 // wave-uniform
    const int m = m0 + q;
    const int off = m < p.M ? m * KB + kp * 1024 + ((lane ^ (q & 15)) << 4) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + (s % NSTAGE) * STAGE_B + kp * SLICE_B + q * 1024), 16, off, 0, 0, 0);
  };
  auto issue_stage = [&](int s) __attribute__((always_inline)) { static_for<0, NPW>([&](auto jj) { issue_piece(s, jj); }); };
  issue_stage(0);

  // ---- this wave's 32 couts x its 512 K values, in registers for the whole launch. A row fr of MFMA tile t = logical cout
  //      cbase + (fr >> 2) * 8 + t * 4 + (fr & 3): a lane's 8 accumulator values of a pixel are 8 CONSECUTIVE channels (one 16-byte
  //      store), and in pack.py's row permutation those 16 rows are exactly physical tile (cbase / 32) * 2 + t of the packed matrix.
  //      The 32 tiles (1 KiB each, contiguous) come through LDS: whole-line LDS-DMA copies (lane l fetches chunk (l & 3) ^ swz(l >> 2) of
  //      row l >> 2: the ring kernels' conflict-free image) into an 8 KiB scratch per wave behind stage 0, 4 tiles per round, two rounds
  //      in flight. Cycles per workgroup until the weights are in place (round-5 phase stamps, res4's conv1): this form 9 900; the
  //      fragment layout loaded directly (16 rows x 16 B per quarter wave = 8 cache lines for 256 bytes) 12 400; coalesced 1 KiB register
  //      loads + a lane transpose by ds_bpermute_b32 11 600. All of them are bound by what an XCD's L2 delivers when its 32 CUs pull
  //      256 KiB each at the same moment (8 MB per XCD and launch - the price of keeping the weights in every CU's register file).
  const int cbase = slice * CW + cs * 32;
  u32x4 wfr[32];
  {
    const int n_planes = p.kpad * 2 / 64;
    unsigned char* const scr = smem + Cfg::WSCR + wave * 8192;
    const int lrow = lane >> 2;
    const int goff = lrow * 64 + (((lane & 3) ^ swz(lrow)) << 4);
    const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(p.w) + goff;
    const unsigned char* const rd = scr + fr * 64 + ((fq ^ swz(fr)) << 4);
    auto issue_round = [&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = r * 4 + j, t = k >> 4, c = k & 15;
        const long long tile = (long long)((cbase >> 5) * 2 + t) * n_planes + ks * 16 + c;
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(wsrc + tile * 1024), DP_LDS_PTR(scr + (r & 1) * 4096 + j * 1024), 16, 0, 0);
      }
    };
    auto read_round = [&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) wfr[r * 4 + j] = *reinterpret_cast<const u32x4*>(rd + (r & 1) * 4096 + j * 1024);
    };
    issue_round(std::integral_constant<int, 0>{});
    issue_round(std::integral_constant<int, 1>{});
    static_for<0, 8>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      if constexpr (r < 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      read_round(rr);
      if constexpr (r + 2 < 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the region is read before it is refilled
        issue_round(std::integral_constant<int, (r + 2 < 8 ? r + 2 : 0)>{});
      }
    });
  }

  // ---- output ownership: which wave of a cout sub-slice adds the K slices of a block (pixel tile pt, cout tile t) and stores it
  //      SK = 1: every block is its wave's own.
  //      lockstep: SK = 2: K slice pt owns pixel tile pt. SK = 4 (one pixel tile): K slice t owns cout tile t, slices 2 / 3 only send.
  //      SKEW:     the Y half finalizes - SK = 2: K slice 1 owns all four blocks. SK = 4: K slice 2 + t owns cout tile t.
  const bool owner = SK == 1 || (SKEW ? isY : (SK == 2 || ks < 2));
  const int o_t = SK == 4 ? (ks & 1) : 0;
  constexpr int NOWN = (SK == 1 || (SK == 2 && SKEW)) ? 2 : 1;   // 16-pixel rows of output an owner finalizes per step
  constexpr int NV = SK == 4 ? 4 : 8;                             // channels per lane and row
  auto own_pt = [&](int r) __attribute__((always_inline)) -> int { return (SK == 2 && !SKEW) ? ks : r; };   // pixel tile of owned row r
  const int cown = SK == 4 ? cbase + fq * 8 + o_t * 4 : cbase + fq * 8;
  // the bias is the initial value of K slice 0's accumulators (tile t, rows 4 fq .. 4 fq + 3 = couts cbase + fq * 8 + t * 4 + e): the
  // epilogue has no bias add
  f32x4 bias_t[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bias_t[t] = ks == 0 ? *reinterpret_cast<const f32x4*>(p.bias + cbase + fq * 8 + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int opix = p.cout * 2;
  auto res_row_off = [&](int m) __attribute__((always_inline)) -> int {   // byte offset of pixel m's residual row
    if (m >= p.M) return OOB;
    if (!p.res_up) return m * opix;
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    return (n * p.rsN + (ho >> 1) * p.rsH + (wo >> 1) * p.rsW) * 2;
  };

  // fragment addresses: pixel fr of tile pt, K step c of this wave's slice: stage + ks * SLICE_B + pt * 16 KiB + (c >> 2) * 256 + fo[c & 3]
  int fo[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) fo[cl] = ks * SLICE_B + fr * 1024 + ((cl ^ (fr >> 2)) << 6) + ((fq ^ (fr & 3)) << 4);

  // exchange slots of one parity: [cs][block = pt * 2 + t][sender = K slice, skipping the owner's][lane * 16]
  auto xslot = [&](int par, int blk, int sender, int own_ks) __attribute__((always_inline)) -> unsigned char* {
    const int j = sender < own_ks ? sender : sender - 1;
    return smem + XCH0 + par * XCH_B + ((cs * NBLK + blk) * (SK - 1) + j) * 1024 + lane * 16;
  };
  auto blk_owner = [&](int pt, int t) __attribute__((always_inline)) -> int {     // K slice that owns block (pt, t), SK > 1
    if constexpr (SK == 2) return SKEW ? 1 : pt;
    else return SKEW ? 2 + t : t;
  };

  f32x4 acc[PT][2];             // the step's blocks: live from the matrix phase to the owner's next memory phase
  u32x4 rv[NOWN];               // residual values of the step that is finalized next
#pragma unroll
  for (int r = 0; r < NOWN; ++r) rv[r] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) acc[pt][0] = acc[pt][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue of step s (s < 0: stores masked): K slices added in slice order (slice 0 carries the bias), residual, ReLU, one rounding
  auto finalize = [&](int s) __attribute__((always_inline)) {
    const int par = s & 1;
    const int m0 = (s_begin + s) * P;
#pragma unroll
    for (int r = 0; r < NOWN; ++r) {
      const int pt = SK == 4 ? 0 : own_pt(r);
      float v[NV];
      if constexpr (SK == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = acc[r][0][k]; v[4 + k] = acc[r][1][k]; }
      } else if constexpr (SK == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(xslot(par, pt * 2 + t, 1 - ks, ks));
          const f32x4 mine = pt == 0 ? acc[0][t] : acc[1][t];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[t * 4 + k] = o[k] + mine[k];     // two slices: a + b is commutative bit for bit
        }
      } else {
        f32x4 sum;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
          if (q == ks) o = o_t == 0 ? acc[0][0] : acc[0][1];
          else o = *reinterpret_cast<const f32x4*>(xslot(par, o_t, q, ks));
          if (q == 0) sum = o;
          else {
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] += o[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = sum[k];
      }
      if constexpr (has_res) {
#pragma unroll
        for (int k = 0; k < NV / 2; ++k) {
          v[2 * k] += Elem<T>::unpack(rv[r][k] & 0xffffu);
          v[2 * k + 1] += Elem<T>::unpack(rv[r][k] >> 16);
        }
      }
      if (p.relu) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      const int m = m0 + pt * 16 + fr;
      const int off = (s >= 0 && m < p.M) ? m * opix + cown * 2 : OOB;
      if constexpr (SK == 4) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, off, 0, 0);
      } else {
        const u32x4 pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, off, 0, 0);
      }
    }
  };
  auto res_issue = [&](int s) __attribute__((always_inline)) {
    const int m0 = s < nst ? (s_begin + s) * P : p.M;
#pragma unroll
    for (int r = 0; r < NOWN; ++r) {
      const int pt = SK == 4 ? 0 : own_pt(r);
      const int ro = res_row_off(m0 + pt * 16 + fr);
      const int off = ro == OOB ? OOB : ro + cown * 2;
      if constexpr (SK == 4) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(rs_res, off, 0, 0);
        rv[r] = u32x4{x[0], x[1], 0u, 0u};
      } else {
        rv[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
      }
    }
  };

  // memory phase of iteration i (SKEW): the pieces of step i + 1 FIRST (before this wave has LDS reads in flight; unconditional - a step
  // past the range reads nothing - so that the compiler can count the vector-memory operations between a residual load and its use), then
  // the owner's epilogue of the step whose matrix phase it finished last (X, SK = 1: step i - 1 from the previous iteration; Y: step
  // i - 1 from the first half of this one), then the residual values of the step it finalizes next
  auto mem_phase = [&](int i) __attribute__((always_inline)) {
    if constexpr (!(DP_PWS_EXP & 1)) issue_stage(i + 1);
    asm volatile("" ::: "memory");
    DP_STAMP(7)
    if (owner) {
      finalize(i - 1);
      if constexpr (has_res) res_issue(i);
    }
  };
  // SKEW: the first AHEAD fragments of a matrix phase are read at the END of the wave's memory phase, in front of the barrier: the stage
  // a half computes on next was published to that half a barrier earlier (its own pieces, waited for at the end of its previous matrix
  // phase). Exception: SK = 1, half X - its stage also holds pieces of half Y that the middle barrier publishes.
  constexpr int AHEAD = (DP_PWS_OPT & 2) ? 10 : (SK == 2 ? 5 : 6);     // SK = 2 + residual: 6 spills 7 registers
  constexpr int NF = 16 * PT;
  u32x4 bf[AHEAD + 1];
  int va[4];
  auto frag_at = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
    constexpr int f = decltype(ff)::value;
    constexpr int c = f / PT, pt = f % PT;
    return *reinterpret_cast<const u32x4*>(smem + va[c & 3] + (pt * 16384 + (c >> 2) * 256));
  };
  auto mat_prefetch = [&](int s) __attribute__((always_inline)) {
    const int sb = ((s < 0 ? 0 : s) % NSTAGE) * STAGE_B;
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) va[cl] = sb + fo[cl];
    static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag_at(ff); });
  };
  // matrix phase of step s: fragment (K step c, pixel tile pt) feeds both cout tiles; then the blocks this wave does not own go to
  // their owners' exchange slots. pre: mat_prefetch(s) has run. DMA: the pieces of step s + 1 are issued BETWEEN the MFMA groups
  // (lockstep form: the stage they fill was read in step s - 1 and is free from the barrier on; a wave that stalls on an issue leaves
  // the matrix pipe to its SIMD partner, which is in the same phase)
  auto mat_phase = [&](int s, bool pre, auto dma) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    if (s >= 0 && s < nst) {
      if (!pre) mat_prefetch(s);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) { acc[pt][0] = bias_t[0]; acc[pt][1] = bias_t[1]; }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        constexpr int c = f / PT, pt = f % PT;
        if constexpr (f + AHEAD < NF && !(DP_PWS_EXP & 8)) bf[(f + AHEAD) % (AHEAD + 1)] = frag_at(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DP_PWS_EXP & 4)) {
          Mma<T>::run(wfr[c], bf[f % (AHEAD + 1)], acc[pt][0]);
          Mma<T>::run(wfr[16 + c], bf[f % (AHEAD + 1)], acc[pt][1]);
        } else {
          acc[pt][0][0] += __builtin_bit_cast(float, bf[f % (AHEAD + 1)][0] ^ wfr[c][0]);
          acc[pt][1][0] += __builtin_bit_cast(float, bf[f % (AHEAD + 1)][1] ^ wfr[16 + c][1]);
        }
        if constexpr (DMA && !(DP_PWS_EXP & 1)) {
          constexpr int EV = NF / NPW / 2 > 0 ? NF / NPW / 2 : 1;      // the pieces go out in the first half of the phase
          if constexpr (f % EV == EV - 1 && f / EV < NPW) issue_piece(s + 1, std::integral_constant<int, (f / EV < NPW ? f / EV : 0)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (SK > 1) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ow = blk_owner(pt, t);
            if (ks != ow) *reinterpret_cast<f32x4*>(xslot(s & 1, pt * 2 + t, ks, ow)) = acc[pt][t];
          }
      }
    }
  };

  __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): the builtin, so that the compiler's counter model knows
  DP_STAMP(0)
  __builtin_amdgcn_s_barrier();
  DP_STAMP(1)

  if constexpr (SKEW) {
    // Iteration i, first half:  X memory phase (pieces of step i + 1, SK = 1: epilogue of step i - 1) | Y matrix phase of step i - 1
    //              second half: X matrix phase of step i                                             | Y memory phase (pieces of step
    //                                                                                                   i + 1, epilogue of step i - 1)
    // A half waits for its own pieces at the end of its matrix phase (a whole phase of latency cover); the barrier behind that wait
    // publishes them. Stage reuse: X refills buffer (i + 1) % NSTAGE while Y reads step i - 1: another buffer for SK = 1 (three stages),
    // another SLICE for SK > 1; Y refills what it read in the first half of the same iteration, behind the middle barrier.
    // Exchange parity = step parity: X writes step i at the end of iteration i, Y reads it in the second half of iteration i + 1,
    // while X writes step i + 1 into the other parity.
    for (int i = 0; i <= nst; ++i) {
      if (!isY) {
        mem_phase(i);
        if constexpr (SK > 1) mat_prefetch(i);
        DP_STAMP(2)
      } else {
        mat_phase(i - 1, true, std::false_type{});
        DP_STAMP(3)
        __builtin_amdgcn_s_waitcnt(0x0070);
        DP_STAMP(4)
      }
      __builtin_amdgcn_s_barrier();
      DP_STAMP(5)
      if (!isY) {
        mat_phase(i, SK > 1, std::false_type{});
        DP_STAMP(3)
        __builtin_amdgcn_s_waitcnt(0x0070);
        DP_STAMP(4)
      } else {
        mem_phase(i);
        mat_prefetch(i);
        DP_STAMP(2)
      }
      __builtin_amdgcn_s_barrier();
      DP_STAMP(6)
    }
  } else {
    // Lockstep: every wave finalizes step i - 1 (the exchange slots were published by the barrier), loads the residual values of step
    // i, runs the matrix phase of step i with the pieces of step i + 1 going out between its MFMA groups, sends; one barrier per step.
    // Both waves of a SIMD are in the matrix phase together: the pipe is never idle while either has an MFMA ready.
    for (int i = 0; i <= nst; ++i) {
      if (owner) {
        finalize(i - 1);
        if constexpr (has_res) res_issue(i);
      }
      DP_STAMP(2)
      __builtin_amdgcn_sched_barrier(0);
      mat_phase(i, false, std::true_type{});
      DP_STAMP(3)
      __builtin_amdgcn_s_waitcnt(0x0070);
      DP_STAMP(4)
      __builtin_amdgcn_s_barrier();
      DP_STAMP(6)
    }
  }
  if constexpr (DP_PWS_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[(blockIdx.x * 8 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 8 + wave) * 16 + 8] = nst;
    }
  }
#undef DP_STAMP
}

// =====================================================================================================
// K = 256 channels (conv3 of res4: 256 -> 1024 + residual + ReLU, resnet.py:199-205 - five launches for R50, twenty-two for R101). The
// same scheme with the register budget cut the other way: a wave keeps 64 couts x 256 K (4 MFMA row tiles x 8 K steps = the same 32
// fragments), the eight waves are eight cout sub-slices of one K slice - no exchange - and a workgroup covers 512 couts. A pixel row is
// 512 bytes: one LDS-DMA piece carries two of them, a step's stage is 16 KiB. On the streaming kernel (weights in LDS, pixel fragments
// from global memory, one wave per SIMD) the layer ran at 3.3 - 3.6 TB/s of its algorithmic bytes with all of them in the Infinity Cache.
// =====================================================================================================
struct PwqCfg {
  static constexpr int CW = 512, P = 32, PT = 2, KC = 8, NT = 4, RB = 512;     // couts / pixels per step / pixel tiles / K steps / cout tiles / row bytes
  static constexpr int STAGE_B = P * RB;                                       // 16 KiB
  static constexpr int WSCR = 2 * STAGE_B;
  static constexpr int LDS = WSCR + 8 * 8192;
};

template <typename T, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv1x1_pwq_kernel(const PwsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int CW = PwqCfg::CW, P = PwqCfg::P, PT = PwqCfg::PT, KC = PwqCfg::KC, NT = PwqCfg::NT, RB = PwqCfg::RB, STAGE_B = PwqCfg::STAGE_B;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));
  const int s_begin = (int)((long long)p.S * pg / p.n_pg), s_end = (int)((long long)p.S * (pg + 1) / p.n_pg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.in), 0, HAS_RES ? p.res_bytes : 0u, 0x00020000);

  // piece j of this wave = pixels 2 (2 wave + j), + 1 of the step (two 512-byte rows per KiB): lane l -> pixel + (l >> 5), position l & 31,
  // which holds chunk (l & 31) ^ (pixel & 15)
  auto issue_piece = [&](int s, auto jj) __attribute__((always_inline)) {
    constexpr int j = decltype(jj)::value;
    const int m0 = s < nst ? (s_begin + s) * P : p.M;
    const int q = (wave * 2 + j) * 2 + (lane >> 5);
    const int m = m0 + q;
    const int off = m < p.M ? m * RB + (((lane & 31) ^ (q & 15)) << 4) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(smem + (s & 1) * STAGE_B + (wave * 2 + j) * 1024), 16, off, 0, 0, 0);
  };
  issue_piece(0, std::integral_constant<int, 0>{});
  issue_piece(0, std::integral_constant<int, 1>{});

  // ---- weights: cout sub-slice `wave` (64 couts = physical tiles cbase / 16 .. + 3 of the packed matrix) x 8 K planes, through LDS
  const int cbase = slice * CW + wave * 64;
  u32x4 wfr[32];
  {
    const int n_planes = p.kpad * 2 / 64;
    unsigned char* const scr = smem + PwqCfg::WSCR + wave * 8192;
    const int lrow = lane >> 2;
    const unsigned char* __restrict__ wsrc = reinterpret_cast<const unsigned char*>(p.w) + lrow * 64 + (((lane & 3) ^ swz(lrow)) << 4);
    const unsigned char* const rd = scr + fr * 64 + ((fq ^ swz(fr)) << 4);
    auto issue_round = [&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = r * 4 + j, t = k / KC, c = k % KC;
        const long long tile = (long long)((cbase >> 4) + t) * n_planes + c;
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(wsrc + tile * 1024), DP_LDS_PTR(scr + (r & 1) * 4096 + j * 1024), 16, 0, 0);
      }
    };
    issue_round(std::integral_constant<int, 0>{});
    issue_round(std::integral_constant<int, 1>{});
    static_for<0, 8>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      if constexpr (r < 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) wfr[r * 4 + j] = *reinterpret_cast<const u32x4*>(rd + (r & 1) * 4096 + j * 1024);
      if constexpr (r + 2 < 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_round(std::integral_constant<int, (r + 2 < 8 ? r + 2 : 0)>{});
      }
    });
  }
  // bias = initial accumulator value: tile t, rows 4 fq .. + 3 = couts cbase + (t >> 1) * 32 + fq * 8 + (t & 1) * 4 + e
  f32x4 bias_t[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bias_t[t] = *reinterpret_cast<const f32x4*>(p.bias + cbase + (t >> 1) * 32 + fq * 8 + (t & 1) * 4);
  const int opix = p.cout * 2;
  int fo[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) fo[cl] = fr * RB + ((cl ^ (fr >> 2)) << 6) + ((fq ^ (fr & 3)) << 4);

  f32x4 acc[PT][NT];
  u32x4 rv[PT][2];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    rv[pt][0] = rv[pt][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto finalize = [&](int s) __attribute__((always_inline)) {
    const int m0 = (s_begin + s) * P;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = acc[pt][2 * h][k]; v[4 + k] = acc[pt][2 * h + 1][k]; }
        if constexpr (HAS_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += Elem<T>::unpack(rv[pt][h][k] & 0xffffu);
            v[2 * k + 1] += Elem<T>::unpack(rv[pt][h][k] >> 16);
          }
        }
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        const int m = m0 + pt * 16 + fr;
        const int off = (s >= 0 && m < p.M) ? m * opix + (cbase + h * 32 + fq * 8) * 2 : OOB;
        const u32x4 pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]), Elem<T>::pack2(v[4], v[5]), Elem<T>::pack2(v[6], v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, off, 0, 0);
      }
  };
  auto res_issue = [&](int s) __attribute__((always_inline)) {
    const int m0 = s < nst ? (s_begin + s) * P : p.M;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = m0 + pt * 16 + fr;
        rv[pt][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, m < p.M ? m * opix + (cbase + h * 32 + fq * 8) * 2 : OOB, 0, 0);
      }
  };
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();

  constexpr int AHEAD = 4, NF = KC * PT;
  for (int i = 0; i <= nst; ++i) {
    finalize(i - 1);
    if constexpr (HAS_RES) res_issue(i);
    __builtin_amdgcn_sched_barrier(0);
    if (i < nst) {
      const int sb = (i & 1) * STAGE_B;
      int va[4];
#pragma unroll
      for (int cl = 0; cl < 4; ++cl) va[cl] = sb + fo[cl];
      auto frag_at = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
        constexpr int f = decltype(ff)::value;
        constexpr int c = f / PT, pt = f % PT;
        return *reinterpret_cast<const u32x4*>(smem + va[c & 3] + (pt * 16 * RB + (c >> 2) * 256));
      };
      u32x4 bf[AHEAD + 1];
      static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag_at(ff); });
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[pt][t] = bias_t[t];
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        constexpr int c = f / PT, pt = f % PT;
        if constexpr (f + AHEAD < NF) bf[(f + AHEAD) % (AHEAD + 1)] = frag_at(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) Mma<T>::run(wfr[t * KC + c], bf[f % (AHEAD + 1)], acc[pt][t]);
        // the two pieces of step i + 1 go out between the MFMA groups of the first half of the phase
        if constexpr (f == 1) issue_piece(i + 1, std::integral_constant<int, 0>{});
        if constexpr (f == 5) issue_piece(i + 1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
  }
}

template <typename T, bool HAS_RES>
int launch_pwq_r(PwsArgs a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_pwq_kernel<T, HAS_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, PwqCfg::LDS);
    attr_set = true;
  }
  a.S = (a.M + PwqCfg::P - 1) / PwqCfg::P;
  a.n_slices = a.cout / PwqCfg::CW;
  int groups = pws_num_cus() / (8 * a.n_slices);
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8;
  a.dbg = nullptr;
  hipLaunchKernelGGL((conv1x1_pwq_kernel<T, HAS_RES>), dim3(a.n_pg * a.n_slices), dim3(512), PwqCfg::LDS, stream, a);
  return dp_check_launch("conv1x1_pwq_kernel");
}

template <typename T, int SK, bool HAS_RES, bool SKEW>
int launch_pws_r(PwsArgs a, hipStream_t stream) {
  using Cfg = PwsCfg<T, SK>;
  constexpr int kLds = Cfg::lds(SKEW);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_pws_kernel<T, SK, HAS_RES, SKEW>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr_set = true;
  }
  a.S = (a.M + Cfg::P - 1) / Cfg::P;
  a.n_slices = a.cout / Cfg::CW;
  int groups = pws_num_cus() / (8 * a.n_slices);
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8;
  a.dbg = nullptr;
#if DP_PWS_EXP & 16
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_pg * a.n_slices;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 16 * 8 * 4096);
  a.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 16 * 8 * nblk, stream);
#endif
  hipLaunchKernelGGL((conv1x1_pws_kernel<T, SK, HAS_RES, SKEW>), dim3(a.n_pg * a.n_slices), dim3(512), kLds, stream, a);
#if DP_PWS_EXP & 16
  {
    static int shown = 0;
    if (shown++ == 40) {   // a warm launch
      (void)hipStreamSynchronize(stream);
      unsigned long long* hbuf = (unsigned long long*)malloc(sizeof(unsigned long long) * 128 * nblk);
      (void)hipMemcpy(hbuf, dbg, sizeof(unsigned long long) * 128 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) {
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n = 0, live = 0;
        for (int b = 0; b < nblk; ++b) {
          if (hbuf[(b * 8 + w) * 16 + 8] == 0) continue;
          live += 1;
          for (int k = 0; k < 8; ++k) sum[k] += (double)hbuf[(b * 8 + w) * 16 + k];
          n += (double)hbuf[(b * 8 + w) * 16 + 8];
        }
        fprintf(stderr, "pws<SK=%d,%s> wave %d: prologue %.0f + barrier %.0f cycles per workgroup; per step: dma issue %.0f  epilogue %.0f  matrix phase %.0f  wait %.0f  middle barrier %.0f  end barrier %.0f  (steps/wg %.1f, %0.f workgroups)\n",
                SK, SKEW ? "skew" : "lockstep", w, sum[0] / live, sum[1] / live, sum[7] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n, n / live, live);
      }
      free(hbuf);
    }
  }
#endif
  return dp_check_launch("conv1x1_pws_kernel");
}

// Schedule: lockstep (one barrier per step, the next step's pieces between the MFMA groups). The SKEW form - the halves of the workgroup
// half a step apart, two barriers per step - measured slower on every layer (res4 conv1 21.7 against 20.9 us, res5 conv3 26.1 against
// 22.5: a half's 64-MFMA phase runs at 23 cycles per MFMA beside the other half's epilogue, whose VALU work gets two issue slots per MFMA;
// both waves of a SIMD in the matrix phase TOGETHER keep the pipe at 16 - 20) and is compiled into `make exp` builds only (policy key
// pws_skew = 1 there). Same bits either way: the K slices of an output are added in slice order in both.
template <typename T, int SK>
int launch_pws(const PwsArgs& a, hipStream_t stream) {
#ifdef DP_EXPERIMENTS
  if (dp_policy().pws_skew == 1) return a.res ? launch_pws_r<T, SK, true, true>(a, stream) : launch_pws_r<T, SK, false, true>(a, stream);
#endif
  return a.res ? launch_pws_r<T, SK, true, false>(a, stream) : launch_pws_r<T, SK, false, false>(a, stream);
}

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): a pointwise stride-1 layer with K = 256 / 512 / 1024 / 2048 channels whose cout count is a whole
// number of workgroup slices (512 / 256 / 128 / 64), plain NHWC output, optional residual (plain NHWC of the output's shape, or the half-size top-down
// map of the FPN read through a nearest x2 up-sampling). No size threshold: the summation order is this kernel's own.
bool dp_conv_pws_ok(const dp_conv_params* p) {
  if (dp_policy().conv_pws == 0) return false;
  if (!(p->dtype == DP_BF16 || p->dtype == DP_F16)) return false;
  const int sk = p->Cin == 512 ? 1 : (p->Cin == 1024 ? 2 : (p->Cin == 2048 ? 4 : 0));
  if (sk == 0 && p->Cin != 256) return false;
  const int cw = p->Cin == 256 ? 512 : 256 / sk;      // 256 channels: the 64-couts-per-wave form (conv1x1_pwq_kernel)
  const bool lin_out = p->osW == p->Cout && p->osH == (long long)p->Wo * p->osW && p->osN == (long long)p->Ho * p->osH;
  const bool lin_res = !p->residual || (p->rshift == 0 && p->rsW == p->Cout && p->rsH == (long long)p->Wo * p->rsW && p->rsN == (long long)p->Ho * p->rsH);
  const bool up_res = p->residual && p->rshift == 1 && p->Ho % 2 == 0 && p->Wo % 2 == 0 && p->rsW == p->Cout && p->rsH == (long long)(p->Wo / 2) * p->rsW &&
                      p->rsN == (long long)(p->Ho / 2) * p->rsH;
  return p->ntaps == 1 && p->stride == 1 && (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == 0 && p->wi_off == 0 && p->H == p->Ho && p->W == p->Wo &&
         p->Kpad == p->Cin && p->Cout % cw == 0 && p->Cout <= p->Cout_w && !p->n_dev && !p->in2 && !p->head_out && !p->post_res && p->post_mode == 0 &&
         p->split_k <= 1 && !p->out_f32 && p->out && lin_out && (lin_res || up_res) &&
         // (one image inside the 32-bit offset range; a BATCH beyond it is cut into image chunks by dp_conv_pws_launch - the class never depends on N)
         ((long long)p->Ho * p->Wo + 64) * p->Cin * 2 < (1ll << 31) && ((long long)p->Ho * p->Wo + 64) * p->Cout * 2 < (1ll << 31) &&
         (((uintptr_t)p->in | (uintptr_t)p->out | (uintptr_t)p->weight | (uintptr_t)p->residual) & 15) == 0;
}

int dp_conv_pws_launch(const dp_conv_params* p, dp_stream_t stream) {
  // 32-bit buffer offsets: a batch whose tensors pass the range goes image chunk by image chunk (pixels are independent; the per-pixel
  // arithmetic does not know about the chunking). rows_chunk_bytes: the limit, lowered by tests.
  const long long hw = (long long)p->Ho * p->Wo;
  const long long per_img = (hw + 64) * (p->Cin > p->Cout ? p->Cin : p->Cout) * 2;
  const long long lim = dp_policy().rows_chunk_bytes < (1ll << 31) - 1 ? dp_policy().rows_chunk_bytes : (1ll << 31) - 1;
  int per = (int)(lim / per_img);
  if (per < 1) per = 1;
  hipStream_t s = as_stream(stream);
  const bool bf = p->dtype == DP_BF16;
  for (int n0 = 0; n0 < p->N; n0 += per) {
    const int n = p->N - n0 < per ? p->N - n0 : per;
    PwsArgs a;
    a.in = reinterpret_cast<const unsigned char*>(p->in) + (long long)n0 * hw * p->Cin * 2;
    a.w = p->weight; a.bias = p->bias;
    a.res = p->residual ? reinterpret_cast<const unsigned char*>(p->residual) + (long long)n0 * p->rsN * 2 : nullptr;
    a.out = reinterpret_cast<unsigned char*>(p->out) + (long long)n0 * hw * p->Cout * 2;
    a.M = (int)(n * hw); a.cout = p->Cout; a.relu = p->relu; a.kpad = p->Kpad;
    a.n_slices = a.n_pg = a.S = 0;
    a.dbg = nullptr;
    a.in_bytes = (unsigned)((long long)a.M * p->Cin * 2);
    a.out_bytes = (unsigned)((long long)a.M * p->Cout * 2);
    a.res_up = p->residual && p->rshift == 1;
    a.res_bytes = p->residual ? (unsigned)((long long)n * p->rsN * 2) : 0u;
    a.HoWo = p->Ho * p->Wo; a.Wo = p->Wo;
    a.rsN = (int)p->rsN; a.rsH = (int)p->rsH; a.rsW = (int)p->rsW;
    int rc;
    if (p->Cin == 256) {
      if (a.res) rc = bf ? launch_pwq_r<uint16_t, true>(a, s) : launch_pwq_r<f16_t, true>(a, s);
      else rc = bf ? launch_pwq_r<uint16_t, false>(a, s) : launch_pwq_r<f16_t, false>(a, s);
    } else if (p->Cin == 512) rc = bf ? launch_pws<uint16_t, 1>(a, s) : launch_pws<f16_t, 1>(a, s);
    else if (p->Cin == 1024) rc = bf ? launch_pws<uint16_t, 2>(a, s) : launch_pws<f16_t, 2>(a, s);
    else rc = bf ? launch_pws<uint16_t, 4>(a, s) : launch_pws<f16_t, 4>(a, s);
    if (rc != DP_OK) return rc;
  }
  return DP_OK;
}
