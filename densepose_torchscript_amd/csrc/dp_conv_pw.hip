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
};

template <typename T, int SK>
struct PwsCfg {
  static constexpr int NCS = 8 / SK;                 // cout sub-slices (32 couts) per workgroup
  static constexpr int CW = NCS * 32;                // couts per workgroup
  static constexpr int PT = SK == 4 ? 1 : 2;         // pixel tiles per step
  static constexpr int P = 16 * PT;
  static constexpr int SLICE_B = P * 1024;           // one K slice of a stage
  static constexpr int STAGE_B = SK * SLICE_B;
  static constexpr int NSTAGE = 2;
  static constexpr int NBLK = PT * 2;                // output blocks (pixel tile, cout tile) per wave
  static constexpr int XCH_B = SK == 1 ? 0 : NCS * NBLK * (SK - 1) * 1024;   // exchange buffer of one parity
  static constexpr int LDS = NSTAGE * STAGE_B + 2 * XCH_B;
  static constexpr int NPW = P * SK / 8;             // DMA pieces per wave and step
};

template <typename T, int SK, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv1x1_pws_kernel(const PwsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(SK == 1 || SK == 2 || SK == 4, "K = 512 / 1024 / 2048 channels");
  using Cfg = PwsCfg<T, SK>;
  constexpr int NCS = Cfg::NCS, CW = Cfg::CW, PT = Cfg::PT, P = Cfg::P, SLICE_B = Cfg::SLICE_B, STAGE_B = Cfg::STAGE_B, NSTAGE = Cfg::NSTAGE;
  constexpr int NBLK = Cfg::NBLK, XCH_B = Cfg::XCH_B, NPW = Cfg::NPW;
  constexpr int KB = SK * 1024;                      // bytes of a pixel's channel row
  constexpr int XCH0 = NSTAGE * STAGE_B;
  constexpr int OOB = (int)0x80000000;
  static_assert(Cfg::LDS <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cs = wave % NCS, ks = wave / NCS;
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

  // ---- pixel rows of step s -> stage s % NSTAGE: piece j of this wave = (K slice, pixel) pc = wave * NPW + j
  auto issue_stage = [&](int s) __attribute__((always_inline)) {
    const int m0 = s < nst ? (s_begin + s) * P : p.M;      // past the workgroup's range: every lane out of range, nothing moves
    unsigned char* const sb = smem + (s % NSTAGE) * STAGE_B;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pc = wave * NPW + j;
      const int kp = pc / P, q = pc % P;               // wave-uniform
      const int m = m0 + q;
      const int off = m < p.M ? m * KB + kp * 1024 + ((lane ^ (q & 15)) << 4) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(sb + kp * SLICE_B + q * 1024), 16, off, 0, 0, 0);
    }
  };
  issue_stage(0);

  // ---- this wave's 32 couts x its 512 K values, in registers for the whole launch. Rows of the two MFMA tiles (A row fr of tile t):
  //      SK <= 2: logical cout cbase + (fr >> 2) * 8 + t * 4 + (fr & 3) - a lane's 8 accumulator values of a pixel are 8 consecutive
  //               channels (one 16-byte store); SK = 4 (an owner stores ONE tile): cbase + t * 16 + fr - 4 consecutive channels per lane
  const int cbase = slice * CW + cs * 32;
  u32x4 wfr[32];
  {
    const int n_planes = p.kpad * 2 / 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int L = SK == 4 ? cbase + t * 16 + fr : cbase + (fr >> 2) * 8 + t * 4 + (fr & 3);
      const int l64 = L & 63, rem = l64 & 31;
      const int phys = (L & ~63) + (((l64 >> 5) * 2 + ((rem >> 2) & 1)) * 16) + (rem >> 3) * 4 + (rem & 3);   // pack.py's row permutation
      const unsigned char* __restrict__ w = reinterpret_cast<const unsigned char*>(p.w) + dp_wtile_off(phys, ks * 16, fq, n_planes);
#pragma unroll
      for (int c = 0; c < 16; ++c) wfr[t * 16 + c] = *reinterpret_cast<const u32x4*>(w + c * 1024);
    }
  }

  // ---- output ownership. SK = 1: every block is the wave's own. SK = 2: pixel tile ks (both cout tiles). SK = 4 (one pixel tile):
  //      K slices 0 / 1 own cout tile 0 / 1, slices 2 / 3 only send.
  const bool owner = SK == 1 || SK == 2 || ks < 2;
  const int o_pt = SK == 2 ? ks : 0;                  // SK = 1: both pixel tiles (loop below)
  const int o_t = SK == 4 ? (ks & 1) : 0;
  constexpr int NOWN = SK == 1 ? 2 : 1;                // 16-pixel rows of output this wave finalizes per step
  constexpr int NV = SK == 4 ? 4 : 8;                  // channels per lane and row
  const int cown = SK == 4 ? cbase + o_t * 16 + fq * 4 : cbase + fq * 8;
  float bias[NV];
#pragma unroll
  for (int k = 0; k < NV; k += 4) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cown + k);
    bias[k] = bv[0]; bias[k + 1] = bv[1]; bias[k + 2] = bv[2]; bias[k + 3] = bv[3];
  }
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

  // exchange slots of one parity: [cs][block][sender index among the non-owners][lane * 16]
  auto xslot = [&](int par, int blk, int sender, int own_ks) __attribute__((always_inline)) -> unsigned char* {
    const int j = sender < own_ks ? sender : sender - 1;
    return smem + XCH0 + par * XCH_B + ((cs * NBLK + blk) * (SK - 1) + j) * 1024 + lane * 16;
  };

  f32x4 eacc[NOWN][2];          // the pending step's own blocks (SK = 4: [0][0] only)
  u32x4 rv[NOWN];               // its residual values
#pragma unroll
  for (int r = 0; r < NOWN; ++r) { rv[r] = u32x4{0u, 0u, 0u, 0u}; eacc[r][0] = eacc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  int e_m0 = 0;
  bool have_prev = false;

  auto finalize = [&](int par, bool live) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NOWN; ++r) {
      const int pt = SK == 1 ? r : o_pt;
      float v[NV];
      if constexpr (SK == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = eacc[r][0][k]; v[4 + k] = eacc[r][1][k]; }
      } else if constexpr (SK == 2) {
        // two K slices: own + partner (a + b is commutative bit for bit: no order to keep)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(xslot(par, pt * 2 + t, 1 - ks, ks));
#pragma unroll
          for (int k = 0; k < 4; ++k) v[t * 4 + k] = ks == 0 ? eacc[0][t][k] + o[k] : o[k] + eacc[0][t][k];
        }
      } else {
        // four K slices, added in slice order whoever owns the block
        f32x4 s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
          if (q == ks) o = eacc[0][0];
          else o = *reinterpret_cast<const f32x4*>(xslot(par, o_t, q, ks));
          if (q == 0) s = o;
          else {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += o[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = s[k];
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] += bias[k];
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
      const int m = e_m0 + pt * 16 + fr;
      const int off = (live && m < p.M) ? m * opix + cown * 2 : OOB;
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
  auto res_issue = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NOWN; ++r) {
      const int pt = SK == 1 ? r : o_pt;
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

  __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0): the builtin, so that the compiler's counter model knows
  __builtin_amdgcn_s_barrier();

  for (int i = 0; i <= nst; ++i) {
    // (1) the next step's pixel rows: issued before this wave has LDS reads in flight
    //     (unconditional - a step past the range reads nothing - so that the compiler can COUNT the vector-memory operations
    //     between a residual load and its use: behind a branch it waits vmcnt(0), i.e. for the pieces just issued)
    issue_stage(i + 1);
    asm volatile("" ::: "memory");
    // (2) the previous step's epilogue (iteration 0: masked stores), (3) this step's residual values (consumed one iteration later)
    if (SK != 4 || owner) finalize((i - 1) & 1, have_prev);
    have_prev = false;
    if (i < nst) {
      const int m0 = (s_begin + i) * P;
      if constexpr (has_res) { if (SK != 4 || owner) res_issue(m0); }
      __builtin_amdgcn_sched_barrier(0);

      // (4) the step's matrix work: fragment (K step c, pixel tile pt) feeds both cout tiles
      const int sb = (i % NSTAGE) * STAGE_B;
      int va[4];
#pragma unroll
      for (int cl = 0; cl < 4; ++cl) va[cl] = sb + fo[cl];
      f32x4 acc[PT][2];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int NF = 16 * PT;
      auto frag = [&](auto ff) __attribute__((always_inline)) -> u32x4 {
        constexpr int f = decltype(ff)::value;
        constexpr int c = f / PT, pt = f % PT;
        return *reinterpret_cast<const u32x4*>(smem + va[c & 3] + (pt * 16384 + (c >> 2) * 256));
      };
      constexpr int AHEAD = 6;
      u32x4 bf[AHEAD + 1];
      static_for<0, AHEAD>([&](auto ff) { bf[decltype(ff)::value] = frag(ff); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        constexpr int c = f / PT, pt = f % PT;
        if constexpr (f + AHEAD < NF) bf[(f + AHEAD) % (AHEAD + 1)] = frag(std::integral_constant<int, (f + AHEAD < NF ? f + AHEAD : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        Mma<T>::run(wfr[c], bf[f % (AHEAD + 1)], acc[pt][0]);
        Mma<T>::run(wfr[16 + c], bf[f % (AHEAD + 1)], acc[pt][1]);
        __builtin_amdgcn_sched_barrier(0);
      });

      // (5) own blocks wait in registers for the next iteration, the others go to their owners' exchange slots
      if constexpr (SK == 1) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) { eacc[pt][0] = acc[pt][0]; eacc[pt][1] = acc[pt][1]; }
      } else if constexpr (SK == 2) {
        // wave ks owns pixel tile ks: the other tile's two blocks go to the partner (K slice 1 - ks)
        const f32x4 s0 = ks == 0 ? acc[1][0] : acc[0][0], s1 = ks == 0 ? acc[1][1] : acc[0][1];
        const int spt = 1 - ks;
        *reinterpret_cast<f32x4*>(xslot(i & 1, spt * 2 + 0, ks, 1 - ks)) = s0;
        *reinterpret_cast<f32x4*>(xslot(i & 1, spt * 2 + 1, ks, 1 - ks)) = s1;
        eacc[0][0] = ks == 0 ? acc[0][0] : acc[1][0];
        eacc[0][1] = ks == 0 ? acc[0][1] : acc[1][1];
      } else {
        // block t belongs to K slice t: every other slice sends it there
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (ks != t) *reinterpret_cast<f32x4*>(xslot(i & 1, t, ks, t)) = acc[0][t];
        }
        eacc[0][0] = ks == 1 ? acc[0][1] : acc[0][0];
      }
      e_m0 = m0;
      have_prev = true;
    }
    // the next stage has landed, the exchange slots are written: one barrier per step
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
  }
}

template <typename T, int SK, bool HAS_RES>
int launch_pws_r(PwsArgs a, hipStream_t stream) {
  using Cfg = PwsCfg<T, SK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_pws_kernel<T, SK, HAS_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_set = true;
  }
  a.S = (a.M + Cfg::P - 1) / Cfg::P;
  a.n_slices = a.cout / Cfg::CW;
  int groups = pws_num_cus() / (8 * a.n_slices);
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8;
  hipLaunchKernelGGL((conv1x1_pws_kernel<T, SK, HAS_RES>), dim3(a.n_pg * a.n_slices), dim3(512), Cfg::LDS, stream, a);
  return dp_check_launch("conv1x1_pws_kernel");
}

template <typename T, int SK>
int launch_pws(const PwsArgs& a, hipStream_t stream) {
  return a.res ? launch_pws_r<T, SK, true>(a, stream) : launch_pws_r<T, SK, false>(a, stream);
}

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): a pointwise stride-1 layer with K = 512 / 1024 / 2048 channels whose cout count is a whole number
// of workgroup slices (256 / 128 / 64), plain NHWC output, optional residual (plain NHWC of the output's shape, or the half-size top-down
// map of the FPN read through a nearest x2 up-sampling). No size threshold: the summation order is this kernel's own.
bool dp_conv_pws_ok(const dp_conv_params* p) {
  if (dp_policy().conv_pws == 0) return false;
  if (!(p->dtype == DP_BF16 || p->dtype == DP_F16)) return false;
  const int sk = p->Cin == 512 ? 1 : (p->Cin == 1024 ? 2 : (p->Cin == 2048 ? 4 : 0));
  if (sk == 0) return false;
  const int cw = 256 / sk;
  const long long M = (long long)p->N * p->Ho * p->Wo;
  const bool lin_out = p->osW == p->Cout && p->osH == (long long)p->Wo * p->osW && p->osN == (long long)p->Ho * p->osH;
  const bool lin_res = !p->residual || (p->rshift == 0 && p->rsW == p->Cout && p->rsH == (long long)p->Wo * p->rsW && p->rsN == (long long)p->Ho * p->rsH);
  const bool up_res = p->residual && p->rshift == 1 && p->Ho % 2 == 0 && p->Wo % 2 == 0 && p->rsW == p->Cout && p->rsH == (long long)(p->Wo / 2) * p->rsW &&
                      p->rsN == (long long)(p->Ho / 2) * p->rsH;
  return p->ntaps == 1 && p->stride == 1 && (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == 0 && p->wi_off == 0 && p->H == p->Ho && p->W == p->Wo &&
         p->Kpad == p->Cin && p->Cout % cw == 0 && p->Cout <= p->Cout_w && !p->n_dev && !p->in2 && !p->head_out && !p->post_res && p->post_mode == 0 &&
         p->split_k <= 1 && !p->out_f32 && p->out && lin_out && (lin_res || up_res) &&
         (M + 64) * p->Cin * 2 < (1ll << 31) && (M + 64) * p->Cout * 2 < (1ll << 31) &&
         (((uintptr_t)p->in | (uintptr_t)p->out | (uintptr_t)p->weight | (uintptr_t)p->residual) & 15) == 0;
}

int dp_conv_pws_launch(const dp_conv_params* p, dp_stream_t stream) {
  PwsArgs a;
  a.in = p->in; a.w = p->weight; a.bias = p->bias; a.res = p->residual; a.out = p->out;
  a.M = p->N * p->Ho * p->Wo; a.cout = p->Cout; a.relu = p->relu; a.kpad = p->Kpad;
  a.n_slices = a.n_pg = a.S = 0;
  a.in_bytes = (unsigned)((long long)a.M * p->Cin * 2);
  a.out_bytes = (unsigned)((long long)a.M * p->Cout * 2);
  a.res_up = p->residual && p->rshift == 1;
  a.res_bytes = p->residual ? (unsigned)((long long)p->N * p->rsN * 2) : 0u;
  a.HoWo = p->Ho * p->Wo; a.Wo = p->Wo;
  a.rsN = (int)p->rsN; a.rsH = (int)p->rsH; a.rsW = (int)p->rsW;
  hipStream_t s = as_stream(stream);
  const bool bf = p->dtype == DP_BF16;
  if (p->Cin == 512) return bf ? launch_pws<uint16_t, 1>(a, s) : launch_pws<f16_t, 1>(a, s);
  if (p->Cin == 1024) return bf ? launch_pws<uint16_t, 2>(a, s) : launch_pws<f16_t, 2>(a, s);
  return bf ? launch_pws<uint16_t, 4>(a, s) : launch_pws<f16_t, 4>(a, s);
}
