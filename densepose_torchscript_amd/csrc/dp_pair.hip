// conv3 (+ residual, ReLU) of a bottleneck block and conv1 (+ ReLU) of the NEXT block in ONE launch, for the plain blocks of res3
// (128 -> 512 -> 128 channels; /root/reference/detectron2/modeling/backbone/resnet.py:199-205 and :192-193 of the following block):
//
//     out = relu(conv3_1x1(t2) + b3 + residual)        128 -> 512
//     t1' = relu(conv1'_1x1(out) + b1')                512 -> 128
//
// Run as two launches the block output (138 MB at batch 8) is written by conv3 and read back by conv1 at once: 482 MB of traffic for two
// HBM-bound layers (310 MB at 4.0 TB/s + 172 MB at 4.0 TB/s = 121 us, profiles/r4_layers.txt). Chained, it is written once: 344 MB.
//
// Both weight matrices (128 KiB each) live in the register file of the workgroup for the whole launch, split so that the chain needs no
// transposition: wave w of 8 owns output channels [64 w, 64 w + 64) of conv3 - and those 64 channels are exactly K steps 2 w, 2 w + 1 of
// conv1', whose weights for ALL 128 couts it holds as well (16 + 16 fragments = 128 VGPRs). pack.py's row permutation makes a lane's
// accumulator values of a 64-cout block two runs of 8 consecutive channels of one pixel = the B fragments of those two K steps
// (dp_bottleneck.hip uses the same property), so conv3's epilogue registers feed conv1' directly. conv1' is therefore K-split eight ways:
// every wave holds a partial sum of all 128 couts over its 64 channels; MFMA row tile t of the result belongs to wave t, the seven others
// send their block through an exchange buffer (double-buffered by step parity) and the owner adds the eight partial sums IN WAVE ORDER -
// a fixed order, so a pixel's bits do not depend on the batch or on where the pixel falls - then bias, ReLU, one rounding.
//   * a step = 16 pixels (the flat pixel axis: both layers are pointwise), a workgroup owns a contiguous range of steps, one barrier per
//     step; the epilogue of conv1' runs one step late (after the barrier that publishes the exchange slots).
//   * memory: t2 rows (256 B) come by LDS-DMA into a 4-stage ring, residual runs (16 B per lane) straight into a 3-deep register ring, both
//     TWO steps ahead of their use (a step is ~1.5 us: one step of distance would cap the chip at ~2.5 TB/s of loads in flight); counted
//     vmcnt waits - every wave issues the same six vector-memory operations per step (waves 4 .. 7 an out-of-range LDS-DMA piece) - so that
//     waiting for step i + 1's operands never waits for step i + 2's.
//   * conv3's result is bit-identical to the separate launch (same K order, (acc + bias) + residual); conv1' adds eight 64-channel partial
//     sums instead of one 512-channel chain: its own summation order, which is why a call site runs here for every batch size or never.
#include "dp_common.h"
#include "dp_mma.h"
#include "dp_policy.h"

namespace {

int pair_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

struct PairArgs {
  const void* t2;
  const void* res;
  void* out;
  void* t1n;
  const void* w3;
  const void* w1;
  const float* b3;
  const float* b1;
  int M, S, n_wg;
  unsigned t2_bytes, out_bytes, t1n_bytes;
};

constexpr int kPairP = 16;                        // pixels per step
constexpr int kPairStage = kPairP * 256;          // a step's t2 rows
constexpr int kPairNStage = 4;
constexpr int kPairXch0 = kPairNStage * kPairStage;         // exchange: [parity][owner tile 8][sender slot 7][1 KiB]
constexpr int kPairXch = 8 * 7 * 1024;
constexpr int kPairBias = kPairXch0 + 2 * kPairXch;         // b3 (512 floats)
constexpr int kPairDump = kPairBias + 512 * 4;              // where the out-of-range pieces of waves 4 .. 7 land (an out-of-range LDS-DMA writes ZEROS)
constexpr int kPairLds = kPairDump + 1024;

template <typename T>
__global__ __launch_bounds__(512, 2) void bottleneck_pair128_kernel(const PairArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const b3_s = reinterpret_cast<float*>(smem + kPairBias);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int s_begin = (int)((long long)p.S * blockIdx.x / p.n_wg), s_end = (int)((long long)p.S * (blockIdx.x + 1) / p.n_wg);
  const int nst = s_end - s_begin;
  if (nst <= 0) return;

  const __amdgpu_buffer_rsrc_t rs_t2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t2), 0, p.t2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t1n = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);

  // ---- weights -> registers through LDS (whole-tile LDS-DMA copies, the ring kernels' conflict-free image; scratch = the exchange area)
  //      w3: physical tiles 4 wave + t (t < 4: this wave's 64 couts) x K planes c < 4;  w1: physical tiles t < 8 x K planes 2 wave + h
  u32x4 w3fr[16], w1fr[16];
  {
    unsigned char* const scr = smem + kPairXch0 + wave * 8192;
    const int lrow = lane >> 2;
    const int goff = lrow * 64 + (((lane & 3) ^ swz(lrow)) << 4);
    const unsigned char* __restrict__ s3 = reinterpret_cast<const unsigned char*>(p.w3) + goff;
    const unsigned char* __restrict__ s1 = reinterpret_cast<const unsigned char*>(p.w1) + goff;
    const unsigned char* const rd = scr + fr * 64 + ((fq ^ swz(fr)) << 4);
    auto issue_round = [&](auto rr) __attribute__((always_inline)) {
      constexpr int r = decltype(rr)::value;          // rounds 0 .. 3: w3 tile r, planes 0 .. 3; rounds 4 .. 7: w1 tiles 2 (r - 4), + 1, planes 2 wave, + 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned char* src;
        if constexpr (r < 4) src = s3 + (long long)((4 * wave + r) * 4 + j) * 1024;
        else src = s1 + (long long)((2 * (r - 4) + (j >> 1)) * 16 + 2 * wave + (j & 1)) * 1024;
        __builtin_amdgcn_global_load_lds(DP_GLOBAL_PTR(src), DP_LDS_PTR(scr + (r & 1) * 4096 + j * 1024), 16, 0, 0);
      }
    };
    issue_round(std::integral_constant<int, 0>{});
    issue_round(std::integral_constant<int, 1>{});
    static_for<0, 8>([&](auto rr) {
      constexpr int r = decltype(rr)::value;
      if constexpr (r < 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(rd + (r & 1) * 4096 + j * 1024);
        if constexpr (r < 4) w3fr[r * 4 + j] = v;              // [tile r][plane j]
        else w1fr[(2 * (r - 4) + (j >> 1)) * 2 + (j & 1)] = v;  // [tile][h]
      }
      if constexpr (r + 2 < 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_round(std::integral_constant<int, (r + 2 < 8 ? r + 2 : 0)>{});
      }
    });
    b3_s[tid] = p.b3[tid];
  }
  // conv1' tile `wave` is this wave's: rows 4 fq .. + 3 = logical couts (wave >> 2) * 64 + ((wave & 3) >> 1) * 32 + fq * 8 + (wave & 1) * 4 + e
  const int c1 = (wave >> 2) * 64 + ((wave & 3) >> 1) * 32 + fq * 8 + (wave & 1) * 4;
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + c1);
  const int c3 = wave * 64 + fq * 8;          // this lane's runs of conv3: channels c3 + h * 32 .. + 7

  // ---- memory: per step and wave one LDS-DMA piece (4 pixels x 256 B; waves 0 .. 3 - the others issue it out of range), two residual
  //      loads, two output stores, one t1' store
  auto issue_dma = [&](int s) __attribute__((always_inline)) {
    const int q = wave * 4 + (lane >> 4);                 // pixel of the step (waves >= 4: none)
    const int m = (s_begin + s) * kPairP + q;
    const int off = (wave < 4 && s < nst && m < p.M) ? m * 256 + (((lane & 15) ^ (q & 15)) << 4) : OOB;
    unsigned char* const dst = wave < 4 ? smem + (s & (kPairNStage - 1)) * kPairStage + wave * 1024 : smem + kPairDump;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t2, DP_LDS_PTR(dst), 16, off, 0, 0, 0);
  };
  auto res_load = [&](int s, u32x4 (&dst)[2]) __attribute__((always_inline)) {
    const int m = (s_begin + s) * kPairP + fr;
    const int off = (s < nst && m < p.M) ? m * 1024 + c3 * 2 : OOB;
    dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0);
    dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off == OOB ? OOB : off + 64, 0, 0);
  };
  u32x4 rv[3][2];
  issue_dma(0);
  res_load(0, rv[0]);
  issue_dma(1);
  res_load(1, rv[1]);
  rv[2][0] = rv[2][1] = u32x4{0u, 0u, 0u, 0u};

  // exchange slot of partial block (owner tile t, sender q): the senders of a tile in wave order, skipping the owner
  auto xslot = [&](int par, int t, int q) __attribute__((always_inline)) -> unsigned char* {
    return smem + kPairXch0 + par * kPairXch + (t * 7 + (q < t ? q : q - 1)) * 1024 + lane * 16;
  };
  const int frag_lane = fr * 256;            // + ((c * 4 + fq) ^ fr) * 16 per K step c

  f32x4 own = f32x4{0.f, 0.f, 0.f, 0.f};     // this wave's partial of ITS tile, pending step
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();

  // one step; R = register-ring slot of its residual values (steps are unrolled by three so that the slot is static)
  auto step = [&](int i, auto rr) __attribute__((always_inline)) {
    constexpr int R = decltype(rr)::value;
    // [A] conv1' epilogue of step i - 1: eight partial sums in wave order (+ bias, ReLU), 4 channels per lane
    {
      const int par = (i - 1) & 1;
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (j == wave) {
#pragma unroll
          for (int k = 0; k < 4; ++k) s[k] += own[k];
        }
        const f32x4 o = *reinterpret_cast<const f32x4*>(smem + kPairXch0 + par * kPairXch + (wave * 7 + j) * 1024 + lane * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += o[k];
      }
      if (wave == 7) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += own[k];
      }
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(s[k] + bias1[k], 0.f);
      typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t pk = {Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3])};
      const int m = (s_begin + i - 1) * kPairP + fr;
      __builtin_amdgcn_raw_buffer_store_b64(pk, rs_t1n, (i >= 1 && i <= nst && m < p.M) ? m * 256 + c1 * 2 : OOB, 0, 0);
    }
    // [B] t2 rows and [C] residual runs of step i + 2
    issue_dma(i + 2);
    res_load(i + 2, rv[(R + 2) % 3]);
    __builtin_amdgcn_sched_barrier(0);

    // conv3: this wave's 64 couts of the step's 16 pixels
    const unsigned char* const st = smem + (i & (kPairNStage - 1)) * kPairStage + frag_lane;
    u32x4 bf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bf[c] = *reinterpret_cast<const u32x4*>(st + (((c * 4 + fq) ^ fr) << 4));
    f32x4 acc3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) Mma<T>::run(w3fr[t * 4 + c], bf[c], acc3[t]);
    u32x4 xf[2];
    const int m = (s_begin + i) * kPairP + fr;
    const int o_off = (i < nst && m < p.M) ? m * 1024 + c3 * 2 : OOB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b3_s + c3 + h * 32), b1 = *reinterpret_cast<const f32x4*>(b3_s + c3 + h * 32 + 4);
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = acc3[2 * h][k] + b0[k]; v[4 + k] = acc3[2 * h + 1][k] + b1[k]; }
      const u32x4 r = rv[R][h];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += Elem<T>::unpack(r[k] & 0xffffu);
        v[2 * k + 1] += Elem<T>::unpack(r[k] >> 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) xf[h][k] = Elem<T>::pack2(fmaxf(v[2 * k], 0.f), fmaxf(v[2 * k + 1], 0.f));
      // [D] the block output
      __builtin_amdgcn_raw_buffer_store_b128(xf[h], rs_out, o_off == OOB ? OOB : o_off + h * 64, 0, 0);
    }
    // conv1' partial sums over this wave's 64 channels (K steps 2 wave, 2 wave + 1), all 128 couts
    f32x4 acc1[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 8; ++t) Mma<T>::run(w1fr[t * 2 + h], xf[h], acc1[t]);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t == wave) own = acc1[t];
      else *reinterpret_cast<f32x4*>(xslot(i & 1, t, wave)) = acc1[t];
    }
    // step i + 1's operands (issued one step ago) have landed: everything but this step's own six operations is complete
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int i = 0; i <= nst; i += 3) {
    step(i, std::integral_constant<int, 0>{});
    step(i + 1, std::integral_constant<int, 1>{});
    step(i + 2, std::integral_constant<int, 2>{});
  }
}

template <typename T>
int launch_pair(PairArgs a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_pair128_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kPairLds);
    attr_set = true;
  }
  a.S = (a.M + kPairP - 1) / kPairP;
  a.n_wg = pair_num_cus();
  if (a.n_wg > a.S) a.n_wg = a.S;
  hipLaunchKernelGGL((bottleneck_pair128_kernel<T>), dim3(a.n_wg), dim3(512), kPairLds, stream, a);
  return dp_check_launch("bottleneck_pair128_kernel");
}

const char* pair_unsupported(const dp_pair_params* p) {
  if (!(p->dtype == DP_BF16 || p->dtype == DP_F16)) return "16-bit storage only (fp32 parity mode runs the two layers one by one)";
  const bool r3 = p->Cmid == 128 && p->Cout == 512 && p->Cmid_next == 128, r4 = p->Cmid == 256 && p->Cout == 1024 && p->Cmid_next == 256;
  if (!r3 && !r4) return "channel counts 128 -> 512 -> 128 (the plain blocks of res3) or 256 -> 1024 -> 256 (res4) only";
  if (r4 && dp_policy().pair256 == 0) return "the res4 form is off by default (policy key pair256): slower than the two launches, profiles/r6_pair256_experiments.txt";
  if (p->Kpad3 != p->Cmid || p->Kpad1n != p->Cout) return "packed K lengths = the channel counts expected";
  if (p->M < 0 || (p->M + 64) * p->Cout * 2 >= (1ll << 31)) return "tensor too large for 32-bit buffer offsets (caller chunks the batch)";
  return nullptr;
}

}  // namespace

extern "C" int dp_bottleneck_pair_supported(const dp_pair_params* p) {
  if (!p) return 0;
  return pair_unsupported(p) == nullptr ? 1 : 0;
}

extern "C" int dp_bottleneck_pair_nhwc(const dp_pair_params* p, dp_stream_t stream) {
  DP_REQUIRE(p != nullptr, "dp_bottleneck_pair_nhwc: null params");
  const char* why = pair_unsupported(p);
  if (why) return dp_fail(DP_ERR_UNSUPPORTED, "dp_bottleneck_pair_nhwc: %s", why);
  if (p->M == 0) return DP_OK;
  DP_REQUIRE(p->t2 && p->residual && p->out && p->next_t1 && p->w3 && p->w1n && p->b3 && p->b1n, "dp_bottleneck_pair_nhwc: null pointer");
  DP_REQUIRE((((uintptr_t)p->t2 | (uintptr_t)p->residual | (uintptr_t)p->out | (uintptr_t)p->next_t1 | (uintptr_t)p->w3 | (uintptr_t)p->w1n) & 15) == 0,
             "dp_bottleneck_pair_nhwc: tensors must be 16-byte aligned");
  if (p->Cmid == 256) return dp_pair256_launch(p, stream);      // res4: both matrices streamed through LDS (dp_pair256.hip)
  PairArgs a;
  a.t2 = p->t2; a.res = p->residual; a.out = p->out; a.t1n = p->next_t1;
  a.w3 = p->w3; a.w1 = p->w1n; a.b3 = p->b3; a.b1 = p->b1n;
  a.M = (int)p->M; a.S = a.n_wg = 0;
  a.t2_bytes = (unsigned)(p->M * 256); a.out_bytes = (unsigned)(p->M * 1024); a.t1n_bytes = (unsigned)(p->M * 256);
  hipStream_t s = as_stream(stream);
  return p->dtype == DP_BF16 ? launch_pair<uint16_t>(a, s) : launch_pair<f16_t>(a, s);
}
