// conv3 (+ residual, ReLU) of a bottleneck block and conv1 (+ ReLU) of the NEXT block in ONE launch, for the plain blocks of res4
// (256 -> 1024 -> 256 channels; /root/reference/detectron2/modeling/backbone/resnet.py:199-205 and :192-193 of the following block;
// resnet.py:659-688 builds 5 such pairs for R50, 22 for R101):
//
//     out = relu(conv3_1x1(t2) + b3 + residual)        256 -> 1024
//     t1' = relu(conv1'_1x1(out) + b1')               1024 -> 256
//
// Two launches write the block output (69 MB at batch 8) and read it straight back: 241 MB for two layers; chained it is written once:
// 172 MB. Unlike res3's pair (dp_pair.hip) the two weight matrices (512 KiB each) fit neither the register file nor LDS, so here they
// STREAM: a workgroup owns a job of up to 128 pixels whose t2 fragments and conv1' accumulators stay in registers, and walks the 1024
// middle channels in 16 chunks of 64 - conv3's cout chunk IS conv1''s K chunk. The weight fragments come straight from L2 into registers
// (a 1 KiB tile of the packed matrices is one MFMA A fragment) through a rolling queue of 16 loads in flight: no LDS staging, no
// LDS-DMA. (First form, measured and dropped - profiles/r6_pair256_experiments.txt: both chunks double-buffered in LDS by LDS-DMA, 160 KiB:
// 16 pieces per wave and chunk stall a lone wave's MFMA issue, and three short fragment loops per chunk each expose an LDS latency: 5230
// cycles per chunk for 2048 MFMA cycles.)
//   * four waves, one per SIMD (512 registers): wave (b, s) owns pixel block b (up to 4 tiles of 16 pixels) and the cout PAIR s of a
//     chunk in conv3 (physical row tiles 2 s, 2 s + 1 = 32 logical channels = one K step of conv1', pack.py's row permutation makes
//     the epilogue registers that step's B fragment) and the cout HALF s in conv1'. The B fragment of the other K step of the chunk
//     comes from the partner wave (b, 1 - s) through LDS (4 KiB per wave and chunk, parity double-buffered);
//   * per chunk and wave 128 MFMAs (64 + 64), 32 weight-fragment loads of 1 KiB + 4 exchange reads, one barrier;
//   * the weights of a job come from L2 twice (the two pixel blocks read the same fragments: 2 MiB per job, the second read mostly an L1 hit);
//   * summation: conv3 one chain over K = 256 from the bias; conv1' one chain per output from the bias, K step 2 c + s before 2 c + 1 - s
//     in every chunk c (own fragment first, partner's second): fixed orders - a call site runs here for every batch size or never.
// Jobs: floor(tiles / 8 / CUs) rounds of full 8-tile jobs, then the remaining tiles spread evenly over up to one job per CU.
#include "dp_common.h"
#include "dp_mma.h"

#ifndef DP_PAIR4_EXP
#define DP_PAIR4_EXP 0     // diagnostic builds: 16 = in-kernel phase stamps
#endif

namespace {

int pair4_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

struct Pair4Args {
  const void* t2;
  const void* res;
  void* out;
  void* t1n;
  const void* w3;
  const void* w1;
  const float* b3;
  const float* b1;
  int M;
  int n_full, n_tail, tail_q, tail_rem;      // jobs: n_full of 8 tiles, then n_tail of tail_q (+ 1 for the first tail_rem) tiles
  unsigned t2_bytes, out_bytes, t1n_bytes;
  unsigned long long* dbg;
};

constexpr int kP4T = 4;                                  // pixel tiles per wave block
constexpr int kP4Xch = 0;                                // exchange [parity][wave][tile] of 1 KiB
constexpr int kP4Lds = 2 * 4 * kP4T * 1024;              // 32 KiB
constexpr int kP4Q = 16;                                 // weight fragments in flight (a divisor of the 32 fragments of an iteration)
constexpr unsigned kP4Oob = 0x80000000u;

template <typename T>
__global__ __launch_bounds__(256, 1) void bottleneck_pair256_kernel(const Pair4Args p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int TT = kP4T, Q = kP4Q;
  static_assert(32 % Q == 0, "the fragment queue keeps its register of a slot across iterations");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int blk = wave >> 1, s = wave & 1;

  // ---- this workgroup's job: pixel tiles [tile0, tile0 + nt), this wave's block [bt0, bt0 + ntb)
  int tile0, nt;
  {
    const int id = blockIdx.x;
    if (id < p.n_full) { tile0 = id * 8; nt = 8; }
    else {
      const int k = id - p.n_full;
      nt = p.tail_q + (k < p.tail_rem ? 1 : 0);
      tile0 = p.n_full * 8 + k * p.tail_q + min(k, p.tail_rem);
    }
  }
  const int nb0 = (nt + 1) >> 1;                          // block 0 takes the larger half
  const int bt0 = tile0 + (blk ? nb0 : 0), ntb = blk ? nt - nb0 : nb0;

  const __amdgpu_buffer_rsrc_t rs_t2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t2), 0, p.t2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t1n = __builtin_amdgcn_make_buffer_rsrc(p.t1n, 0, p.t1n_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), 0, 1024u * 256u * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 256u * 1024u * 2u, 0x00020000);

  // ---- weight fragments straight from L2 into registers: a 1 KiB tile of the packed matrices (16 rows x 64 B of K) IS an A fragment -
  //      lane (row fr, chunk fq) reads its 16 bytes at fr * 64 + fq * 16. The 32 fragments of an iteration, in the order they are used:
  //        slots  0 ..  7   conv1'(c), K step 2 c + s (the wave's own B fragments):  W1 row tile 8 s + j,     plane 2 c + s
  //        slots  8 .. 15   conv1'(c), K step 2 c + 1 - s (the partner's):           W1 row tile 8 s + j,     plane 2 c + 1 - s
  //        slots 16 .. 31   conv3(c + 1), K step kk, row tile r:                     W3 row tile 4 (c + 1) + 2 s + r, plane kk
  //      loaded Q - 1 slots ahead of their use into a ring of Q registers (the queue runs across iterations: Q divides 32).
  const int wlane = fr * 64 + fq * 16;
  u32x4 wf[Q];
  auto wload = [&](int c, auto slot_) __attribute__((always_inline)) {
    constexpr int slot = decltype(slot_)::value;
    if constexpr (slot < 16) {
      constexpr int j = slot & 7, other = slot >> 3;
      const int cc = max(min(c, 15), 0);
      const int pl = other ? 1 - s : s;
      wf[slot % Q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, wlane + j * 32768, (8 * s * 32 + 2 * cc + pl) * 1024, 0));
    } else {
      constexpr int kk = (slot - 16) >> 1, r = (slot - 16) & 1;
      const int cc = min(c + 1, 15);
      wf[slot % Q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, wlane + (r * 8 + kk) * 1024, (4 * cc + 2 * s) * 8192, 0));
    }
  };

  // ---- t2 fragments of this wave's pixel tiles: B[k][n] = channel 32 kk + 8 fq + j of pixel fr, in registers for the whole job
  u32x4 t2f[TT][8];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int m = (bt0 + t) * 16 + fr;
    const int off = (t < ntb && m < p.M) ? m * 512 + fq * 16 : (int)kP4Oob;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) t2f[t][kk] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t2, off == (int)kP4Oob ? off : off + kk * 64, 0, 0));
  }
  // conv1' accumulators start from its bias: row tile 8 s + j, register e = logical cout (i >> 2) * 64 + ((i & 3) >> 1) * 32 + fq * 8 + (i & 1) * 4 + e
  f32x4 acc1[TT][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = 8 * s + j;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b1 + (i >> 2) * 64 + ((i & 3) >> 1) * 32 + fq * 8 + (i & 1) * 4);
#pragma unroll
    for (int t = 0; t < TT; ++t) acc1[t][j] = bv;
  }
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+a"(t2f[t][kk]));      // register classes pinned once (dp_conv_wq.hip)

  // residual values and conv3 bias of a chunk: this lane's 8 channels 64 c + 32 s + 8 fq .. + 7 of its pixel in every tile
  auto res_issue = [&](int c, u32x4 (&rv)[TT], f32x4 (&bv)[2]) __attribute__((always_inline)) {
    const int ch = 64 * c + 32 * s + 8 * fq;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int m = (bt0 + t) * 16 + fr;
      rv[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (c < 16 && t < ntb && m < p.M) ? m * 2048 + ch * 2 : (int)kP4Oob, 0, 0));
    }
    const float* bp = p.b3 + (c < 16 ? ch : 0);
    bv[0] = *reinterpret_cast<const f32x4*>(bp);
    bv[1] = *reinterpret_cast<const f32x4*>(bp + 4);
  };
  // the epilogue of conv3(c): bias (the accumulators' initial value), residual, ReLU, one rounding: the block output (stored) and, as it
  // stands in the registers, the B fragment of K step 2 c + s of conv1' (kept, and handed to the partner wave through LDS)
  u32x4 xf[TT];
  auto conv3_epilogue = [&](int c, f32x4 (&acc3)[TT][2], const u32x4 (&rv)[TT]) __attribute__((always_inline)) {
    const int ch = 64 * c + 32 * s + 8 * fq;
    unsigned char* xb = smem + kP4Xch + (((c & 1) * 4 + wave) * TT) * 1024 + lane * 16;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = acc3[t][0][k]; v[4 + k] = acc3[t][1][k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += Elem<T>::unpack(rv[t][k] & 0xffffu);
        v[2 * k + 1] += Elem<T>::unpack(rv[t][k] >> 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) xf[t][k] = Elem<T>::pack2(fmaxf(v[2 * k], 0.f), fmaxf(v[2 * k + 1], 0.f));
      const int m = (bt0 + t) * 16 + fr;
      __builtin_amdgcn_raw_buffer_store_b128(xf[t], rs_out, (t < ntb && m < p.M) ? m * 2048 + ch * 2 : (int)kP4Oob, 0, 0);
      *reinterpret_cast<u32x4*>(xb + t * 1024) = xf[t];
    }
  };

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(k) if constexpr (DP_PAIR4_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  unsigned long long tl = (DP_PAIR4_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;

  // ---- one iteration = the 32 fragment slots; FROM = 16 runs conv3 only (the prologue: "iteration -1" computes conv3(0))
  u32x4 rv[TT];
  f32x4 bv[2];
  u32x4 oth[TT];
  auto run_slots = [&](int c, auto from_) __attribute__((always_inline)) {
    constexpr int FROM = decltype(from_)::value;
    f32x4 acc3[TT][2];
#pragma unroll
    for (int t = 0; t < TT; ++t) { acc3[t][0] = bv[0]; acc3[t][1] = bv[1]; }
    static_for<FROM, 32>([&](auto sl) {
      constexpr int slot = decltype(sl)::value;
      // the queue: slot + Q - 1 of this iteration, or of the next one
      if constexpr (slot + Q - 1 < 32) wload(c, std::integral_constant<int, (slot + Q - 1) % 32>{});
      else wload(c + 1, std::integral_constant<int, (slot + Q - 1) % 32>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (slot < 8) {
#pragma unroll
        for (int t = 0; t < TT; ++t) Mma<T>::run(wf[slot % Q], xf[t], acc1[t][slot]);
      } else if constexpr (slot < 16) {
#pragma unroll
        for (int t = 0; t < TT; ++t) Mma<T>::run(wf[slot % Q], oth[t], acc1[t][slot - 8]);
      } else {
        constexpr int kk = (slot - 16) >> 1, r = (slot - 16) & 1;
#pragma unroll
        for (int t = 0; t < TT; ++t) Mma<T>::run(wf[slot % Q], t2f[t][kk], acc3[t][r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    DP_STAMP(1)
    if (c + 1 < 16) conv3_epilogue(c + 1, acc3, rv);
    res_issue(c + 2, rv, bv);        // (the registers the epilogue has just consumed; used a whole iteration from now)
    DP_STAMP(2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DP_STAMP(3)
  };
  // ---- prologue
  res_issue(0, rv, bv);
  static_for<16, 16 + Q - 1>([&](auto sl) { wload(-1, sl); });
  run_slots(-1, std::integral_constant<int, 16>{});
  DP_STAMP(0)
  // iteration c: [barrier(c) passed: the xf(c) of every wave are in LDS]  conv1'(c): own K step (xf is still in the registers), then the
  // partner's;  conv3(c + 1) + its epilogue (stores, xf(c + 1) -> registers + LDS);  barrier(c + 1)
  for (int c = 0; c < 16; ++c) {
    const unsigned char* xb = smem + kP4Xch + (((c & 1) * 4 + (wave ^ 1)) * TT) * 1024 + lane * 16;
#pragma unroll
    for (int t = 0; t < TT; ++t) oth[t] = *reinterpret_cast<const u32x4*>(xb + t * 1024);
    run_slots(c, std::integral_constant<int, 0>{});
  }

  // ---- t1' = relu(acc1) (the bias is in it), row tiles 8 s + 2 jj, + 1 = 8 consecutive couts per lane: one 16-byte store per pixel tile and pair
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int m = (bt0 + t) * 16 + fr;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int i = 8 * s + 2 * jj;
      const int c1 = (i >> 2) * 64 + ((i & 3) >> 1) * 32 + fq * 8;
      const f32x4 a = acc1[t][2 * jj], bq = acc1[t][2 * jj + 1];
      const u32x4 pk = {Elem<T>::pack2(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f)), Elem<T>::pack2(fmaxf(a[2], 0.f), fmaxf(a[3], 0.f)),
                        Elem<T>::pack2(fmaxf(bq[0], 0.f), fmaxf(bq[1], 0.f)), Elem<T>::pack2(fmaxf(bq[2], 0.f), fmaxf(bq[3], 0.f))};
      __builtin_amdgcn_raw_buffer_store_b128(pk, rs_t1n, (t < ntb && m < p.M) ? m * 512 + c1 * 2 : (int)kP4Oob, 0, 0);
    }
  }
  if constexpr (DP_PAIR4_EXP & 16) {
    DP_STAMP(4)
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 5; ++k) p.dbg[(blockIdx.x * 4 + wave) * 8 + k] = ph[k];
      p.dbg[(blockIdx.x * 4 + wave) * 8 + 5] = nt;
    }
  }
#undef DP_STAMP
}

template <typename T>
int launch_pair4(Pair4Args a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_pair256_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kP4Lds);
    attr_set = true;
  }
  const int cus = pair4_num_cus();
  const int tiles = (a.M + 15) / 16;
  a.n_full = (tiles / (8 * cus)) * cus;                 // whole rounds of full jobs
  const int rest = tiles - a.n_full * 8;
  a.n_tail = rest < cus ? rest : cus;                   // rest < 8 * cus: at most 8 tiles per tail job
  a.tail_q = a.n_tail ? rest / a.n_tail : 0;
  a.tail_rem = a.n_tail ? rest - a.tail_q * a.n_tail : 0;
  a.dbg = nullptr;
#if DP_PAIR4_EXP & 16
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_full + a.n_tail;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 8 * 4 * 8192);
  a.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 8 * 4 * nblk, stream);
#endif
  hipLaunchKernelGGL((bottleneck_pair256_kernel<T>), dim3(a.n_full + a.n_tail), dim3(256), kP4Lds, stream, a);
#if DP_PAIR4_EXP & 16
  {
    static int shown = 0;
    if (shown++ == 4) {
      (void)hipStreamSynchronize(stream);
      unsigned long long* hbuf = (unsigned long long*)malloc(sizeof(unsigned long long) * 32 * nblk);
      (void)hipMemcpy(hbuf, dbg, sizeof(unsigned long long) * 32 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w) {
        double sum[5] = {0, 0, 0, 0, 0}, n = 0;
        for (int b = 0; b < nblk; ++b) { for (int k = 0; k < 5; ++k) sum[k] += (double)hbuf[(b * 4 + w) * 8 + k]; n += 1; }
        fprintf(stderr, "pair256 wave %d: cycles per job: prologue %.0f | per chunk: matrix loop %.0f  conv3 epilogue %.0f  waits + barrier %.0f | final stores %.0f  (%d jobs: %d full + %d of %d (+1 for %d) tiles)\n",
                w, sum[0] / n, sum[1] / n / 16, sum[2] / n / 16, sum[3] / n / 16, sum[4] / n, nblk, a.n_full, a.n_tail, a.tail_q, a.tail_rem);
      }
      free(hbuf);
    }
  }
#endif
  return dp_check_launch("bottleneck_pair256_kernel");
}

}  // namespace

// used by dp_bottleneck_pair_nhwc (dp_pair.hip): the 256 -> 1024 -> 256 form
int dp_pair256_launch(const dp_pair_params* p, dp_stream_t stream) {
  Pair4Args a;
  a.t2 = p->t2; a.res = p->residual; a.out = p->out; a.t1n = p->next_t1;
  a.w3 = p->w3; a.w1 = p->w1n; a.b3 = p->b3; a.b1 = p->b1n;
  a.M = (int)p->M;
  a.n_full = a.n_tail = a.tail_q = a.tail_rem = 0;
  a.t2_bytes = (unsigned)(p->M * 512); a.out_bytes = (unsigned)(p->M * 2048); a.t1n_bytes = (unsigned)(p->M * 512);
  a.dbg = nullptr;
  hipStream_t s = as_stream(stream);
  return p->dtype == DP_BF16 ? launch_pair4<uint16_t>(a, s) : launch_pair4<f16_t>(a, s);
}
