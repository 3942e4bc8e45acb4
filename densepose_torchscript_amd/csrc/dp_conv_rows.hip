// Row-streaming, K-split, weight-stationary 3x3 convolution (16-bit storage). Behind dp_conv2d_nhwc (dp_conv.hip, kernel class 7).
//
// The layers it is written for are the 512-channel 3x3 convolutions of the DensePose head on R x 28 x 28 ROI maps
// (/root/reference/densepose/modeling/roi_heads/v1convx.py:44-59, deeplab.py:64-74: body_conv_fcn1..8) and res5's conv2
// (/root/reference/detectron2/modeling/backbone/resnet.py:195-197): K = 9 x 512 = 4608, i.e. 4.7 MB of weights that the LDS-ring
// kernels re-stage through LDS for every pixel tile, one barrier per 64-byte K plane (144 plane steps per tile).
//
// Here nothing about the weights moves after the prologue and the pixel operand is read from LDS once per SIX MFMAs:
//   * a workgroup owns NC = NCT x 16 output channels (32 for Cin = 512, 64 for Cin = 256) and its 8 waves split the K axis by INPUT
//     CHANNELS: wave w holds the weights of channels [w * Cin / 8, (w + 1) * Cin / 8) x 9 taps x NC couts = 36 MFMA A fragments =
//     144 VGPRs for the whole launch;
//   * work = column strips of 16 output pixels, walked top to bottom ONE INPUT ROW per step: input row q contributes to the output
//     rows q - 1, q, q + 1 (kernel rows 2, 1, 0), so every pixel fragment (32 channels of 16 pixels at one column tap) read from LDS
//     feeds 3 x NCT MFMAs into three live accumulator sets that rotate; the output row q - 1 is complete after step q;
//   * a wave only ever needs ITS channel slice of the input: each wave keeps a private ring of row slices (19 pixels x Cin / 8
//     channels) that it fills by LDS-DMA D rows ahead and waits for with its own counted vmcnt - no barrier guards the operand ring;
//   * the 8 partial sums of a finished output row meet in an LDS staging buffer (one barrier per step); the reduction, bias,
//     activation and the 16-byte stores rotate over the waves (the reducing wave's SIMD partner has the matrix pipe meanwhile);
//   * strips are cut from the CONCATENATED columns of G = 16 / gcd(W, 16) images: a 28-wide ROI map is 16 + 12 columns, so a plain
//     strip walk wastes an eighth of every MFMA's columns; strip k of a group instead covers virtual columns [16 k, 16 k + 16), at
//     most two segments from two consecutive images, each staged with its own zero halo (the two halo pixels between them are one
//     shared zero pixel: 19 staged pixels, as for a plain strip's 16 + 2).
// Per-pixel arithmetic: partial sum of wave w = kernel row major, then 32-channel block, then kernel column (one fp32 MFMA chain);
// the 8 partials are added in wave order. It does not depend on the strip, the lane or the workgroup a pixel lands in, nor on the
// number of images: a frame's result is the same whatever else is in the batch. It is NOT the K order of the LDS-ring kernels, so a
// layer must run here for every batch size - dp_conv_rows_ok() has no size thresholds.
#include "dp_common.h"
#include "dp_mma.h"
#include <stdlib.h>

#ifndef DP_ROWS2_OPT
#define DP_ROWS2_OPT 0    // experiments on conv3x3_rows2_kernel (results stay exact): 1 fragment reads two batches ahead, 2 no s_setprio, 4 waves 4 .. 7 issue their
                          // LDS-DMA pieces in the middle of the matrix block, 8 waves 0 .. 3 store the reduced row before they issue their pieces
#endif
#ifndef DP_ROWS_EXP
#define DP_ROWS_EXP 0     // diagnostic builds (timing only, results garbage): 1 no LDS-DMA in the loop, 2 no reduction, 4 no MFMAs, 8 no fragment reads,
                          // 16 in-kernel phase stamps (s_memtime sums per wave, printed by the fifth launch)
#endif

namespace {

int rows_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

struct RowsArgs {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  const int* n_dev;            // device-side count of live images (dp_conv_params.n_dev) or null
  int N, H, W, relu;
  int opitch;                  // elements between consecutive output pixels
  int kpad;
  int G, SPG;                  // images per strip group, strips per group (G * W == 16 * SPG)
  int n_slices, n_pg;          // cout slices, pixel groups (grid = n_slices * n_pg workgroups)
  unsigned in_bytes, out_bytes;
  unsigned long long* dbg;     // diagnostic builds (-DDP_ROWS_EXP=16): per-wave phase cycle sums
  int lockstep;                // conv3x3_rows2_kernel: 1 = all eight waves on one schedule (A/B knob), 0 = the two halves in opposite phases
  int n_base;                  // first image of this launch in the caller's tensor (launches are cut into chunks below 2 GiB): *n_dev counts from there
};

template <int N>
__device__ __forceinline__ void rows_wait_vm() {
  static_assert(N == 6 || N == 9, "vmcnt immediate");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

// One step of a workgroup = one input row q of one strip; a workgroup's output rows [a, b) of the linearised (strip, row) space are
// walked segment by segment (a segment = the rows of one strip): input rows r_lo - 1 .. r_hi, r_hi exclusive end of the output rows.
struct RowsIt {
  int strip, q, r_lo, r_hi;
};

template <typename T, int CIN, int NCT>
__global__ __launch_bounds__(512, 2) void conv3x3_rows_kernel(const RowsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert((CIN == 512 && NCT == 2) || (CIN == 256 && NCT == 4), "36 weight fragments per wave");
  constexpr int KPW = CIN / 8;                  // input channels of a wave (its K part)
  constexpr int NCB = KPW / 32;                 // 32-channel blocks (MFMA K steps) per tap
  constexpr int PPW = KPW * 2 + 32;             // bytes per staged pixel of a wave's row slice: chunk c of pixel j sits on 16-byte slot
                                                // (j * PPW / 16 + c) mod 16 - the ds_read_b128 lane groups of a fragment hit 16 different
                                                // slots for 16 consecutive pixels (PPW / 16 = 10 or 6)
  constexpr int NPX = 19;                       // staged pixels per row: 16 + 2 halo (+ 1: the shared zero pixel between two segments)
  constexpr int NP = (NPX * PPW + 1023) / 1024; // LDS-DMA pieces (1 KiB wave instructions) per row slice
  constexpr int ROWB = NP * 1024;
  constexpr int D = 3, NSLOT = D + 1;           // rows in flight ahead of the one being consumed
  constexpr int RING = 8 * NSLOT * ROWB;
  constexpr int STGB = 8 * NCT * 1024;          // staging bytes per parity: [wave][cout tile][lane] x 16 B
  constexpr int NF = NCB * 3;                   // pixel fragments per step: (channel block, column tap)
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;      // the slices of a pixel group run on one XCD (they read the same rows)
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  const int n_live = p.n_dev ? min(max(*p.n_dev - p.n_base, 0), p.N) : p.N;
  // whole groups of G images, then the strips that the columns of the remaining images reach (a strip is SW = G * W / SPG pixels)
  const int n_strips = (n_live / p.G) * p.SPG + ((n_live % p.G) * p.W * p.SPG + p.G * p.W - 1) / (p.G * p.W);
  const long long TR = (long long)n_strips * p.H;
  const int wa = (int)(TR * pg / p.n_pg), wb = (int)(TR * (pg + 1) / p.n_pg);
  if (wb <= wa) return;

  // ---- this wave's weights: NCT cout tiles x NCB channel blocks x 9 taps, in registers for the whole launch. Physical rows of the
  // packed matrix in natural order: tile ct row i of the slice = physical row slice * NCT * 16 + ct * 16 + i, which pack.py's row
  // permutation maps to logical cout (ct >> 1) * 32 + (i >> 2) * 8 + (ct & 1) * 4 + (i & 3) of the slice - so lane (fr, fq) ends up
  // with 8 CONSECUTIVE output channels of pixel fr in each pair of cout tiles (the register epilogue of dp_conv.hip).
  u32x4 wfr[NCT * NCB * 9];
  {
    const int n_planes = p.kpad * 2 / 64;
    const unsigned char* __restrict__ wp = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int cbl = 0; cbl < NCB; ++cbl)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          wfr[(ct * NCB + cbl) * 9 + t] =
              *reinterpret_cast<const u32x4*>(wp + dp_wtile_off(slice * NCT * 16 + ct * 16 + fr, (wave * NCB + cbl) * 9 + t, fq, n_planes));
  }
  const int cout0 = slice * NCT * 16;           // logical cout base of the slice
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const int in_row = p.W * CIN * 2, out_row = p.W * p.opitch * 2;

  // strip -> its two segments: images img0 (columns c00 .. c00 + len0 - 1) and img0 + 1 (columns 0 .. 15 - len0)
  auto decode = [&](int strip, int& img0, int& c00, int& len0) __attribute__((always_inline)) {
    const int grp = strip / p.SPG, k = strip - grp * p.SPG;
    const int x0 = k * 16, ig = x0 / p.W;
    c00 = x0 - ig * p.W;
    img0 = grp * p.G + ig;
    len0 = min(16, p.W - c00);
  };

  // ---- fetch side: this lane's part of the NP pieces of a row slice. Lane l of piece pc lands on ring byte pc * 1024 + l * 16 =
  // staged pixel dj, byte dwb of its PPW bytes (both fixed for the launch); which image column that is depends on the strip.
  static_assert(NP <= 3, "f_boff");
  int f_boff[3];         // byte offset of (image, row 0, column, this wave's channels) of this lane's piece part, or OOB. (A literal
                         // bound: with the dependent `NP` hipcc's HOST pass drops the kernel - an array of dependent size captured by
                         // a lambda - and the library fails to load with the kernel's stub undefined. No diagnostic.)
  auto setup_fetch = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int o = pc * 1024 + lane * 16;        // (recomputed per strip rather than kept: two registers per piece)
      const int j = o / PPW, wbyte = o - j * PPW;
      const bool seg1 = j >= len0 + 2;
      const int col = seg1 ? j - (len0 + 2) : c00 - 1 + j;
      const int img = img0 + (seg1 ? 1 : 0);
      const bool ok = j < NPX && wbyte < KPW * 2 && (unsigned)col < (unsigned)p.W && img < n_live && (!seg1 || len0 < 16);
      f_boff[pc] = ok ? ((img * p.H * p.W + col) * CIN + wave * KPW) * 2 + wbyte : OOB;
    }
  };
  unsigned char* const ring_w = smem + wave * (NSLOT * ROWB);
  auto fetch = [&](const RowsIt& it, bool live, int slot, bool issue) __attribute__((always_inline)) {
    const bool row_ok = live && (unsigned)it.q < (unsigned)p.H;     // rows -1 and H are zero padding: nothing is read (nor computed)
    const int roff = it.q * in_row;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc)
      if (issue)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(ring_w + slot * ROWB + pc * 1024), 16, row_ok ? f_boff[pc] + roff : OOB, 0, 0, 0);
  };

  // ---- compute side: lane (fr, fq) = output pixel fr of the strip; its staged pixel at column tap dx is fr + dx (+ 1 in the
  // second segment, behind the shared zero pixel)
  // ---- reduce side (waves 4 .. 7): lane l adds up the eight partial sums of TWO consecutive output channels of pixel (l & 31) >> 1:
  // 32 lanes read 256 contiguous bytes of a partial tile (ds_read_b64, conflict free), wave 4 + j takes the cout tiles
  // (j & 1) * ND .. + ND - 1 and the channel octets fq = 2 * (j >> 1) + (l >> 5) of every pixel.
  constexpr int ND = NCT / 2;                    // reduction duties (cout tiles) per reducing wave
  const bool is_y = wave >= 4;
  const int rj = wave & 3;
  const int r_px = (lane & 31) >> 1, r_fq = 2 * (rj >> 1) + (lane >> 5), r_half = lane & 1;
  const int r_ct0 = (rj & 1) * ND;
  const int r_lds = (r_fq * 16 + r_px) * 16 + r_half * 8;                 // byte of this lane's pair inside a [64 lanes x 16 B] partial tile
  float r_bias[ND][2];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int ct = r_ct0 + d, co = cout0 + (ct >> 1) * 32 + r_fq * 8 + (ct & 1) * 4 + r_half * 2;
    r_bias[d][0] = p.bias[co];
    r_bias[d][1] = p.bias[co + 1];
  }
  int c_frag = 0, r_obase = OOB;
  auto setup_comp = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
    c_frag = (fr + (fr >= len0 ? 1 : 0)) * PPW + fq * 16;
    const bool in1 = r_px >= len0;
    const int img = img0 + (in1 ? 1 : 0), col = in1 ? r_px - len0 : c00 + r_px;
    r_obase = img < n_live ? ((img * p.H * p.W + col) * p.opitch + cout0 + r_fq * 8 + r_half * 2) * 2 : OOB;
  };

  auto seg_init = [&](RowsIt& it, int strip) __attribute__((always_inline)) {
    it.strip = strip;
    it.r_lo = max(wa - strip * p.H, 0);
    it.r_hi = min(wb - strip * p.H, p.H);
    it.q = it.r_lo - 1;
  };
  const int s_first = wa / p.H, s_last = (wb - 1) / p.H;
  const int n_steps = (wb - wa) + 2 * (s_last - s_first + 1);

  RowsIt it_f, it_c;
  seg_init(it_f, s_first);
  seg_init(it_c, s_first);
  setup_fetch(s_first);
  setup_comp(s_first);
  int f_done = 0;     // steps whose fetch has been issued
  auto fetch_next = [&]() __attribute__((always_inline)) {
    const bool live = f_done < n_steps;
    int slot = f_done % NSLOT;
    fetch(it_f, live, slot, !(DP_ROWS_EXP & 1) || f_done < D);
    ++f_done;
    if (live) {
      if (it_f.q == it_f.r_hi) {
        if (it_f.strip < s_last) { seg_init(it_f, it_f.strip + 1); setup_fetch(it_f.strip); }
      } else {
        ++it_f.q;
      }
    }
  };
  // prologue: the first D rows
#pragma unroll
  for (int d = 0; d < D; ++d) fetch_next();

  f32x4 acc[3][NCT];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[a][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned char* const stg = smem + RING;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl = (DP_ROWS_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
#define DP_STAMP(k) if constexpr (DP_ROWS_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }

  // the finished row of a step: eight partial tiles per cout tile in staging buffer `par` -> bias, activation, 4-byte stores
  auto reduce = [&](int par, int t, int obase) __attribute__((always_inline)) {
    if constexpr (DP_ROWS_EXP & 2) return;
    const unsigned char* const sr = stg + par * STGB + r_lds;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const int ct = r_ct0 + d;
      f32x2 v[8];
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) v[w8] = *reinterpret_cast<const f32x2*>(sr + (w8 * NCT + ct) * 1024);
      float x0 = v[0][0], x1 = v[0][1];
#pragma unroll
      for (int w8 = 1; w8 < 8; ++w8) { x0 += v[w8][0]; x1 += v[w8][1]; }     // wave order: the summation order of a pixel is fixed
      x0 += r_bias[d][0];
      x1 += r_bias[d][1];
      if (p.relu) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
      __builtin_amdgcn_raw_buffer_store_b32(Elem<T>::pack2(x0, x1), rs_out, obase + t * out_row + ((ct >> 1) * 32 + (ct & 1) * 4) * 2, 0, 0);
      asm volatile("" ::: "memory");     // one duty's eight reads in flight at a time (registers)
    }
  };

  // Phase A of step s (no matrix instruction): bookkeeping, the row fetch D steps ahead, the reduction of the row the previous step
  // finished (waves 4 .. 7), the fragment reads of this step's row. Phase B: the step's MFMAs and its partial sums -> staging.
  // Waves 4 .. 7 run ONE BARRIER behind waves 0 .. 3, so on every SIMD one wave is in its phase B while its partner is in phase A.
  u32x4 bf[NF];
  // what the two previous steps left to reduce: the row of step s - 2 is complete in the staging buffer when phase A of step s starts
  bool p1_emit = false, p2_emit = false;
  int p1_t = 0, p2_t = 0, p1_ob = OOB, p2_ob = OOB;
  auto phase_a = [&](int s) __attribute__((always_inline)) {
    DP_STAMP(7)
    if (s > 0) {
      p2_emit = p1_emit; p2_t = p1_t; p2_ob = p1_ob;
      p1_t = it_c.q - 1;
      p1_emit = p1_t >= it_c.r_lo;
      p1_ob = r_obase;
      if (it_c.q == it_c.r_hi) {       // on to this step's row
        if (it_c.strip < s_last) { seg_init(it_c, it_c.strip + 1); setup_comp(it_c.strip); }
      } else {
        ++it_c.q;
      }
    }
    fetch_next();                                  // row of step s + D -> the slot step s - 1 has finished reading
    DP_STAMP(0)
    if (s >= 2 && p2_emit && (((s - 2) & 1) != 0) == is_y) reduce((s - 2) % 3, p2_t, p2_ob);
    DP_STAMP(1)
    if (!(DP_ROWS_EXP & 1)) rows_wait_vm<NP * D>(); // the row of step s has landed (only the D younger rows - and this phase's stores - may be in flight)
    if ((unsigned)it_c.q < (unsigned)p.H) {
      const unsigned char* const row = ring_w + (s % NSLOT) * ROWB + c_frag;
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        if constexpr (DP_ROWS_EXP & 8) bf[f] = u32x4{(unsigned)lane, (unsigned)f, 1u, 2u};
        else bf[f] = *reinterpret_cast<const u32x4*>(row + (f % 3) * PPW + (f / 3) * 64);
      });
    }
    DP_STAMP(2)
  };
  // accumulator roles fixed at compile time: old = acc[(PH + 2) % 3] (output row q - 1, kernel row 2, complete after this step),
  // mid = acc[PH] (row q, kernel row 1), fresh = acc[(PH + 1) % 3] (row q + 1, kernel row 0, starts from zero). Three passes over the
  // step's fragments, one per role, the finished row first: its partial sums go to the staging buffer before the fresh row's
  // accumulators are born, so only two of the three sets are live at any time. Per output pixel the order of the products is
  // kernel row, 32-channel block, kernel column.
  auto phase_b = [&](auto ph_c, int s) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int A_OLD = (PH + 2) % 3, A_MID = PH, A_NEW = (PH + 1) % 3;
    const bool emit = it_c.q - 1 >= it_c.r_lo;
    const bool row_ok = (unsigned)it_c.q < (unsigned)p.H;
    unsigned char* const sw = stg + (s % 3) * STGB + wave * (NCT * 1024) + lane * 16;
    auto mma = [&](auto ff, auto ky_c, f32x4 (&a)[NCT]) __attribute__((always_inline)) {
      constexpr int f = decltype(ff)::value, ky = decltype(ky_c)::value;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        if constexpr (DP_ROWS_EXP & 4) { if (ct == 0) a[0][0] += __builtin_bit_cast(float, bf[f][0]); }
        else if constexpr (DP_ROWS_EXP & 32) Mma<T>::run(wfr[ct], bf[f], a[ct]);     // diagnostic: the same A operand for every MFMA of a cout tile
        else Mma<T>::run(wfr[(ct * NCB + f / 3) * 9 + ky * 3 + f % 3], bf[f], a[ct]);
      }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[A_NEW][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(1);      // the wave in its matrix phase goes first; its partner is in phase A
    if constexpr (NCT >= 4) {
      // three passes over the step's fragments, one per role, the finished row first: its partial sums leave for the staging buffer
      // before the fresh row's accumulators are born - two of the three sets live at a time (32 instead of 48 registers)
      if (row_ok) static_for<0, NF>([&](auto ff) { mma(ff, K2{}, acc[A_OLD]); });
      if (emit) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(sw + ct * 1024) = acc[A_OLD][ct];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (row_ok) {
        static_for<0, NF>([&](auto ff) { mma(ff, K1{}, acc[A_MID]); });
        static_for<0, NF>([&](auto ff) { mma(ff, K0{}, acc[A_NEW]); });
      }
    } else {
      // two cout tiles: the three roles interleaved per fragment, so that an accumulator is touched by every SIXTH matrix
      // instruction. One pass per role touched it every second one - and a wave alone on the matrix pipe then issued an MFMA every
      // ~24 cycles instead of every 16 (phase stamps: 36 MFMAs in 860 cycles): the result of an 8-pass MFMA is not back as a source
      // two instructions later.
      if (row_ok) static_for<0, NF>([&](auto ff) { mma(ff, K2{}, acc[A_OLD]); mma(ff, K1{}, acc[A_MID]); mma(ff, K0{}, acc[A_NEW]); });
      if (emit) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(sw + ct * 1024) = acc[A_OLD][ct];
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (DP_ROWS_EXP & 16) asm volatile("s_nop 0" :: "v"(acc[A_NEW][NCT - 1][0]), "v"(acc[A_MID][NCT - 1][0]));
    DP_STAMP(4)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the partial sums are in the staging buffer before the barrier
    DP_STAMP(5)
  };

  // unrolled by the three accumulator roles (a run-time role would keep all three sets live across the loop edge). ONE barrier per
  // step, and it sits in a different place for the two halves of the workgroup: waves 0 .. 3 run A B | A B | ..., waves 4 .. 7
  // A | B A | B A | ... - between two barriers a SIMD's older wave goes A -> B while its partner goes B -> A.
  auto unit = [&](auto ph_c, int s) __attribute__((always_inline)) {
    phase_a(s);
    if (is_y) { __builtin_amdgcn_s_barrier(); DP_STAMP(3) }
    phase_b(ph_c, s);
    if (!is_y) { __builtin_amdgcn_s_barrier(); DP_STAMP(6) }
  };
  for (int s = 0; s < n_steps; s += 3) {
    unit(std::integral_constant<int, 0>{}, s);
    if (s + 1 < n_steps) unit(std::integral_constant<int, 1>{}, s + 1);
    if (s + 2 < n_steps) unit(std::integral_constant<int, 2>{}, s + 2);
  }
  __builtin_amdgcn_s_barrier();                  // waves 4 .. 7 have written the last step's partial sums
  {
    // the rows of the last two steps (it_c still stands on the last step)
    const int tl_ = it_c.q - 1;
    if (n_steps >= 2 && p1_emit && (((n_steps - 2) & 1) != 0) == is_y) reduce((n_steps - 2) % 3, p1_t, p1_ob);
    if (tl_ >= it_c.r_lo && (((n_steps - 1) & 1) != 0) == is_y) reduce((n_steps - 1) % 3, tl_, r_obase);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy fetches behind the last step still target this workgroup's LDS
  if constexpr (DP_ROWS_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[(blockIdx.x * 8 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 8 + wave) * 16 + 8] = n_steps;
    }
  }
#undef DP_STAMP
}

#ifdef DP_EXPERIMENTS   // measured slower (profiles/r4_rows_kernel_experiments.txt): built by `make exp` only
// =====================================================================================================
// The same arithmetic as a CHAIN with no workgroup barrier in the loop (round 4, second form). Phase stamps of the kernel above say a
// step is 36 MFMAs per wave against ~1000 cycles of per-step fixed cost - the barrier that guards the staging buffer, the reduction of
// the eight partial tiles, everything in lockstep behind that barrier. Here the eight K parts hand their accumulators ON instead:
// wave h starts the accumulators of an output row from what wave h - 1 has summed over channel parts 0 .. h - 1 (a 2 KiB tile through
// an LDS ring of HB slots per link, a produced / consumed counter pair per link polled with s_sleep), adds its own 576 K values over
// three steps and passes the row to wave h + 1; wave 7 adds the bias, applies the activation and stores. Every wave runs at its own
// pace, two rows behind its predecessor; the two waves of a SIMD drift apart by themselves and fill each other's issue gaps.
// Per output pixel: one fp32 MFMA chain over K in the order channel part, kernel row, 32-channel block, kernel column - fixed,
// independent of strip, lane, workgroup and batch (again NOT the LDS-ring kernels' order).
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(512, 2) void conv3x3_chain_kernel(const RowsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int CIN = 512, NCT = 2, KPW = 64, NCB = 2, PPW = KPW * 2 + 32, NPX = 19, NP = 3, ROWB = NP * 1024, D = 3, NSLOT = D + 1;
  constexpr int NF = NCB * 3;
  constexpr int RING = 8 * NSLOT * ROWB;
  constexpr int HB = 4, HSLOT = NCT * 1024;     // hand-over ring: HB tiles of [cout tile][lane] x 16 B per link
  constexpr int HAND = RING, ZERO = HAND + 7 * HB * HSLOT, TRASH = ZERO + HSLOT, FLAGS = TRASH + HSLOT;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // = K part h
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  const int n_live = p.n_dev ? min(max(*p.n_dev - p.n_base, 0), p.N) : p.N;
  // whole groups of G images, then the strips that the columns of the remaining images reach (a strip is SW = G * W / SPG pixels)
  const int n_strips = (n_live / p.G) * p.SPG + ((n_live % p.G) * p.W * p.SPG + p.G * p.W - 1) / (p.G * p.W);
  const long long TR = (long long)n_strips * p.H;
  const int wa = (int)(TR * pg / p.n_pg), wb = (int)(TR * (pg + 1) / p.n_pg);
  if (wb <= wa) return;

  u32x4 wfr[NCT * NCB * 9];
  {
    const int n_planes = p.kpad * 2 / 64;
    const unsigned char* __restrict__ wp = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int cbl = 0; cbl < NCB; ++cbl)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          wfr[(ct * NCB + cbl) * 9 + t] =
              *reinterpret_cast<const u32x4*>(wp + dp_wtile_off(slice * NCT * 16 + ct * 16 + fr, (wave * NCB + cbl) * 9 + t, fq, n_planes));
  }
  const int cout0 = slice * NCT * 16;
  float bias8[8];
  {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + cout0 + fq * 8), b1 = *reinterpret_cast<const f32x4*>(p.bias + cout0 + fq * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { bias8[k] = b0[k]; bias8[4 + k] = b1[k]; }
  }
  // counters + the zero tile (hand-over source of rows that start here or are never finished)
  volatile int* const flagP = reinterpret_cast<volatile int*>(smem + FLAGS);        // P[h]: rows wave h has handed on
  volatile int* const flagC = flagP + 8;                                             // C[h]: rows wave h has taken over
  if (tid < 16) flagP[tid] = 0;
  for (int i = tid; i < HSLOT / 16; i += 512) reinterpret_cast<u32x4*>(smem + ZERO)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const int in_row = p.W * CIN * 2, out_row = p.W * p.opitch * 2;

  auto decode = [&](int strip, int& img0, int& c00, int& len0) __attribute__((always_inline)) {
    const int grp = strip / p.SPG, k = strip - grp * p.SPG;
    const int x0 = k * 16, ig = x0 / p.W;
    c00 = x0 - ig * p.W;
    img0 = grp * p.G + ig;
    len0 = min(16, p.W - c00);
  };
  int f_boff[3];
  auto setup_fetch = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int o = pc * 1024 + lane * 16;
      const int j = o / PPW, wbyte = o - j * PPW;
      const bool seg1 = j >= len0 + 2;
      const int col = seg1 ? j - (len0 + 2) : c00 - 1 + j;
      const int img = img0 + (seg1 ? 1 : 0);
      const bool ok = j < NPX && wbyte < KPW * 2 && (unsigned)col < (unsigned)p.W && img < n_live && (!seg1 || len0 < 16);
      f_boff[pc] = ok ? ((img * p.H * p.W + col) * CIN + wave * KPW) * 2 + wbyte : OOB;
    }
  };
  int c_frag_n = 0, ob_n = OOB, ob_c = OOB;      // next step's fragment lane offset; output lane offsets of the next / this step's strip
  auto setup_comp = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
    const bool in1 = fr >= len0;
    c_frag_n = (fr + (in1 ? 1 : 0)) * PPW + fq * 16;
    const int img = img0 + (in1 ? 1 : 0), col = in1 ? fr - len0 : c00 + fr;
    ob_n = img < n_live ? ((img * p.H * p.W + col) * p.opitch + cout0 + fq * 8) * 2 : OOB;
  };
  auto seg_init = [&](RowsIt& it, int strip) __attribute__((always_inline)) {
    it.strip = strip;
    it.r_lo = max(wa - strip * p.H, 0);
    it.r_hi = min(wb - strip * p.H, p.H);
    it.q = it.r_lo - 1;
  };
  const int s_first = wa / p.H, s_last = (wb - 1) / p.H;
  const int n_steps = (wb - wa) + 2 * (s_last - s_first + 1);

  unsigned char* const ring_w = smem + wave * (NSLOT * ROWB);
  RowsIt it_f, it_n;
  seg_init(it_f, s_first);
  seg_init(it_n, s_first);
  setup_fetch(s_first);
  setup_comp(s_first);
  int f_left = n_steps, f_slot = 0, f_roff = it_f.q * in_row;
  auto fetch_next = [&]() __attribute__((always_inline)) {
    const bool row_ok = f_left > 0 && (unsigned)it_f.q < (unsigned)p.H;
    const int roff = row_ok ? f_roff : OOB;      // (a row that is zero padding or past the end: every valid lane out of range; nobody multiplies it)
#pragma unroll
    for (int pc = 0; pc < NP; ++pc)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(ring_w + f_slot + pc * 1024), 16, f_boff[pc] + roff, 0, 0, 0);
    f_slot = f_slot == (NSLOT - 1) * ROWB ? 0 : f_slot + ROWB;
    if (f_left > 0) {
      --f_left;
      if (it_f.q == it_f.r_hi) {
        if (it_f.strip < s_last) { seg_init(it_f, it_f.strip + 1); setup_fetch(it_f.strip); f_roff = it_f.q * in_row; }
      } else {
        ++it_f.q;
        f_roff += in_row;
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) fetch_next();

  f32x4 acc[3][NCT];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[a][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 bf[NF];
  int c_q = it_n.q, c_rlo = it_n.r_lo, c_rhi = it_n.r_hi;
  int n_slot = 0;
  int e_in = 0, e_out = 0;                       // rows taken over from wave h - 1 / handed on to wave h + 1 so far
  unsigned char* const hand_in = smem + HAND + (wave - 1) * (HB * HSLOT) + lane * 16;      // (wave 0 never uses it)
  unsigned char* const hand_out = smem + HAND + wave * (HB * HSLOT) + lane * 16;
  // a bounded wait: a protocol error must not hang the GPU (the result is then wrong and the tests say so)
  auto wait_ge = [&](volatile int* f, int v) __attribute__((always_inline)) {
    int spins = 0;
    while (*f < v && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++spins; }
  };

  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // the first row has landed
  {
    const unsigned char* const row = ring_w + c_frag_n;
    static_for<0, NF>([&](auto ff) { constexpr int f = decltype(ff)::value; bf[f] = *reinterpret_cast<const u32x4*>(row + (f % 3) * PPW + (f / 3) * 64); });
  }

  auto step = [&](auto ph_c, int s) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int A_OLD = (PH + 2) % 3, A_MID = PH, A_NEW = (PH + 1) % 3;
    if (s > 0) { c_q = it_n.q; c_rlo = it_n.r_lo; c_rhi = it_n.r_hi; }
    ob_c = ob_n;
    if (s + 1 < n_steps) {
      if (it_n.q == it_n.r_hi) {
        if (it_n.strip < s_last) { seg_init(it_n, it_n.strip + 1); setup_comp(it_n.strip); }
      } else {
        ++it_n.q;
      }
    }
    n_slot = n_slot == (NSLOT - 1) * ROWB ? 0 : n_slot + ROWB;
    const bool row_ok = (unsigned)c_q < (unsigned)p.H;
    const bool need_in = wave > 0 && c_q + 1 >= c_rlo && c_q + 1 < c_rhi;      // the row that starts in this step comes from wave h - 1
    const bool emit = c_q - 1 >= c_rlo;                                        // the row that is complete after this step goes on
    if (need_in) wait_ge(flagP + wave - 1, e_in + 1);
    asm volatile("" ::: "memory");
    const unsigned char* const hin = need_in ? hand_in + (e_in & (HB - 1)) * HSLOT : smem + ZERO + lane * 16;
    fetch_next();                                  // row of step s + D -> the ring slot of step s - 1
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    auto mma = [&](auto ff, auto ky_c, f32x4 (&a)[NCT]) __attribute__((always_inline)) {
      constexpr int f = decltype(ff)::value, ky = decltype(ky_c)::value;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) Mma<T>::run(wfr[(ct * NCB + f / 3) * 9 + ky * 3 + f % 3], bf[f], a[ct]);
    };
    const unsigned char* const row_n = ring_w + n_slot + c_frag_n;
    if (row_ok) {
      // one straight-line block (a branch between two groups makes hipcc wait for every LDS read in flight at the block edge)
      f32x4 hv[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) hv[ct] = *reinterpret_cast<const f32x4*>(hin + ct * 1024);
      static_for<0, NF>([&](auto ff) { mma(ff, K2{}, acc[A_OLD]); });
      static_for<0, NF>([&](auto ff) { mma(ff, K1{}, acc[A_MID]); });
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[A_NEW][ct] = hv[ct];
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // the row of step s + 1 has landed (rows s + 2, s + 3 and this wave's stores may be in flight)
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        mma(ff, K0{}, acc[A_NEW]);
        bf[f] = *reinterpret_cast<const u32x4*>(row_n + (f % 3) * PPW + (f / 3) * 64);
      });
    } else {
      // a zero-padding row: nothing to multiply; the starting row still takes over its partial sums
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[A_NEW][ct] = *reinterpret_cast<const f32x4*>(hin + ct * 1024);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      static_for<0, NF>([&](auto ff) { constexpr int f = decltype(ff)::value; bf[f] = *reinterpret_cast<const u32x4*>(row_n + (f % 3) * PPW + (f / 3) * 64); });
    }
    if (need_in) {
      // (the tile has been read: LDS executes a wave's instructions in order, the counter store follows the reads)
      asm volatile("" ::: "memory");
      ++e_in;
      flagC[wave] = e_in;
    }
    if (emit) {
      if (wave < 7) {
        wait_ge(flagC + wave + 1, e_out - (HB - 1));                // the slot's previous tile has been taken over
        asm volatile("" ::: "memory");
        unsigned char* const ho = hand_out + (e_out & (HB - 1)) * HSLOT;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(ho + ct * 1024) = acc[A_OLD][ct];
        asm volatile("" ::: "memory");
        ++e_out;
        flagP[wave] = e_out;
      } else {
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float x0 = acc[A_OLD][k >> 1][(2 * k) & 3] + bias8[2 * k], x1 = acc[A_OLD][k >> 1][(2 * k + 1) & 3] + bias8[2 * k + 1];
          if (p.relu) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
          pk[k] = Elem<T>::pack2(x0, x1);
        }
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, ob_c + (c_q - 1) * out_row, 0, 0);
      }
    }
  };
  for (int s = 0; s < n_steps; s += 3) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < n_steps) step(std::integral_constant<int, 1>{}, s + 1);
    if (s + 2 < n_steps) step(std::integral_constant<int, 2>{}, s + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy fetches behind the last step still target this workgroup's LDS
}

#endif  // DP_EXPERIMENTS

// =====================================================================================================
// Third form (round 4): 32 pixels per step, waves = 4 K quarters x 2 cout halves. The two forms above spend ~1000 cycles per step on
// work that does not grow with the matrix work (LDS-DMA issue, reduction, bookkeeping, barrier skew) against 36 MFMAs per wave. Here
//   * wave (kq, ch) holds the weights of input channels [128 kq, 128 kq + 128) x 9 taps x the 16 couts of half ch: the same 36 A
//     fragments = 144 VGPRs; a step is one input row of a 32-PIXEL strip = two pixel tiles = 72 MFMAs per wave;
//   * the two waves of a K quarter read the SAME staged row slice (35 pixels x 256 B, 272-byte pixel pitch) from a ring they fill
//     together (five 1 KiB LDS-DMA pieces each per step instead of 2 x 3 for the same pixels); four partial sums per output, not eight;
//   * the quarters 0 / 1 live in waves 0 .. 3, 2 / 3 in waves 4 .. 7, and the two waves of a SIMD (w, w + 4) belong to different
//     quarters: with the barrier in different places for the two halves (waves 0 .. 3: A B |, waves 4 .. 7: A | B) a SIMD's waves
//     are in opposite phases, and every ring is produced and consumed by waves of ONE half, i.e. in lockstep;
//   * fragments are read three at a time (one 32-channel block of one pixel tile, double buffered) between the MFMA groups: the
//     accumulators (3 roles x 2 pixel tiles) and two fragment buffers are 48 registers beside the weights.
// Strips are cut from the concatenated columns of G = 32 / gcd(W, 32) images (28-wide ROI maps: 8 ROIs = 224 columns = 7 strips); a
// strip is at most two segments with one shared zero pixel between them (34 or 35 staged pixels).
// Per output pixel: partial sum of quarter kq = kernel row, 32-channel block, kernel column (one fp32 MFMA chain); the four partials
// are added in quarter order. Fixed, independent of strip, lane, workgroup and batch - and different from the two forms above and from
// the LDS-ring kernels: which form a layer takes is decided by its geometry alone (dp_conv_rows_launch).
// =====================================================================================================
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void rows2_wait_vm() {
  static_assert(N >= 4 && N <= 7, "vmcnt immediate");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
}

template <typename T, int CIN>
__global__ __launch_bounds__(512, 2) void conv3x3_rows2_kernel(const RowsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(CIN == 512 || CIN == 256, "36 weight fragments per wave");
  constexpr int KPQ = CIN / 4;                  // input channels of a K quarter
  constexpr int NCB = KPQ / 32;                 // 32-channel blocks per quarter
  constexpr int NW = 4 / NCB;                   // cout tiles per wave: 1 (512 channels: 32 couts per workgroup) or 2 (256: 64 couts)
  constexpr int PPW = KPQ * 2 + 16, NPX = 35;   // pixel pitch 272 / 144 bytes: 17 / 9 sixteen-byte slots, odd - conflict-free ds_read_b128
  constexpr int ROWB = NPX * PPW;               // 9520 / 5040 bytes per staged row slice
  constexpr int WOFF = ((ROWB / 2 + 511) / 512) * 512;      // wave ch = 0 stages bytes [0, WOFF), ch = 1 [WOFF, ROWB): 5120 / 2560
  constexpr int NPW = (WOFF + 1023) / 1024;     // pieces per wave and row (5 / 3; the last one of a wave may be narrower than 64 lanes)
  constexpr int NSLOT = 3;
  constexpr int RING = 4 * NSLOT * ROWB;
  constexpr int STGB = 8 * NW * 2 * 1024;       // staging bytes per parity: [kq][ch][cout tile][pixel tile][lane] x 16 B
  constexpr int OOB = (int)0x80000000;
  static_assert(ROWB % 16 == 0 && RING + 3 * STGB <= 160 * 1024, "LDS budget");
  // the counted vmcnt waits below assume that EVERY wave issues exactly NPW pieces per row: the last piece of either wave must have at
  // least one live lane (a wave-wide empty exec would be branched around and not counted)
  static_assert((NPW - 1) * 1024 < WOFF && WOFF + (NPW - 1) * 1024 < ROWB && WOFF % 16 == 0, "every wave issues NPW pieces");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int ch = wave & 1, kq = 2 * (wave >> 2) + ((wave >> 1) & 1);
  const int b = blockIdx.x;
  const int slice = (b >> 3) % p.n_slices;
  const int pg = (b & 7) + 8 * (b / (8 * p.n_slices));

  const int n_live = p.n_dev ? min(max(*p.n_dev - p.n_base, 0), p.N) : p.N;
  // whole groups of G images, then the strips that the columns of the remaining images reach (a strip is SW = G * W / SPG pixels)
  const int n_strips = (n_live / p.G) * p.SPG + ((n_live % p.G) * p.W * p.SPG + p.G * p.W - 1) / (p.G * p.W);
  const long long TR = (long long)n_strips * p.H;
  const int wa = (int)(TR * pg / p.n_pg), wb = (int)(TR * (pg + 1) / p.n_pg);
  if (wb <= wa) return;

  // ---- this wave's weights: cout tile ch of the slice x 4 channel blocks of quarter kq x 9 taps
  u32x4 wfr[NW * NCB * 9];
  const int cout0 = slice * (32 * NW);
  {
    const int n_planes = p.kpad * 2 / 64;
    const unsigned char* __restrict__ wp = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
    for (int ct = 0; ct < NW; ++ct)
#pragma unroll
      for (int cbl = 0; cbl < NCB; ++cbl)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          wfr[(ct * NCB + cbl) * 9 + t] =
              *reinterpret_cast<const u32x4*>(wp + dp_wtile_off(cout0 + (ch * NW + ct) * 16 + fr, (kq * NCB + cbl) * 9 + t, fq, n_planes));
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  const int in_row = p.W * CIN * 2, out_row = p.W * p.opitch * 2;

  auto decode = [&](int strip, int& img0, int& c00, int& len0) __attribute__((always_inline)) {
    const int grp = strip / p.SPG, k = strip - grp * p.SPG;
    const int x0 = k * 32, ig = x0 / p.W;
    c00 = x0 - ig * p.W;
    img0 = grp * p.G + ig;
    len0 = min(32, p.W - c00);
  };

  // ---- fetch side: this lane's part of the wave's five pieces of a row slice
  static_assert(NPW <= 5, "f_boff");
  int f_boff[5];
  const bool f_last = ch * WOFF + (NPW - 1) * 1024 + lane * 16 < (ch ? ROWB : WOFF);       // a wave's last piece may be narrower
  auto setup_fetch = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
#pragma unroll
    for (int pc = 0; pc < NPW; ++pc) {
      const int o = ch * WOFF + pc * 1024 + lane * 16;
      const int j = o / PPW, wbyte = o - j * PPW;
      const bool seg1 = j >= len0 + 2;
      const int col = seg1 ? j - (len0 + 2) : c00 - 1 + j;
      const int img = img0 + (seg1 ? 1 : 0);
      const bool ok = j < NPX && wbyte < KPQ * 2 && (unsigned)col < (unsigned)p.W && img < n_live && (!seg1 || len0 < 32);
      f_boff[pc] = ok ? ((img * p.H * p.W + col) * CIN + kq * KPQ) * 2 + wbyte : OOB;
    }
  };
  unsigned char* const ring_q = smem + kq * (NSLOT * ROWB);
  unsigned char* const ring_wv = ring_q + ch * WOFF;

  // ---- compute side: lane (fr, fq) = output pixels fr and 16 + fr of the strip
  // ---- reduce side: wave w adds up the four partial sums of pixel tile w >> 2 in cout half (w & 1).
  // One cout tile per wave (512 channels): lane l takes TWO consecutive output channels of pixel (l & 31) >> 1, channel quad
  // 2 * ((w >> 1) & 1) + (l >> 5). Two cout tiles (256 channels): lane l takes FOUR consecutive channels (one lane's values of one
  // cout tile) of pixel l & 15, channel quad 2 * ((w >> 1) & 1) + ((l >> 4) & 1), cout tile l >> 5.
  const int r_pt = wave >> 2, r_ch = wave & 1;
  const int r_px = NW == 1 ? (lane & 31) >> 1 : lane & 15;
  const int r_fq = 2 * ((wave >> 1) & 1) + (NW == 1 ? lane >> 5 : (lane >> 4) & 1);
  const int r_half = lane & 1, r_ct = NW == 1 ? 0 : lane >> 5;
  const int r_lds = ((r_ch * NW + r_ct) * 2 + r_pt) * 1024 + (r_fq * 16 + r_px) * 16 + (NW == 1 ? r_half * 8 : 0);
  const int r_co = NW == 1 ? r_fq * 8 + r_ch * 4 + r_half * 2 : r_ch * 32 + r_fq * 8 + r_ct * 4;       // first output channel of the lane, in the slice
  constexpr int RN = NW == 1 ? 2 : 4;           // output channels per lane
  float r_bias[RN];
#pragma unroll
  for (int i = 0; i < RN; ++i) r_bias[i] = p.bias[cout0 + r_co + i];
  int c_frag[2] = {0, 0};
  int r_obase = OOB;
  auto setup_comp = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int px = pt * 16 + fr;
      c_frag[pt] = (px + (px >= len0 ? 1 : 0)) * PPW + fq * 16;
    }
    const int px = r_pt * 16 + r_px;
    const bool in1 = px >= len0;
    const int img = img0 + (in1 ? 1 : 0), col = in1 ? px - len0 : c00 + px;
    r_obase = img < n_live ? ((img * p.H * p.W + col) * p.opitch + cout0 + r_co) * 2 : OOB;
  };

  auto seg_init = [&](RowsIt& it, int strip) __attribute__((always_inline)) {
    it.strip = strip;
    it.r_lo = max(wa - strip * p.H, 0);
    it.r_hi = min(wb - strip * p.H, p.H);
    it.q = it.r_lo - 1;
  };
  const int s_first = wa / p.H, s_last = (wb - 1) / p.H;
  const int n_steps = (wb - wa) + 2 * (s_last - s_first + 1);

  RowsIt it_f, it_c;
  seg_init(it_f, s_first);
  seg_init(it_c, s_first);
  setup_fetch(s_first);
  setup_comp(s_first);
  int f_left = n_steps, f_slot = 0;
  auto fetch_next = [&]() __attribute__((always_inline)) {
    const bool row_ok = f_left > 0 && (unsigned)it_f.q < (unsigned)p.H;     // rows -1 and H and the rows behind the end: zeros
    const int roff = it_f.q * in_row;
    unsigned char* const dst = ring_wv + f_slot;
    const bool issue = !(DP_ROWS_EXP & 1) || f_left > n_steps - 2;
#pragma unroll
    for (int pc = 0; pc < NPW - 1; ++pc)
      if (issue) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(dst + pc * 1024), 16, row_ok ? f_boff[pc] + roff : OOB, 0, 0, 0);
    if (f_last && issue)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(dst + (NPW - 1) * 1024), 16, row_ok ? f_boff[NPW - 1] + roff : OOB, 0, 0, 0);
    f_slot = f_slot == (NSLOT - 1) * ROWB ? 0 : f_slot + ROWB;
    if (f_left > 0) {
      --f_left;
      if (it_f.q == it_f.r_hi) {
        if (it_f.strip < s_last) { seg_init(it_f, it_f.strip + 1); setup_fetch(it_f.strip); }
      } else {
        ++it_f.q;
      }
    }
  };

  f32x4 acc[3][NW][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int ct = 0; ct < NW; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[a][ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned char* const stg = smem + RING;
  unsigned char* const stg_w = stg + ((kq * 2 + ch) * NW * 2) * 1024 + lane * 16;

  // the row step s finished: four partial tiles in staging parity s % 3 -> bias, activation, one 4-byte store per lane. ALWAYS one
  // store per step (out of range when there is nothing to store: the hardware drops it) - the counted vmcnt waits below rely on it.
  // In two halves: the four LDS reads are issued at the top of phase A, the sums are taken after the step's LDS-DMA issue.
  using RV = typename std::conditional<NW == 1, f32x2, f32x4>::type;
  RV rv[4];
  auto reduce_issue = [&](int par) __attribute__((always_inline)) {
    const unsigned char* const sr = stg + par * STGB + r_lds;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      if constexpr (DP_ROWS_EXP & 2) { for (int i = 0; i < RN; ++i) rv[q4][i] = 1.f; }
      else rv[q4] = *reinterpret_cast<const RV*>(sr + q4 * (2 * NW * 2 * 1024));
    }
  };
  auto reduce_finish = [&](int t, int obase, bool emit) __attribute__((always_inline)) {
    float x[RN];
#pragma unroll
    for (int i = 0; i < RN; ++i) {
      x[i] = rv[0][i];
#pragma unroll
      for (int q4 = 1; q4 < 4; ++q4) x[i] += rv[q4][i];     // quarter order: the summation order of a pixel is fixed
      x[i] += r_bias[i];
      if (p.relu) x[i] = fmaxf(x[i], 0.f);
    }
    const int off = (emit && !(DP_ROWS_EXP & 2)) ? obase + t * out_row : OOB;
    if constexpr (NW == 1) __builtin_amdgcn_raw_buffer_store_b32(Elem<T>::pack2(x[0], x[1]), rs_out, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{Elem<T>::pack2(x[0], x[1]), Elem<T>::pack2(x[2], x[3])}, rs_out, off, 0, 0);
  };

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl = (DP_ROWS_EXP & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
#define DP_STAMP(k) if constexpr (DP_ROWS_EXP & 16) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tl; tl = t_; }
  const bool sched_y = wave >= 4 || p.lockstep;      // barrier between phase A and phase B (else behind phase B)
  // history of the two previous steps (the row of step s - 2 is complete in the staging buffer when phase A of step s starts)
  bool p1_emit = false, p2_emit = false;
  int p1_t = 0, p2_t = 0, p1_ob = OOB, p2_ob = OOB;
  int c_slot = 0, c_par = 0;                          // ring slot / staging parity of step s
  // bookkeeping of step s around the reduction of the row step s - 2 finished (staging parity (s - 2) % 3: complete since the last
  // barrier, overwritten after the next one - so every wave reduces BEFORE its next barrier)
  auto phase_a = [&](int s) __attribute__((always_inline)) {
    reduce_issue(c_par == 0 ? 2 : c_par - 1);         // (the previous step's parity + 2) % 3, before c_par moves on
    if (s > 0) {
      p2_emit = p1_emit; p2_t = p1_t; p2_ob = p1_ob;
      p1_t = it_c.q - 1;
      p1_emit = p1_t >= it_c.r_lo;
      p1_ob = r_obase;
      if (it_c.q == it_c.r_hi) {
        if (it_c.strip < s_last) { seg_init(it_c, it_c.strip + 1); setup_comp(it_c.strip); }
      } else {
        ++it_c.q;
      }
      c_slot = c_slot == (NSLOT - 1) * ROWB ? 0 : c_slot + ROWB;
      c_par = c_par == 2 ? 0 : c_par + 1;
    }
    DP_STAMP(0)
  };
  constexpr int NBUF = (DP_ROWS2_OPT & 1) ? 3 : 2;
  u32x4 bf[NBUF][3];
  auto phase_b = [&](auto ph_c, int s) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int A_OLD = (PH + 2) % 3, A_MID = PH, A_NEW = (PH + 1) % 3;
    const bool row_ok = (unsigned)it_c.q < (unsigned)p.H;
    const unsigned char* const row0 = ring_q + c_slot + c_frag[0];
    const unsigned char* const row1 = ring_q + c_slot + c_frag[1];
#pragma unroll
    for (int ct = 0; ct < NW; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[A_NEW][ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NB = 2 * NCB;                  // batches per step: (pixel tile, channel block) = 3 fragments = 9 NW MFMAs
    if (row_ok) {
      // one straight-line block, the next batch's reads issued first
      auto ld = [&](auto bb) __attribute__((always_inline)) {
        constexpr int bi = decltype(bb)::value, pt = bi / NCB, cb = bi % NCB;
        const unsigned char* const r = pt ? row1 : row0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if constexpr (DP_ROWS_EXP & 8) bf[bi % NBUF][kx] = u32x4{(unsigned)lane, (unsigned)bi, 1u, 2u};
          else bf[bi % NBUF][kx] = *reinterpret_cast<const u32x4*>(r + kx * PPW + cb * 64);
        }
      };
      ld(std::integral_constant<int, 0>{});
      if constexpr (NBUF == 3) ld(std::integral_constant<int, 1>{});
      if constexpr (!(DP_ROWS2_OPT & 2)) __builtin_amdgcn_s_setprio(1);
      static_for<0, NB>([&](auto bb) {
        constexpr int bi = decltype(bb)::value, pt = bi / NCB, cb = bi % NCB;
        // (fenced: left alone hipcc moves each read down to just before its first use and waits for it there - ten exposed LDS
        // latencies per step)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (bi + NBUF - 1 < NB) ld(std::integral_constant<int, bi + NBUF - 1>{});
        if constexpr ((DP_ROWS2_OPT & 4) != 0 && bi == NB / 2) { if (sched_y) fetch_next(); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if constexpr (DP_ROWS_EXP & 4) { acc[A_OLD][0][pt][0] += __builtin_bit_cast(float, bf[bi % NBUF][kx][0]); }
          else {
#pragma unroll
            for (int ct = 0; ct < NW; ++ct) {
              Mma<T>::run(wfr[(ct * NCB + cb) * 9 + 6 + kx], bf[bi % NBUF][kx], acc[A_OLD][ct][pt]);
              Mma<T>::run(wfr[(ct * NCB + cb) * 9 + 3 + kx], bf[bi % NBUF][kx], acc[A_MID][ct][pt]);
              Mma<T>::run(wfr[(ct * NCB + cb) * 9 + 0 + kx], bf[bi % NBUF][kx], acc[A_NEW][ct][pt]);
            }
          }
        }
      });
      if constexpr (!(DP_ROWS2_OPT & 2)) __builtin_amdgcn_s_setprio(0);
    } else if constexpr ((DP_ROWS2_OPT & 4) != 0) {
      if (sched_y) fetch_next();
    }
    // the finished row's partial sums (whatever they are when nothing is emitted: the reduction then stores out of range)
#pragma unroll
    for (int ct = 0; ct < NW; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) *reinterpret_cast<f32x4*>(stg_w + c_par * STGB + (ct * 2 + pt) * 1024) = acc[A_OLD][ct][pt];
  };

  // prologue: rows 0 and 1 staged and visible to the whole workgroup
  fetch_next();
  fetch_next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // Per wave and step, in issue order: 5 LDS-DMA pieces (4 + the masked one), 1 store. The pieces of row s + 1 must have landed
  // before the barrier that ends the interval in which this wave computes row s: younger than them are the store of that interval,
  // the pieces of row s + 2 and the store of this one = 7 operations (the masked piece counts: exec = 0 lanes still issue).
  // Waves 0 .. 3: A (reduction reads, bookkeeping, LDS-DMA issue of row s + 2, sums + store), B | ; waves 4 .. 7: A (without the
  // LDS-DMA issue) | B, LDS-DMA issue - after every barrier one wave of a SIMD starts with memory work, the other with its matrix
  // block. Row s + 2 goes to the slot of row s - 1, which its ring's readers (waves of the same half) finished before their last barrier.
  auto unit = [&](auto ph_c, int s) __attribute__((always_inline)) {
    DP_STAMP(7)
    phase_a(s);
    if (sched_y) {
      reduce_finish(p2_t, p2_ob, s >= 2 && p2_emit);
      DP_STAMP(1)
      rows2_wait_vm<NPW + 2>();
      DP_STAMP(2)
      __builtin_amdgcn_s_barrier();
      DP_STAMP(3)
    } else if constexpr ((DP_ROWS2_OPT & 8) != 0) {
      reduce_finish(p2_t, p2_ob, s >= 2 && p2_emit);
      DP_STAMP(1)
      fetch_next();
      DP_STAMP(4)
    } else {
      fetch_next();
      DP_STAMP(4)
      reduce_finish(p2_t, p2_ob, s >= 2 && p2_emit);
      DP_STAMP(1)
    }
    phase_b(ph_c, s);
    if constexpr (DP_ROWS_EXP & 16) asm volatile("s_nop 0" :: "v"(acc[0][0][0][0]), "v"(acc[1][0][0][0]), "v"(acc[2][0][0][0]), "v"(acc[0][NW - 1][1][0]), "v"(acc[1][NW - 1][1][0]), "v"(acc[2][NW - 1][1][0]));
    DP_STAMP(5)
    if (!sched_y) {
      if constexpr ((DP_ROWS2_OPT & 8) != 0) rows2_wait_vm<NPW + 1>();     // store, then the pieces: one operation fewer behind row s + 1
      else rows2_wait_vm<NPW + 2>();
      DP_STAMP(2)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      DP_STAMP(6)
      __builtin_amdgcn_s_barrier();
      DP_STAMP(3)
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      DP_STAMP(6)
      if constexpr (!(DP_ROWS2_OPT & 4)) fetch_next();
      DP_STAMP(4)
    }
  };
  for (int s = 0; s < n_steps; s += 3) {
    unit(std::integral_constant<int, 0>{}, s);
    if (s + 1 < n_steps) unit(std::integral_constant<int, 1>{}, s + 1);
    if (s + 2 < n_steps) unit(std::integral_constant<int, 2>{}, s + 2);
  }
  __builtin_amdgcn_s_barrier();                  // waves 4 .. 7 have written the last step's partial sums
  {
    const int tl_ = it_c.q - 1;
    const int par1 = c_par, par2 = c_par == 0 ? 2 : c_par - 1;
    if (n_steps >= 2) { reduce_issue(par2); reduce_finish(p1_t, p1_ob, p1_emit); }
    reduce_issue(par1);
    reduce_finish(tl_, r_obase, tl_ >= it_c.r_lo);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy fetches behind the last step still target this workgroup's LDS
  if constexpr (DP_ROWS_EXP & 16) {
    if (lane == 0 && p.dbg) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p.dbg[(blockIdx.x * 8 + wave) * 16 + k] = ph[k];
      p.dbg[(blockIdx.x * 8 + wave) * 16 + 8] = n_steps;
    }
  }
#undef DP_STAMP
}

template <typename T, int CIN>
int launch_rows2(const RowsArgs& a, hipStream_t stream) {
  constexpr int lds = 4 * 3 * 35 * (CIN / 2 + 16) + 3 * 8 * (CIN == 512 ? 1 : 2) * 2 * 1024;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rows2_kernel<T, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
#if DP_ROWS_EXP & 16
  RowsArgs b = a;
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_pg * a.n_slices;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 8 * 16 * 4096);
  b.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 8 * 16 * nblk, stream);
  hipLaunchKernelGGL((conv3x3_rows2_kernel<T, CIN>), dim3(nblk), dim3(512), lds, stream, b);
  {
    static int shown = 0;
    if (shown++ % 40 == 4) {   // a warm launch (and again for every further mode of tools/rows_micro.py)
      (void)hipStreamSynchronize(stream);
      unsigned long long* hb = (unsigned long long*)malloc(sizeof(unsigned long long) * 128 * nblk);
      (void)hipMemcpy(hb, dbg, sizeof(unsigned long long) * 128 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) {
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n = 0;
        for (int bb = 0; bb < nblk; ++bb) { for (int k = 0; k < 8; ++k) sum[k] += (double)hb[(bb * 8 + w) * 16 + k]; n += (double)hb[(bb * 8 + w) * 16 + 8]; }
        fprintf(stderr, "rows2 wave %d (lockstep %d): cycles per step: bookkeeping%s %.0f  reduce %.0f  vmcnt wait %.0f  barrier %.0f  fetch issue after the barrier %.0f  B: reads + mfma + staging write %.0f  lgkm wait %.0f  loop edge %.0f  (steps/wg %.1f)\n",
                w, a.lockstep, (w < 4 && !a.lockstep) ? " + fetch issue" : "", sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n, sum[7] / n, n / nblk);
      }
      free(hb);
    }
  }
#else
  hipLaunchKernelGGL((conv3x3_rows2_kernel<T, CIN>), dim3(a.n_pg * a.n_slices), dim3(512), lds, stream, a);
#endif
  return dp_check_launch("conv3x3_rows2_kernel");
}

#ifdef DP_EXPERIMENTS
template <typename T>
int launch_chain(const RowsArgs& a, hipStream_t stream) {
  constexpr int lds = 8 * 4 * 3072 + 7 * 4 * 2048 + 2 * 2048 + 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_chain_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_chain_kernel<T>), dim3(a.n_pg * a.n_slices), dim3(512), lds, stream, a);
  return dp_check_launch("conv3x3_chain_kernel");
}
#endif

template <typename T, int CIN, int NCT>
int launch_rows_r(const RowsArgs& a, hipStream_t stream) {
  constexpr int KPW = CIN / 8, PPW = KPW * 2 + 32, NP = (19 * PPW + 1023) / 1024;
  constexpr int lds = 8 * 4 * NP * 1024 + 3 * 8 * NCT * 1024;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rows_kernel<T, CIN, NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
#if DP_ROWS_EXP & 16
  RowsArgs b = a;
  static unsigned long long* dbg = nullptr;
  const int nblk = a.n_pg * a.n_slices;
  if (!dbg) (void)hipMalloc(&dbg, sizeof(unsigned long long) * 8 * 16 * 4096);
  b.dbg = dbg;
  (void)hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 8 * 16 * nblk, stream);
  hipLaunchKernelGGL((conv3x3_rows_kernel<T, CIN, NCT>), dim3(nblk), dim3(512), lds, stream, b);
  {
    static int shown = 0;
    if (shown++ == 4) {   // a warm launch
      (void)hipStreamSynchronize(stream);
      unsigned long long* hb = (unsigned long long*)malloc(sizeof(unsigned long long) * 128 * nblk);
      (void)hipMemcpy(hb, dbg, sizeof(unsigned long long) * 128 * nblk, hipMemcpyDeviceToHost);
      for (int w = 0; w < 8; ++w) {
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n = 0;
        for (int bb = 0; bb < nblk; ++bb) { for (int k = 0; k < 8; ++k) sum[k] += (double)hb[(bb * 8 + w) * 16 + k]; n += (double)hb[(bb * 8 + w) * 16 + 8]; }
        fprintf(stderr, "rows wave %d: cycles per step: A: fetch issue %.0f  reduce %.0f  vmcnt wait + read issue %.0f  barrier (waves 4-7) %.0f | B: mfma + staging write %.0f  lgkm wait %.0f  barrier (waves 0-3) %.0f | bookkeeping %.0f  (steps/wg %.1f)\n",
                w, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n, sum[7] / n, n / nblk);
      }
      free(hb);
    }
  }
#else
  hipLaunchKernelGGL((conv3x3_rows_kernel<T, CIN, NCT>), dim3(a.n_pg * a.n_slices), dim3(512), lds, stream, a);
#endif
  return dp_check_launch("conv3x3_rows_kernel");
}

template <typename T>
int launch_rows(const RowsArgs& a, int cin, hipStream_t stream) {
  return cin == 512 ? launch_rows_r<T, 512, 2>(a, stream) : launch_rows_r<T, 256, 4>(a, stream);
}

int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }

// every strip of a group of G = 16 / gcd(W, 16) images is at most two segments from two consecutive images
bool rows_width_ok(int W) {
  if (W < 8) return false;
  const int G = 16 / gcd_i(W, 16), spg = G * W / 16;
  for (int k = 0; k < spg; ++k) {
    const int c00 = (16 * k) % W, len0 = W - c00 < 16 ? W - c00 : 16;
    if (16 - len0 > W) return false;
  }
  return true;
}

// every 32-pixel strip of a group of G = 32 / gcd(W, 32) images is at most two segments from two consecutive images
bool rows2_width_ok(int W) {
  if (W < 16) return false;
  const int G = 32 / gcd_i(W, 32), spg = G * W / 32;
  for (int k = 0; k < spg; ++k) {
    const int c00 = (32 * k) % W, len0 = W - c00 < 32 ? W - c00 : 32;
    if (32 - len0 > W) return false;
  }
  return true;
}

bool rows_common_ok(const dp_conv_params* p, int g) {
  return (p->dtype == DP_BF16 || p->dtype == DP_F16) && p->ntaps == 9 && p->Kpad == 9 * p->Cin && p->stride == 1 &&
         (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == -1 && p->wi_off == -1 && p->H == p->Ho && p->W == p->Wo &&
         !p->residual && !p->out_f32 && !p->head_out && !p->in2 && !p->post_res && p->post_mode == 0 && p->split_k <= 1 && p->out &&
         p->osW >= p->Cout && p->osW % 8 == 0 && p->osH == (long long)p->W * p->osW && p->osN == (long long)p->H * p->W * p->osW &&
         p->Cout <= p->Cout_w && p->Cout_w % 64 == 0 &&
         // 16-byte LDS-DMA loads and buffer stores relative to these bases: an unaligned caller falls through to the tiled kernels
         (((uintptr_t)p->in | (uintptr_t)p->out | (uintptr_t)p->weight) & 15) == 0 &&
         (long long)(2 * g) * p->H * p->W * p->Cin * 2 < (1ll << 30) && (long long)(2 * g) * p->H * p->W * p->osW * 2 < (1ll << 30);
}

// the 32-pixel form: 512 input channels and a width whose strip groups are small (decided by the geometry alone: its summation order
// differs from the 16-pixel form's). Policy key conv_rows2 = 0: never.
bool rows2_ok(const dp_conv_params* p) {
  const DpPolicy& pol = dp_policy();
  if (pol.conv_rows2 == 0 || pol.conv_rows == 0) return false;
  if (p->W <= 0) return false;
  const int g = 32 / gcd_i(p->W, 32);
  // 512 channels: widths with strip groups of at most 8 images (28-wide ROI maps; res5's 42-wide maps - groups of 16 - stay on the 16-pixel
  // form). 256 channels (64 couts per workgroup): the layers with another cout count (the DensePose head's 256 -> 512 first layer: 93 us
  // against 105 on the LDS-ring kernel). On 256 -> 256 layers it measures the same as the weight-stationary kernel of dp_conv_ws.hip
  // (465 / 121 / 40 us against 461 / 121 / 39 at the three FPN levels), and the DeepLab head's device-sized 256 -> 256 layers gain nothing
  // end to end (profiles/r4_rows_kernel_experiments.txt). Policy key conv_rows2_256: 0 never, 1 every 256-channel layer.
  const int m256 = (int)pol.conv_rows2_256;
  const bool c256 = p->Cin == 256 && p->Cout % 64 == 0 && m256 != 0 && (m256 == 1 || p->Cout != 256);
  const int maxg = (int)pol.conv_rows2_maxg;           // experiments: largest strip group the 512-channel layers take (default 8)
  const bool shape = (p->Cin == 512 && p->Cout % 32 == 0 && g <= maxg) || c256;
  return shape && rows2_width_ok(p->W) && rows_common_ok(p, g);
}

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): is this launch a 3x3 / pad 1 / stride 1 layer the kernels here are written for? No size
// thresholds: their summation orders differ from the LDS-ring kernels' (and from each other), so a layer either always runs on one form
// or never (batch invariance).
bool dp_conv_rows_ok(const dp_conv_params* p) {
  // Default (mode 1): the 512-channel layers. Widths with small 32-pixel strip groups (the DensePose head's 28-wide ROI maps, with or
  // without a device-side count) take the 32-pixel form; the others (res5's conv2 at 42 columns) the 16-pixel form when the launch is
  // not sized on the device - for n_dev launches the 16-pixel form measured at par with the ring kernel
  // (profiles/r4_rows_kernel_experiments.txt), and which kernel a call site takes must not depend on the batch. Policy key conv_rows:
  // 0 never, 2 the 16-pixel form also for n_dev launches and the 256 -> 512 layer.
  const int mode = (int)dp_policy().conv_rows;
  if (mode == 0) return false;
  if (mode != 2 && p->n_dev != nullptr) return false;
  const bool shape = (p->Cin == 512 && p->Cout % 32 == 0) || (p->Cin == 256 && p->Cout % 64 == 0 && mode == 2);
  return shape && p->W > 0 && rows_width_ok(p->W) && rows_common_ok(p, 16 / gcd_i(p->W, 16));
}

bool dp_conv_rows2_ok(const dp_conv_params* p) { return rows2_ok(p); }

// 32-bit buffer offsets inside the kernels: a launch whose tensors exceed 2 GiB (thousands of ROI slots) goes as several launches over
// chunks of whole strip groups - NOT to another kernel class: these kernels' summation order is their own, and an image's result
// must not depend on how many others are in the batch. Policy key rows_chunk_bytes (tests) lowers the limit.
int dp_conv_rows_launch(const dp_conv_params* p, dp_stream_t stream) {
  RowsArgs a;
  a.w = p->weight; a.bias = p->bias; a.n_dev = p->n_dev;
  a.H = p->H; a.W = p->W; a.relu = p->relu; a.opitch = (int)p->osW; a.kpad = p->Kpad;
  const bool two = rows2_ok(p);
  const int sw = two ? 32 : 16;
  a.G = sw / gcd_i(p->W, sw);
  a.SPG = a.G * p->W / sw;
  const int nc = p->Cin == 512 ? 32 : 64;
  a.n_slices = p->Cout / nc;
  int groups = rows_num_cus() / (8 * a.n_slices);
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8;
  a.dbg = nullptr;
  a.lockstep = 0;
  hipStream_t s = as_stream(stream);
  const DpPolicy& pol = dp_policy();
  if (two) a.lockstep = pol.conv_rows2_lockstep == 1;    // policy key conv_rows2_lockstep: 1 = all eight waves on one schedule
#ifdef DP_EXPERIMENTS
  const bool chain = !two && p->Cin == 512 && pol.conv_rows_chain == 1;   // the barrier-free chain form for the 512-channel layers
#endif
  const long long in_img = (long long)p->H * p->W * p->Cin * 2, out_img = (long long)p->H * p->W * p->osW * 2;
  long long lim = pol.rows_chunk_bytes;
  if (lim < 1) lim = 1;
  long long per = lim / (in_img > out_img ? in_img : out_img);
  per = per / a.G * a.G;                                 // whole strip groups
  if (per < a.G) per = a.G;
  for (long long n0 = 0; n0 < p->N || n0 == 0; n0 += per) {
    const int n = (int)(p->N - n0 < per ? p->N - n0 : per);
    if (n <= 0) break;
    a.in = reinterpret_cast<const unsigned char*>(p->in) + n0 * in_img;
    a.out = reinterpret_cast<unsigned char*>(p->out) + n0 * out_img;
    a.N = n;
    a.n_base = (int)n0;
    a.in_bytes = (unsigned)(n * in_img);
    a.out_bytes = (unsigned)(n * out_img);
    int rc;
    if (two) {
      if (p->Cin == 256) rc = p->dtype == DP_BF16 ? launch_rows2<uint16_t, 256>(a, s) : launch_rows2<f16_t, 256>(a, s);
      else rc = p->dtype == DP_BF16 ? launch_rows2<uint16_t, 512>(a, s) : launch_rows2<f16_t, 512>(a, s);
#ifdef DP_EXPERIMENTS
    } else if (chain) {
      rc = p->dtype == DP_BF16 ? launch_chain<uint16_t>(a, s) : launch_chain<f16_t>(a, s);
#endif
    } else {
      rc = p->dtype == DP_BF16 ? launch_rows<uint16_t>(a, p->Cin, s) : launch_rows<f16_t>(a, p->Cin, s);
    }
    if (rc != DP_OK) return rc;
  }
  return DP_OK;
}
