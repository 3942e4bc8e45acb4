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
};

// Compiler-level fence that eight VGPR values pass through: memory operations stay on their side of it and the values must exist
// when it is reached. Device pass only: "v" is no x86 register constraint, and a kernel body the HOST pass cannot parse is dropped
// without a diagnostic - the library then fails to load with the kernel's host stub undefined.
#if defined(__HIP_DEVICE_COMPILE__)
#define DP_ROWS_PIN8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : : "memory")
#else
#define DP_ROWS_PIN8(a) ((void)0)
#endif

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
  constexpr int NU = NCT / 2;                   // reduction units of 32 couts per finished row
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

  const int n_live = p.n_dev ? min(*p.n_dev, p.N) : p.N;
  const int n_strips = ((n_live + p.G - 1) / p.G) * p.SPG;
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
  const int unit = NU == 1 ? 0 : (wave & 1);    // the reduction unit this wave takes when it is its turn
  // the slice's bias waits in LDS behind the staging buffers (8 registers less in a kernel that sits at the 256-register line)
  float* const bias_s = reinterpret_cast<float*>(smem + RING + 2 * STGB);
  if (tid < NCT * 16) bias_s[tid] = p.bias[cout0 + tid];
  __syncthreads();

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
  auto fetch = [&](const RowsIt& it, bool live, int slot) __attribute__((always_inline)) {
    const bool row_ok = live && (unsigned)it.q < (unsigned)p.H;     // rows -1 and H are zero padding: nothing is read (nor computed)
    const int roff = it.q * in_row;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, DP_LDS_PTR(ring_w + slot * ROWB + pc * 1024), 16, row_ok ? f_boff[pc] + roff : OOB, 0, 0, 0);
  };

  // ---- compute side: lane (fr, fq) = output pixel fr of the strip; its staged pixel at column tap dx is fr + dx (+ 1 in the
  // second segment, behind the shared zero pixel)
  int c_frag = 0, c_obase = OOB;
  auto setup_comp = [&](int strip) __attribute__((always_inline)) {
    int img0, c00, len0;
    decode(strip, img0, c00, len0);
    const bool in1 = fr >= len0;
    c_frag = (fr + (in1 ? 1 : 0)) * PPW + fq * 16;
    const int img = img0 + (in1 ? 1 : 0), col = in1 ? fr - len0 : c00 + fr;
    c_obase = img < n_live ? ((img * p.H * p.W + col) * p.opitch + cout0 + unit * 32 + fq * 8) * 2 : OOB;
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
  // prologue: the first D rows
  int f_done = 0;     // steps whose fetch has been issued
  auto fetch_next = [&]() __attribute__((always_inline)) {
    const bool live = f_done < n_steps;
    int slot = f_done % NSLOT;
    fetch(it_f, live, slot);
    ++f_done;
    if (live) {
      if (it_f.q == it_f.r_hi) {
        if (it_f.strip < s_last) { seg_init(it_f, it_f.strip + 1); setup_fetch(it_f.strip); }
      } else {
        ++it_f.q;
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

  unsigned char* const stg = smem + RING;
  int ec = 0;          // emitted rows so far: staging parity and whose turn the reduction is

  // one step with the accumulator roles fixed at compile time: old = acc[(PH + 2) % 3] (output row q - 1, kernel row 2, complete after
  // this step), mid = acc[PH] (row q, kernel row 1), fresh = acc[(PH + 1) % 3] (row q + 1, kernel row 0, starts from zero)
  auto step = [&](auto ph_c, int s) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    constexpr int A_OLD = (PH + 2) % 3, A_MID = PH, A_NEW = (PH + 1) % 3;
    fetch_next();                                  // row of step s + D -> the slot step s - 1 has just finished reading
    rows_wait_vm<NP * D>();                        // ... and the row of step s has landed (only the D younger rows may be in flight)
    const int t = it_c.q - 1;                      // the output row that is complete after this step
    const bool emit = t >= it_c.r_lo;
    unsigned char* const sw = stg + (ec & 1) * STGB + wave * (NCT * 1024) + lane * 16;
    const bool row_ok = (unsigned)it_c.q < (unsigned)p.H;
    // Three passes over the step's fragments, one per accumulator role, the finished row first: its partial sums go to the staging
    // buffer before the fresh row's accumulators are born, so only two of the three sets are live at any time (NCT = 4: 32 instead
    // of 48 registers in a kernel that sits at the 256-register line). Per output pixel the order of the products is unchanged.
    u32x4 bf[NF];
    if (row_ok) {
      const unsigned char* const row = ring_w + (s % NSLOT) * ROWB + c_frag;
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        bf[f] = *reinterpret_cast<const u32x4*>(row + (f % 3) * PPW + (f / 3) * 64);
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Mma<T>::run(wfr[(ct * NCB + f / 3) * 9 + 6 + f % 3], bf[f], acc[A_OLD][ct]);
      });
    }
    if (emit) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(sw + ct * 1024) = acc[A_OLD][ct];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[A_NEW][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Mma<T>::run(wfr[(ct * NCB + f / 3) * 9 + 3 + f % 3], bf[f], acc[A_MID][ct]);
      });
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Mma<T>::run(wfr[(ct * NCB + f / 3) * 9 + 0 + f % 3], bf[f], acc[A_NEW][ct]);
      });
    }
    if (emit) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool my_turn = NU == 1 ? (wave == (ec & 7)) : ((wave >> 1) == (ec & 3));
      if (my_turn) {
        const unsigned char* const sr = stg + (ec & 1) * STGB + unit * 2048 + lane * 16;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        // two partials (four reads) in flight at a time: left alone the scheduler issues all 16 reads first and keeps their 64
        // registers live to the end (a sched_barrier does not stop it; a compiler-level fence that ALL eight running sums pass
        // through does) - the kernel sits at the 256-register line and spilled weight fragments
#pragma unroll
        for (int w8 = 0; w8 < 8; w8 += 2) {
          const f32x4 lo0 = *reinterpret_cast<const f32x4*>(sr + w8 * (NCT * 1024));
          const f32x4 hi0 = *reinterpret_cast<const f32x4*>(sr + w8 * (NCT * 1024) + 1024);
          const f32x4 lo1 = *reinterpret_cast<const f32x4*>(sr + (w8 + 1) * (NCT * 1024));
          const f32x4 hi1 = *reinterpret_cast<const f32x4*>(sr + (w8 + 1) * (NCT * 1024) + 1024);
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[k] += lo0[k]; v[4 + k] += hi0[k]; }
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[k] += lo1[k]; v[4 + k] += hi1[k]; }
          DP_ROWS_PIN8(v);
        }
        const f32x4 bz0 = *reinterpret_cast<const f32x4*>(bias_s + unit * 32 + fq * 8), bz1 = *reinterpret_cast<const f32x4*>(bias_s + unit * 32 + fq * 8 + 4);
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float x0 = v[2 * k] + (k < 2 ? bz0[2 * k] : bz1[2 * k - 4]), x1 = v[2 * k + 1] + (k < 2 ? bz0[2 * k + 1] : bz1[2 * k - 3]);
          if (p.relu) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
          pk[k] = Elem<T>::pack2(x0, x1);
        }
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, c_obase + t * out_row, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      ++ec;
    }
    // next step of the compute walk
    if (it_c.q == it_c.r_hi) {
      if (it_c.strip < s_last) { seg_init(it_c, it_c.strip + 1); setup_comp(it_c.strip); }
    } else {
      ++it_c.q;
    }
  };

  int s = 0;
  for (; s + 3 <= n_steps; s += 3) {
    step(std::integral_constant<int, 0>{}, s);
    step(std::integral_constant<int, 1>{}, s + 1);
    step(std::integral_constant<int, 2>{}, s + 2);
  }
  if (s < n_steps) step(std::integral_constant<int, 0>{}, s);
  if (s + 1 < n_steps) step(std::integral_constant<int, 1>{}, s + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy fetches behind the last step still target this workgroup's LDS
}

template <typename T, int CIN, int NCT>
int launch_rows_r(const RowsArgs& a, hipStream_t stream) {
  constexpr int KPW = CIN / 8, PPW = KPW * 2 + 32, NP = (19 * PPW + 1023) / 1024;
  constexpr int lds = 8 * 4 * NP * 1024 + 2 * 8 * NCT * 1024 + NCT * 16 * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rows_kernel<T, CIN, NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_rows_kernel<T, CIN, NCT>), dim3(a.n_pg * a.n_slices), dim3(512), lds, stream, a);
  return dp_check_launch("conv3x3_rows_kernel");
}

template <typename T>
int launch_rows(const RowsArgs& a, int cin, hipStream_t stream) {
  return cin == 512 ? launch_rows_r<T, 512, 2>(a, stream) : launch_rows_r<T, 256, 4>(a, stream);
}

int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): is this launch a 3x3 / pad 1 / stride 1 layer the kernel is written for? No size thresholds:
// the kernel's summation order differs from the LDS-ring kernels', so a layer either always runs here or never (batch invariance).
bool dp_conv_rows_ok(const dp_conv_params* p) {
  const char* e = getenv("DP_CONV_ROWS");    // A/B knob: 0 keeps these layers on the other kernels; 2 also takes the 256-channel layers
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) return false;
  const bool shape = (p->Cin == 512 && p->Cout % 32 == 0) || (p->Cin == 256 && p->Cout % 64 == 0 && (mode == 2 || p->Cout == 512));
  const int g = 16 / gcd_i(p->W > 0 ? p->W : 16, 16);
  return (p->dtype == DP_BF16 || p->dtype == DP_F16) && shape && p->ntaps == 9 && p->Kpad == 9 * p->Cin && p->stride == 1 &&
         (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == -1 && p->wi_off == -1 && p->H == p->Ho && p->W == p->Wo && p->W >= 16 &&
         !p->residual && !p->out_f32 && !p->head_out && !p->in2 && !p->post_res && p->post_mode == 0 && p->split_k <= 1 && p->out &&
         p->osW >= p->Cout && p->osW % 8 == 0 && p->osH == (long long)p->W * p->osW && p->osN == (long long)p->H * p->W * p->osW &&
         p->Cout <= p->Cout_w && p->Cout_w % 64 == 0 &&
         (long long)(p->N + g) * p->H * p->W * p->Cin * 2 < (1ll << 31) && (long long)(p->N + g) * p->H * p->W * p->osW * 2 < (1ll << 31);
}

int dp_conv_rows_launch(const dp_conv_params* p, dp_stream_t stream) {
  RowsArgs a;
  a.in = p->in; a.w = p->weight; a.bias = p->bias; a.out = p->out; a.n_dev = p->n_dev;
  a.N = p->N; a.H = p->H; a.W = p->W; a.relu = p->relu; a.opitch = (int)p->osW; a.kpad = p->Kpad;
  a.G = 16 / gcd_i(p->W, 16);
  a.SPG = a.G * p->W / 16;
  const int nc = p->Cin == 512 ? 32 : 64;
  a.n_slices = p->Cout / nc;
  int groups = rows_num_cus() / (8 * a.n_slices);
  if (groups < 1) groups = 1;
  a.n_pg = groups * 8;
  a.in_bytes = (unsigned)((long long)p->N * p->H * p->W * p->Cin * 2);
  a.out_bytes = (unsigned)((long long)p->N * p->H * p->W * p->osW * 2);
  hipStream_t s = as_stream(stream);
  return p->dtype == DP_BF16 ? launch_rows<uint16_t>(a, p->Cin, s) : launch_rows<f16_t>(a, p->Cin, s);
}
