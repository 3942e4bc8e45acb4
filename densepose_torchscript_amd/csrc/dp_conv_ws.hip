// Weight-stationary 3x3 convolution, 128 -> 128 channels (the conv2 layers of res3,
// /root/reference/detectron2/modeling/backbone/resnet.py:195-197), 16-bit storage.
//
// On the LDS-ring kernels this layer runs at 0.22 of the MFMA peak: per 64-byte K plane a workgroup re-stages 128 weight rows
// and its pixel rows, waits, synchronises - 36 plane steps whose cost is latency, not arithmetic. But the
// whole 128 x 1152 weight matrix is only 288 KiB: it fits the REGISTER FILE of one CU (512 KiB). So here nothing about the
// weights ever moves after the prologue:
//   * wave w of a workgroup (8 waves, one workgroup per CU) owns output channels 16w .. 16w + 15 and keeps their 36 K-step
//     fragments (144 VGPRs) for the whole launch;
//   * the workgroup walks down a 16-pixel-wide column strip, RP = 2 output rows per step. The input rows it needs (r - 1 .. r + RP,
//     18 pixels x 256 B each: the strip plus a halo pixel each side) live in an LDS ring, filled one step ahead with
//     whole-line loads; zero padding is decided at the load (out-of-range buffer offsets return 0);
//   * every wave reads the same B fragments from LDS (48 per step: 4 input rows x 3 column taps x 4 channel blocks; the fragment
//     of input row q feeds output row t with kernel row q - t) and runs 72 MFMAs on them: no barrier, no global traffic, no
//     weight traffic inside a step;
//   * the 16-channel slices of the 8 waves are assembled into whole 256-byte pixel rows in an LDS buffer and stored as
//     whole lines. Two barriers per step. (RP = 4 halves the barriers per row and measured 5 % SLOWER: the step is not
//     barrier-bound.)
// K order per output pixel is the packed one (channel block major, taps inner), so results are bit-identical to the ring kernel.
#include "dp_common.h"
#include "dp_mma.h"
#include <stdlib.h>

namespace {

constexpr int kWsPx = 18;                    // staged pixels per input row: 16 + one halo pixel each side
constexpr int kWsRow = kWsPx * 256;          // bytes
#ifndef DP_WS_RP
#define DP_WS_RP 2
#endif
#ifndef DP_EXP
#define DP_EXP 0
#endif
constexpr int kWsRP = DP_WS_RP;              // output rows per step
constexpr int kWsSlots = kWsRP <= 2 ? 8 : 16; // ring slots >= RP + 2 rows in use + RP being prefetched
constexpr int kWsOut = kWsRP * 16 * 256;     // RP output rows x 16 pixels x 128 channels
constexpr int kWsLds = kWsSlots * kWsRow + kWsOut;   // 90,112 B
constexpr int kWsRowItems = kWsPx * 16;      // 16-byte items of one staged row

struct WsArgs {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  int N, H, W, relu, kpad;
  int n_strips, n_seg, seg_rows, n_jobs;
  unsigned bytes;     // extent of in / out
};

// chunk c (16 B) of pixel p of a row buffer; any 16 consecutive pixels at one chunk index hit 16 different 16-byte slots
__device__ __forceinline__ int ws_addr(int p, int c) { return p * 256 + ((c ^ (p & 15)) << 4); }

template <typename T>
__global__ __launch_bounds__(512, 2) void conv3x3_ws128_kernel(const WsArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int RP = kWsRP;
  constexpr int NF = 4 * (RP + 2) * 3;       // fragments per step: channel block x input row x column tap
  constexpr int PRE = (RP * kWsRowItems + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ring = smem;
  unsigned char* const obuf = smem + kWsSlots * kWsRow;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- this wave's 16 output channels x K = 1152, in registers for the whole launch
  u32x4 wfr[36];
  {
    const T* __restrict__ w = reinterpret_cast<const T*>(p.w) + (long long)(wave * 16 + fr) * p.kpad + fq * 8;
#pragma unroll
    for (int s = 0; s < 36; ++s) wfr[s] = *reinterpret_cast<const u32x4*>(w + s * 32);
  }
  // physical weight row 16 w + 4 fq + e carries logical channel c_l + e (pack.py row permutation inside every 64-cout block)
  const int wi = wave & 3;
  const int c_l = (wave >> 2) * 64 + (wi >> 1) * 32 + fq * 8 + (wi & 1) * 4;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + c_l);
  const int gran = c_l >> 2;     // 8-byte granule of the pixel's 256 bytes this lane produces

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.bytes, 0x00020000);
  constexpr int OOB = (int)0x80000000;

  // Work = steps (RP output rows of one 16-pixel column strip of one image), numbered column strip by column strip, top to
  // bottom. Workgroup g takes the contiguous range [g S / G, (g + 1) S / G): every CU gets the same number of steps (+- 1) and
  // at most two (re)starts of the row ring - no job queue, no imbalance from unequal job counts.
  const int steps_per_col = (p.H + RP - 1) / RP;
  const long long S = (long long)p.N * p.n_strips * steps_per_col;
  const int s_begin = (int)(S * blockIdx.x / gridDim.x), s_end = (int)(S * (blockIdx.x + 1) / gridDim.x);
  for (int sidx = s_begin; sidx < s_end;) {
    const int colid = sidx / steps_per_col;
    const int k0 = sidx - colid * steps_per_col;
    const int k1 = min(steps_per_col, k0 + (s_end - sidx));      // steps of this column this workgroup owns: k0 .. k1 - 1
    const int strip = colid % p.n_strips;
    const int n = colid / p.n_strips;
    const int r0 = k0 * RP;
    const int r1 = min(k1 * RP, p.H);
    const int c0 = strip * 16;
    sidx += k1 - k0;

    auto slot = [&](int row) __attribute__((always_inline)) -> int { return ((row + kWsSlots) & (kWsSlots - 1)) * kWsRow; };
    // item `it` of a row fill: staged pixel px (image column c0 - 1 + px), 16-byte chunk c
    auto row_off = [&](int row, int it) __attribute__((always_inline)) -> int {
      const int px = it >> 4, c = it & 15, col = c0 - 1 + px;
      return ((unsigned)row < (unsigned)p.H && (unsigned)col < (unsigned)p.W) ? ((n * p.H + row) * p.W + col) * 256 + c * 16 : OOB;
    };

    __syncthreads();      // the previous column is done with the ring
    for (int idx = tid; idx < (RP + 2) * kWsRowItems; idx += 512) {   // rows r0 - 1 .. r0 + RP
      const int q = idx / kWsRowItems, it = idx - q * kWsRowItems;
      const int row = r0 - 1 + q;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, row_off(row, it), 0, 0);
      *reinterpret_cast<u32x4*>(ring + slot(row) + ws_addr(it >> 4, it & 15)) = v;
    }
    __syncthreads();

    for (int r = r0; r < r1; r += RP) {
      // rows r + RP + 1 .. r + 2 RP (the next step's new rows): fetched now, written to the ring after this step's MFMAs
      u32x4 pre[PRE];
      int pre_dst[PRE];
      const bool more = r + RP < r1;
#pragma unroll
      for (int k = 0; k < PRE; ++k) {
        const int idx = tid + k * 512;
        const int q = idx / kWsRowItems, it = idx - q * kWsRowItems;
        const int row = r + RP + 1 + q;
        const bool in_range = idx < RP * kWsRowItems && more;
        pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (in_range && row <= r1) ? row_off(row, it) : OOB, 0, 0);
        pre_dst[k] = in_range ? slot(row) + ws_addr(it >> 4, it & 15) : -1;
      }

      f32x4 acc[RP];      // output rows r .. r + RP - 1
#pragma unroll
      for (int t = 0; t < RP; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragment f = (channel block cb, input row q, column tap dx): input row r - 1 + q feeds output row r + t with kernel row q - t
      auto frag = [&](int f) __attribute__((always_inline)) -> u32x4 {
        const int cb = f / (3 * (RP + 2)), q = (f % (3 * (RP + 2))) / 3, dx = f % 3;
        return *reinterpret_cast<const u32x4*>(ring + slot(r - 1 + q) + ws_addr(fr + dx, cb * 4 + fq));
      };
      constexpr int AHEAD = 4;              // fragments in flight ahead of their MFMAs (8 waves share the LDS)
      u32x4 bf[AHEAD + 1];
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) bf[f] = frag(f);
      static_for<0, NF>([&](auto ff) {
        constexpr int f = decltype(ff)::value;
        constexpr int cb = f / (3 * (RP + 2)), q = (f % (3 * (RP + 2))) / 3, dx = f % 3;
        if constexpr (f + AHEAD < NF && !((DP_EXP & 1) && f >= 1)) bf[(f + AHEAD) % (AHEAD + 1)] = frag(f + AHEAD);
        static_for<0, RP>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          if constexpr (q - t >= 0 && q - t <= 2) Mma<T>::run(wfr[cb * 9 + (q - t) * 3 + dx], bf[f % (AHEAD + 1)], acc[t]);
        });
      });

      // ---- epilogue: 4 channels of pixel fr of each output row per lane -> 8-byte granules of the output buffer
#pragma unroll
      for (int t = 0; t < RP; ++t) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[t][e] + bias[e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        const int P = t * 16 + fr;
        uint2 pk = make_uint2(Elem<T>::pack2(v[0], v[1]), Elem<T>::pack2(v[2], v[3]));
        *reinterpret_cast<uint2*>(obuf + P * 256 + (((gran >> 1) ^ (P & 15)) << 4) + (gran & 1) * 8) = pk;
      }
      if (DP_EXP & 2) continue;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < (RP * 256 + 511) / 512; ++k) {   // whole lines out: item = (pixel P of the RP x 16 block, chunk c)
        const int idx = tid + k * 512;
        const int P = (idx >> 4) & (RP * 16 - 1), c = idx & 15;
        const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + ws_addr(P, c));
        const int row = r + (P >> 4), col = c0 + (P & 15);
        const bool ok = idx < RP * 256 && row < r1 && col < p.W;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? ((n * p.H + row) * p.W + col) * 256 + c * 16 : OOB, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < PRE; ++k)
        if (pre_dst[k] >= 0) *reinterpret_cast<u32x4*>(ring + pre_dst[k]) = pre[k];
      __syncthreads();
    }
  }
}

int ws_num_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}

template <typename T>
int launch_ws(WsArgs a, hipStream_t stream) {
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws128_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kWsLds);
    cus = ws_num_cus();
    attr_set = true;
  }
  a.n_strips = (a.W + 15) / 16;
  const long long steps = (long long)a.N * a.n_strips * ((a.H + kWsRP - 1) / kWsRP);
  a.n_seg = a.seg_rows = 0;
  a.n_jobs = (int)steps;
  const int gx = steps < cus ? (int)steps : cus;
  hipLaunchKernelGGL((conv3x3_ws128_kernel<T>), dim3(gx), dim3(512), kWsLds, stream, a);
  return dp_check_launch("conv3x3_ws128_kernel");
}

}  // namespace

// used by dp_conv2d_nhwc (dp_conv.hip): is this launch the 128 -> 128 3x3 / pad 1 / stride 1 layer the kernel is written for?
bool dp_conv_ws128_ok(const dp_conv_params* p) {
  const char* e = getenv("DP_CONV_WS");    // A/B knob: 0 keeps the layer on the ring kernels
  if (e && atoi(e) == 0) return false;
  const long long M = (long long)p->N * p->H * p->W;
  return (p->dtype == DP_BF16 || p->dtype == DP_F16) && p->Cin == 128 && p->Cout == 128 && p->Cout_w == 128 && p->ntaps == 9 &&
         p->Kpad == 1152 && p->stride == 1 && (p->stride_w == 0 || p->stride_w == 1) && p->hi_off == -1 && p->wi_off == -1 &&
         p->H == p->Ho && p->W == p->Wo && !p->residual && !p->out_f32 && !p->head_out && p->out && p->osW == 128 &&
         p->osH == (long long)p->W * 128 && p->osN == (long long)p->H * p->W * 128 && M >= 2048 && M * 256 < (1ll << 31);
}

int dp_conv_ws128_launch(const dp_conv_params* p, dp_stream_t stream) {
  WsArgs a;
  a.in = p->in; a.w = p->weight; a.bias = p->bias; a.out = p->out;
  a.N = p->N; a.H = p->H; a.W = p->W; a.relu = p->relu; a.kpad = p->Kpad;
  a.n_strips = a.n_seg = a.seg_rows = a.n_jobs = 0;
  a.bytes = (unsigned)((long long)p->N * p->H * p->W * 256);
  hipStream_t s = as_stream(stream);
  return p->dtype == DP_BF16 ? launch_ws<uint16_t>(a, s) : launch_ws<f16_t>(a, s);
}
