// Stem in ONE launch: 7x7 stride-2 convolution + FrozenBN + ReLU + 3x3 stride-2 max-pool
// (/root/reference/detectron2/modeling/backbone/resnet.py:350-354), for the 64-channel stem in 16-bit storage.
//
// Layer by layer the stem writes its 400x672x64 output (275 MB at batch 8) and the pool reads it back to produce a quarter of it;
// fused, the launch reads the 137 MB image and writes the 69 MB pooled map. A workgroup (8 waves) walks DOWN a strip of 56
// pooled columns, one pooled row per step:
//   * the image arrives in the paired-pixel layout (dp_preprocess_u8, paired = 1): cell c of a row holds the 4-channel pixels
//     2c - 3, 2c - 2, so conv column c reads cells c .. c + 3 and one kernel row is one 32-element K step (pack.stem_paired_conv).
//     The input rows a step needs live in a 16-slot LDS ring (132 cells x 16 B each); 4 new rows are fetched per step, whole
//     lines, one step ahead. The B fragment of (conv column tile, kernel row) is one ds_read_b128 at cell (column + lane >> 4).
//   * the 64 x 224 weight matrix stays in REGISTERS for the whole launch (7 K steps x 4 cout tiles = 112 VGPRs per lane).
//   * the two new conv rows of a step (2i, 2i + 1; row 2i - 1 is left over from the previous step) are written, bias + ReLU
//     applied and rounded to the storage type exactly like the separate conv launch, into a 5-slot LDS ring of conv rows
//     (128 columns x 64 channels); the pool then takes the 3 x 3 maximum from LDS and stores whole 128-byte lines.
//     Positions outside the conv map hold -inf, so the maximum runs over the same elements as F.max_pool2d's padding rule.
// Same K order and rounding as conv + pool run separately: bit-identical results (tests/test_gpu_kernels.py).
#include "dp_common.h"
#include "dp_mma.h"

namespace {

constexpr int kPoolCols = 56;               // pooled columns per strip
constexpr int kConvCols = 128;              // conv columns computed per strip (8 MFMA tiles): 2 * 56 + 1 = 113 are used
constexpr int kCells = 132;                 // input cells staged per row: 128 + 3, rounded up
constexpr int kInRow = kCells * 16;         // bytes
constexpr int kInSlots = 16;
constexpr int kConvRow = kConvCols * 128;   // bytes: 64 channels x 2 B per column
constexpr int kConvSlots = 5;               // conv rows 2 i - 3 .. 2 i + 1 are alive while step i is computed and step i - 1 is pooled
constexpr int kStemLds = kConvSlots * kConvRow + kInSlots * kInRow;   // 115,712 B

struct StemArgs {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  int N, Hp, Wq, Hc, Wc, Ho, Wo, kpad;
  int n_strips, n_seg, seg_rows, n_jobs;
  unsigned in_bytes, out_bytes;
};

// conv-row buffer: column p is stored at position pi(p) (bits 0 and 1 swapped: the pool reads every second column, which would
// otherwise use only half of the 64 banks), chunk c of a position at slot c ^ f(pi)
__device__ __forceinline__ int conv_addr(int p, int c) {
  const int pi = (p & ~3) | ((p & 1) << 1) | ((p >> 1) & 1);
  const int f = ((pi >> 1) & 3) | ((pi & 1) << 2);
  return pi * 128 + ((c ^ f) << 4);
}

template <typename T>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const StemArgs p) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const conv_s = smem;                      // [5][128 columns][128 B]
  unsigned char* const in_s = smem + kConvSlots * kConvRow;   // [16][132 cells][16 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  constexpr unsigned NEG_INF2 = sizeof(T) == 2 && std::is_same<T, uint16_t>::value ? 0xFF80FF80u : 0xFC00FC00u;   // two -inf

  // ---- weights: 7 K steps (kernel rows) x 4 cout tiles of 16, straight into registers, once
  u32x4 wfr[7][4];
  {
    // tiled weight matrix (dp_wtile_off): cout tile i, K plane dy (one kernel row = 4 cells x 8 channels = 64 bytes)
    const unsigned char* __restrict__ w = reinterpret_cast<const unsigned char*>(p.w) + dp_wtile_off(fr, 0, fq, p.kpad * 2 / 64);
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
      for (int i = 0; i < 4; ++i) wfr[dy][i] = *reinterpret_cast<const u32x4*>(w + ((long long)i * (p.kpad * 2 / 64) + dy) * 1024);
  }
  float bias[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + h * 32 + fq * 8);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + h * 32 + fq * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { bias[h][k] = b0[k]; bias[h][4 + k] = b1[k]; }
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
  constexpr int OOB = (int)0x80000000;

  for (int job = blockIdx.x; job < p.n_jobs; job += gridDim.x) {
    const int seg = job % p.n_seg;
    const int jt = job / p.n_seg;
    const int strip = jt % p.n_strips;
    const int n = jt / p.n_strips;
    const int i0 = seg * p.seg_rows;
    const int i1 = min(i0 + p.seg_rows, p.Ho);
    const int j0 = strip * kPoolCols;
    const int cb = 2 * j0 - 1;                 // conv column of buffer column 0 (= input cell of staged cell 0)

    // input row `row`, staged cell k  <-  global cell cb + k (zeros outside the image: the padding of the convolution)
    auto in_off = [&](int row, int k) __attribute__((always_inline)) -> int {
      const int cell = cb + k;
      return ((unsigned)row < (unsigned)p.Hp && (unsigned)cell < (unsigned)p.Wq) ? ((n * p.Hp + row) * p.Wq + cell) * 16 : OOB;
    };
    auto in_slot = [&](int row) __attribute__((always_inline)) -> int { return ((row + 32) & (kInSlots - 1)) * kInRow; };

    // one conv row y, column tile t -> conv ring slot (y mod 3)
    auto conv_tile = [&](int y, int t) __attribute__((always_inline)) {
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 7; ++dy) {
        const u32x4 b = *reinterpret_cast<const u32x4*>(in_s + in_slot(2 * y - 3 + dy) + (16 * t + fr + fq) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(wfr[dy][i], b, acc[i]);
      }
      const int col = 16 * t + fr;
      const bool col_ok = (unsigned)(cb + col) < (unsigned)p.Wc;
      unsigned char* const dst = conv_s + ((y + kConvSlots) % kConvSlots) * kConvRow;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = fmaxf(acc[2 * h][k] + bias[h][k], 0.f);
          v[4 + k] = fmaxf(acc[2 * h + 1][k] + bias[h][4 + k], 0.f);
        }
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = col_ok ? Elem<T>::pack2(v[2 * k], v[2 * k + 1]) : NEG_INF2;
        *reinterpret_cast<u32x4*>(dst + conv_addr(col, h * 4 + fq)) = pk;
      }
    };

    // ---- prologue: input rows 4 i0 - 5 .. 4 i0 + 5, conv row 2 i0 - 1 (or -inf above the map)
    __syncthreads();   // the previous job is done with both rings
    for (int idx = tid; idx < 11 * kCells; idx += 512) {
      const int r = idx / kCells, k = idx - r * kCells;
      const int row = 4 * i0 - 5 + r;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, in_off(row, k), 0, 0);
      *reinterpret_cast<u32x4*>(in_s + in_slot(row) + k * 16) = v;
    }
    __syncthreads();
    if (2 * i0 - 1 >= 0) {
      conv_tile(2 * i0 - 1, wave);
    } else {
      const u32x4 ninf = {NEG_INF2, NEG_INF2, NEG_INF2, NEG_INF2};
      for (int idx = tid; idx < kConvRow / 16; idx += 512) *reinterpret_cast<u32x4*>(conv_s + (kConvSlots - 1) * kConvRow + idx * 16) = ninf;   // row -1
    }

    // Steps, software-pipelined over the two phases (round 6): iteration i computes the conv rows of step i AND pools step i - 1, with ONE
    // barrier per step. The two waves of a SIMD (w and w + 4) take the phases in opposite order - while one runs its 56 MFMAs the other does
    // the pool's LDS reads and VALU - where every wave used to wait at a barrier between "all convolve" and "all pool" (two per step, the
    // matrix pipe idle through the pool). Five conv-row slots: step i writes rows 2 i, 2 i + 1 while the pool still reads 2 i - 3 .. 2 i - 1;
    // the slots of rows 2 i + 2, 2 i + 3 are those of 2 i - 3, 2 i - 2, free after this iteration's barrier. The 16-slot input ring never
    // needed the second barrier: the rows fetched for step i + 1 replace rows 4 i - 10 .. 4 i - 7, dead since step i - 2.
    auto pool_step = [&](int ip) __attribute__((always_inline)) {
      // 3 x 3 max over conv rows 2 ip - 1 .. 2 ip + 1, columns 2 j - 1 .. 2 j + 1 (buffer columns 2 jl .. 2 jl + 2); one thread per
      // (pooled column, 8-channel chunk); rows / columns outside the map hold -inf
      if (tid < kPoolCols * 8) {
        const int jl = tid >> 3, cidx = tid & 7;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const unsigned char* src = conv_s + ((2 * ip - 1 + dy + kConvSlots) % kConvSlots) * kConvRow;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + conv_addr(2 * jl + dx, cidx));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              m[2 * k] = fmaxf(m[2 * k], Elem<T>::unpack(v[k] & 0xffffu));
              m[2 * k + 1] = fmaxf(m[2 * k + 1], Elem<T>::unpack(v[k] >> 16));
            }
          }
        }
        u32x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = Elem<T>::pack2(m[2 * k], m[2 * k + 1]);
        const int j = j0 + jl;
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, j < p.Wo ? ((n * p.Ho + ip) * p.Wo + j) * 128 + cidx * 16 : OOB, 0, 0);
      }
    };
    auto conv_step = [&](int ic) __attribute__((always_inline)) {
      // conv rows 2 ic (waves 0-3) and 2 ic + 1 (waves 4-7), two column tiles per wave
      const int y = 2 * ic + (wave >> 2);
      conv_tile(y, (wave & 3) * 2);
      conv_tile(y, (wave & 3) * 2 + 1);
    };
    for (int i = i0; i <= i1; ++i) {     // (conv row 2 i0 - 1 above is first read by the pool in iteration i0 + 1, behind a barrier)
      // the 4 input rows the NEXT step adds (4 i + 6 .. 4 i + 9), fetched now, written to the ring behind this iteration's work
      u32x4 pre[2];
      int pre_dst[2];
      const bool more = i + 1 < i1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * 512;
        const int r = idx / kCells, k = idx - r * kCells;
        const int row = 4 * i + 6 + r;
        const bool use = more && idx < 4 * kCells;
        pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, use ? in_off(row, k) : OOB, 0, 0);
        pre_dst[q] = use ? in_slot(row) + k * 16 : -1;
      }
      if (wave < 4) {
        if (i < i1) conv_step(i);
        if (i > i0) pool_step(i - 1);
      } else {
        if (i > i0) pool_step(i - 1);
        if (i < i1) conv_step(i);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (pre_dst[q] >= 0) *reinterpret_cast<u32x4*>(in_s + pre_dst[q]) = pre[q];
      __syncthreads();
    }
  }
}

static int stem_num_cus() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    return prop.multiProcessorCount;
  return 256;
}

template <typename T>
int launch_stem(const StemArgs& a0, hipStream_t stream) {
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, kStemLds);
    cus = stem_num_cus();
    attr_set = true;
  }
  StemArgs a = a0;
  a.n_strips = (a.Wo + kPoolCols - 1) / kPoolCols;
  // pooled rows per job: about four jobs per CU (each job recomputes one conv row above its first pooled row), at least 4
  const long long cols = (long long)a.N * a.n_strips;
  long long want = (4ll * cus + cols - 1) / cols;
  if (want < 1) want = 1;
  int seg_rows = (int)((a.Ho + want - 1) / want);
  if (seg_rows < 4) seg_rows = a.Ho < 4 ? a.Ho : 4;
  a.seg_rows = seg_rows;
  a.n_seg = (a.Ho + seg_rows - 1) / seg_rows;
  a.n_jobs = (int)(cols * a.n_seg);
  const int gx = a.n_jobs < cus ? a.n_jobs : cus;
  hipLaunchKernelGGL((stem_pool_kernel<T>), dim3(gx), dim3(512), kStemLds, stream, a);
  return dp_check_launch("stem_pool_kernel");
}

}  // namespace

static const char* stem_unsupported(const dp_stem_pool_params* p) {
  if (p->dtype != DP_BF16 && p->dtype != DP_F16) return "16-bit storage only";
  if (p->Cout != 64 || p->Kpad != 256) return "only the 64-channel stem in the paired-pixel packing (Kpad 256) is fused";
  if (p->Hp % 4 != 0 || p->Wp % 4 != 0) return "padded image size must be a multiple of 4";
  const long long cells = (long long)p->N * p->Hp * (p->Wp / 2 + 3);
  const long long outs = (long long)p->N * (p->Hp / 4) * (p->Wp / 4);
  if (cells * 16 >= (1ll << 31) || outs * 128 >= (1ll << 31)) return "tensor too large for 32-bit buffer offsets (split the batch)";
  return nullptr;
}

extern "C" int dp_stem_pool_supported(const dp_stem_pool_params* p) {
  if (!p) return 0;
  return stem_unsupported(p) == nullptr ? 1 : 0;
}

extern "C" int dp_stem_pool_nhwc(const dp_stem_pool_params* p, dp_stream_t stream) {
  DP_REQUIRE(p != nullptr, "dp_stem_pool_nhwc: null params");
  DP_REQUIRE(p->N >= 0 && p->Hp > 0 && p->Wp > 0, "dp_stem_pool_nhwc: bad shape");
  const char* why = stem_unsupported(p);
  if (why) return dp_fail(DP_ERR_UNSUPPORTED, "dp_stem_pool_nhwc: %s", why);
  if (p->N == 0) return DP_OK;
  DP_REQUIRE(p->in && p->weight && p->bias && p->out, "dp_stem_pool_nhwc: null pointer");
  StemArgs a;
  a.in = p->in; a.w = p->weight; a.bias = p->bias; a.out = p->out;
  a.N = p->N; a.Hp = p->Hp; a.Wq = p->Wp / 2 + 3; a.Hc = p->Hp / 2; a.Wc = p->Wp / 2;
  a.Ho = (a.Hc - 1) / 2 + 1; a.Wo = (a.Wc - 1) / 2 + 1; a.kpad = p->Kpad;
  a.n_strips = a.n_seg = a.seg_rows = a.n_jobs = 0;
  a.in_bytes = (unsigned)((long long)a.N * a.Hp * a.Wq * 16);
  a.out_bytes = (unsigned)((long long)a.N * a.Ho * a.Wo * 128);
  hipStream_t s = as_stream(stream);
  return p->dtype == DP_BF16 ? launch_stem<uint16_t>(a, s) : launch_stem<f16_t>(a, s);
}
