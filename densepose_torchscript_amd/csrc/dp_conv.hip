// NHWC implicit-GEMM convolution on MFMA for gfx950 (MI355X).
//
//   out[m][co] = act( sum_k A[m][k] * Wt[co][k] + bias[co] (+ residual) ),  m = (n, ho, wo), k = (tap, c)
//
// Replaces every F.conv2d / nn.Linear / ConvTranspose2d call site of the reference's hot path
// (detectron2/layers/wrappers.py:105-111 and the callers listed in include/densepose_hip.h).
//
// Design (wave64, 4 waves per workgroup, 128 pixels x {128|64} couts per workgroup, K-step = 128 bytes):
//   * the weight tile is the MFMA "A" operand (rows = couts) and the im2col pixel tile the "B" operand
//     (cols = pixels): the 16x16 accumulator then holds 4 CONSECUTIVE channels of one pixel per lane, so
//     the NHWC epilogue stores 8/16 contiguous bytes per lane with bias/residual/ReLU fused.
//   * both tiles are staged global -> registers -> LDS (16-byte chunks), double buffered: the loads of
//     K-step t+1 are issued before the MFMAs of step t and written to the other buffer after them, one
//     barrier per K-step. Zero padding / ragged M / K padding are handled by predicating the 16-byte load.
//   * LDS image: two planes of [rows][64 B]; chunk c of row r sits at slot c ^ ((-(r>>2))&3), plane 1 also
//     swaps row pairs (r^1): ds_write_b128 and the fragment ds_read_b128 are both bank-conflict free.
//   * the K axis is table driven (ktab: {dy, dx, c0, valid} per 16-byte chunk), so 1x1 / 3x3 / dilated /
//     7x7-stem / 2x2 sub-pixel (deconv) / fully-connected layers all run through this one kernel.
//   * DP_BF16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate). DP_F32 (parity mode): v_mfma_f32_16x16x4_f32,
//     bit-exact fp32 FMA chain; same LDS image in bytes, 4 MFMAs per fragment instead of 1.
//   * workgroup ids are remapped so that consecutive tiles (same pixel rows, neighbouring cout tiles)
//     run on the same XCD and share its L2.
#include "dp_common.h"

namespace {

// native clang vectors (HIP's uint4/int4 are structs; arrays of them ended up in scratch memory)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;       // pixels per workgroup
constexpr int kKB = 128;       // bytes of K per step
constexpr int kThreads = 256;

struct ConvArgs {
  const void* in;
  const void* weight;
  const i32x4* ktab;
  const float* bias;
  const void* residual;
  void* out;
  int N, H, W, Cin, Ho, Wo, Cout, Kpad, stride, hi_off, wi_off, relu, rshift, out_f32;
  long long osN, osH, osW, rsN, rsH, rsW;
  int M, tiles_n, n_ktiles, HoWo, n_tiles;
};

__device__ __forceinline__ int swz(int r) { return (-(r >> 2)) & 3; }

// 16 zero bytes in global memory: out-of-image / K-padding chunks are loaded from here, so every staging load is
// unconditional (a predicated load makes hipcc branch around it and wait vmcnt(0) per load: serialised round trips).
__device__ u32x4 g_zero16 = {0u, 0u, 0u, 0u};

template <typename T>
struct Mma;
template <>
struct Mma<uint16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    // whole-vector bit_cast, then index: bit_cast of a single ext-vector element (a.y ...) silently read element 0
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], c, 0, 0, 0);
  }
};

template <typename T, int BN>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int ES = sizeof(T);
  constexpr int CH = 16 / ES;           // elements per chunk
  constexpr int KT = kKB / ES;          // elements of K per step
  constexpr int A_PLANE = kBM * 64;
  constexpr int B_PLANE = BN * 64;
  constexpr int A_BUF = 2 * A_PLANE;
  constexpr int B_BUF = 2 * B_PLANE;
  constexpr int BUF = A_BUF + B_BUF;
  constexpr int A_PASSES = kBM / 32;
  constexpr int B_PASSES = BN / 32;
  // wave tiling: BN=128 -> 2(pixels) x 2(couts) waves of 64x64 ; BN=64 -> 4 x 1 waves of 32 x 64
  constexpr int WAVES_C = BN / 64;
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int TP = kBM / WAVES_P / 16;
  constexpr int TC = 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  // XCD-aware tile id (bijective remap of blockIdx over 8 XCDs)
  int tile;
  {
    const int nwg = p.n_tiles, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const int mt = tile / p.tiles_n;
  const int nt = tile - mt * p.tiles_n;
  const int m0 = mt * kBM;
  const int n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wp = wave % WAVES_P;
  const int wc = wave / WAVES_P;

  // ---- staging assignment: thread -> (row lrow + 32*i, chunk c) ----
  const int lrow = tid >> 3;
  const int c = tid & 7;
  const int plane = c >> 2;
  const int st_off = plane * 0 + ((lrow ^ plane) * 64) + (((c & 3) ^ swz(lrow)) << 4);  // + plane*PLANE added per region

  int a_hi0[A_PASSES], a_wi0[A_PASSES], a_pix[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < p.M) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride + p.hi_off;
      a_wi0[i] = wo * p.stride + p.wi_off;
      a_pix[i] = n * p.H * p.W;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_pix[i] = 0;
    }
  }
  const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.weight) + (long long)(n0 + lrow) * p.Kpad + c * CH;

  u32x4 ra[A_PASSES], rb[B_PASSES];
  const T* __restrict__ zsrc = reinterpret_cast<const T*>(&g_zero16);
  unsigned char* const st_a = smem + plane * A_PLANE + st_off;
  unsigned char* const st_b = smem + A_BUF + plane * B_PLANE + st_off;

// NOTE: plain macros, not lambdas: a closure capturing ra/rb by reference made hipcc keep them in scratch memory.
#define DP_LOAD_TILE(KT_IDX, E)                                                                        \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                             \
      const int hi = a_hi0[i] + (E).x;                                                                 \
      const int wi = a_wi0[i] + (E).y;                                                                 \
      const bool ok = (E).w && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;           \
      const long long off = (long long)(a_pix[i] + hi * p.W + wi) * p.Cin + (E).z;                     \
      const T* src = ok ? (in + off) : zsrc;                                                           \
      ra[i] = *reinterpret_cast<const u32x4*>(src);                                                    \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i)                                               \
        rb[i] = *reinterpret_cast<const u32x4*>(wgt + (long long)(32 * i) * p.Kpad + (KT_IDX) * KT);   \
  }
#define DP_STORE_TILE(BUF_IDX)                                                                         \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i)                                               \
        *reinterpret_cast<u32x4*>(st_a + (BUF_IDX) * BUF + i * 32 * 64) = ra[i];                       \
    _Pragma("unroll") for (int i = 0; i < B_PASSES; ++i)                                               \
        *reinterpret_cast<u32x4*>(st_b + (BUF_IDX) * BUF + i * 32 * 64) = rb[i];                       \
  }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (row = lane&15 inside a 16-row tile, chunk = lane>>4)
  const int fr = lane & 15;
  const int fq = lane >> 4;
  const int rd_sw = (fq ^ swz(fr)) << 4;
  const int rd_p0 = fr * 64 + rd_sw;          // plane 0
  const int rd_p1 = (fr ^ 1) * 64 + rd_sw;    // plane 1 (row pairs swapped)
  const int a_row0 = (wp * TP * 16) * 64;     // pixel rows of this wave
  const int b_row0 = (wc * TC * 16) * 64;     // cout rows of this wave

  const int nk = p.n_ktiles;
  // the tap-table entry is fetched one K-step ahead of the loads that depend on it
  i32x4 e_next = p.ktab[(nk > 1 ? 8 : 0) + c];
  {
    const i32x4 e0 = p.ktab[c];
    DP_LOAD_TILE(0, e0);
  }
  DP_STORE_TILE(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      DP_LOAD_TILE(kt + 1, e_next);
      const int kn = kt + 2 < nk ? kt + 2 : nk - 1;
      e_next = p.ktab[kn * 8 + c];
    }
    const unsigned char* sa = smem + cur * BUF + a_row0;
    const unsigned char* sb = smem + cur * BUF + A_BUF + b_row0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ro = ks ? rd_p1 : rd_p0;
      u32x4 fp[TP], fc[TC];
#pragma unroll
      for (int j = 0; j < TP; ++j) fp[j] = *reinterpret_cast<const u32x4*>(sa + ks * A_PLANE + j * 16 * 64 + ro);
#pragma unroll
      for (int i = 0; i < TC; ++i) fc[i] = *reinterpret_cast<const u32x4*>(sb + ks * B_PLANE + i * 16 * 64 + ro);
#pragma unroll
      for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) Mma<T>::run(fc[i], fp[j], acc[i][j]);
    }
    if (kt + 1 < nk) DP_STORE_TILE(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds channels co..co+3 of pixel m for each (i, j) ----
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int m = m0 + wp * TP * 16 + j * 16 + fr;
    if (m >= p.M) continue;
    const int n = m / p.HoWo;
    const int rem = m - n * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    const long long obase = n * p.osN + ho * p.osH + wo * p.osW;
    long long rbase = 0;
    if (res) rbase = n * p.rsN + (ho >> p.rshift) * p.rsH + (wo >> p.rshift) * p.rsW;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      const int co = n0 + wc * TC * 16 + i * 16 + fq * 4;
      if (co >= p.Cout) continue;
      const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
      float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
      if (res) {
        const float4 rv = load4(res + rbase + co);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (p.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (p.out_f32) store4(reinterpret_cast<float*>(p.out) + obase + co, v);
      else store4(reinterpret_cast<T*>(p.out) + obase + co, v);
    }
  }
}

#undef DP_LOAD_TILE
#undef DP_STORE_TILE

template <typename T, int BN>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (2 * kBM * 64 + 2 * BN * 64);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN>), dim3(a.n_tiles), dim3(kThreads), lds, stream, a);
  return dp_check_launch("conv_igemm_kernel");
}

}  // namespace

extern "C" int dp_conv2d_nhwc(const dp_conv_params* p, dp_stream_t stream) {
  DP_REQUIRE(p != nullptr, "dp_conv2d_nhwc: null params");
  DP_REQUIRE(p->dtype == DP_F32 || p->dtype == DP_BF16, "dp_conv2d_nhwc: bad dtype %d", p->dtype);
  const int es = p->dtype == DP_F32 ? 4 : 2;
  DP_REQUIRE(p->N >= 0 && p->H > 0 && p->W > 0 && p->Ho > 0 && p->Wo > 0, "dp_conv2d_nhwc: bad spatial shape");
  const long long M = (long long)p->N * p->Ho * p->Wo;
  if (M == 0) return DP_OK;  // R = 0 detections is legal
  DP_REQUIRE(p->in && p->weight && p->ktab && p->bias && p->out, "dp_conv2d_nhwc: null pointer");
  DP_REQUIRE(p->Cin > 0 && p->Cin % 8 == 0, "dp_conv2d_nhwc: Cin=%d must be a positive multiple of 8", p->Cin);
  DP_REQUIRE(p->Cout > 0 && p->Cout % 8 == 0 && p->Cout <= p->Cout_w, "dp_conv2d_nhwc: Cout=%d Cout_w=%d", p->Cout, p->Cout_w);
  DP_REQUIRE(p->Cout_w % 128 == 0, "dp_conv2d_nhwc: Cout_w=%d must be a multiple of 128", p->Cout_w);
  DP_REQUIRE(p->Kpad > 0 && (p->Kpad * es) % kKB == 0, "dp_conv2d_nhwc: Kpad=%d not a multiple of %d bytes", p->Kpad, kKB);
  DP_REQUIRE(p->stride >= 1, "dp_conv2d_nhwc: stride");
  DP_REQUIRE(M < (1ll << 31) && (long long)p->N * p->H * p->W * p->Cin < (1ll << 31), "dp_conv2d_nhwc: tensor too large for 32-bit pixel index");
  ConvArgs a;
  a.in = p->in; a.weight = p->weight; a.ktab = reinterpret_cast<const i32x4*>(p->ktab); a.bias = p->bias;
  a.residual = p->residual; a.out = p->out;
  a.N = p->N; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout; a.Kpad = p->Kpad;
  a.stride = p->stride; a.hi_off = p->hi_off; a.wi_off = p->wi_off; a.relu = p->relu; a.rshift = p->rshift; a.out_f32 = p->out_f32;
  a.osN = p->osN; a.osH = p->osH; a.osW = p->osW; a.rsN = p->rsN; a.rsH = p->rsH; a.rsW = p->rsW;
  a.M = (int)M; a.HoWo = p->Ho * p->Wo; a.n_ktiles = p->Kpad * es / kKB;
  const int tiles_m = (int)((M + kBM - 1) / kBM);
  hipStream_t s = as_stream(stream);
  if (p->Cout <= 64) {
    a.tiles_n = (p->Cout + 63) / 64;
    a.n_tiles = tiles_m * a.tiles_n;
    return p->dtype == DP_F32 ? launch_conv<float, 64>(a, s) : launch_conv<uint16_t, 64>(a, s);
  }
  a.tiles_n = (p->Cout + 127) / 128;
  a.n_tiles = tiles_m * a.tiles_n;
  return p->dtype == DP_F32 ? launch_conv<float, 128>(a, s) : launch_conv<uint16_t, 128>(a, s);
}
