// "Next" rows of SURVEY §8(f): GPU-side uint8 resize (defaults.py:89) and the visualiser's IUV
// extraction (visualizer.py:10-30). Both are HBM/latency-bound byte work.
#include "dp_common.h"

#pragma clang fp contract(off)

namespace {

// one separable pass of ATen's uint8 bilinear resize: out = clip((W0*x[i0] + W1*x[i1] + 2^(p-1)) >> p)
// grid = (x blocks, 3*H rows, frames): no integer division; frames of one geometry are resized by ONE launch per pass
constexpr int kMaxResizeBatch = 64;
struct ResizeSrcs { const uint8_t* p[kMaxResizeBatch]; };

// dp_preprocess_u8 (paired layout) reading n separate frames instead of one stacked tensor: grid (cell blocks, Hp rows, frames). The same
// expressions as preprocess_paired_kernel (dp_ops.hip): bit-identical output.
template <typename T, bool HWC>
__global__ void preprocess_paired_frames_kernel(const ResizeSrcs srcs, T* __restrict__ dst, int h, int w, int Hp, int Wq, float m0, float m1,
                                                float m2, float s0, float s1, float s2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Wq) return;
  const int y = blockIdx.y, n = blockIdx.z;
  const uint8_t* __restrict__ src = srcs.p[n];
  float4 px[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int x = 2 * j + e - 3;
    px[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < h && x >= 0 && x < w) {
      int v0, v1, v2;
      if constexpr (HWC) {
        const uint8_t* s = src + ((long long)y * w + x) * 3;
        v0 = s[0]; v1 = s[1]; v2 = s[2];
      } else {
        const uint8_t* s = src + (long long)y * w + x;
        v0 = s[0]; v1 = s[(long long)h * w]; v2 = s[2ll * h * w];
      }
      px[e].x = ((float)v0 - m0) / s0;
      px[e].y = ((float)v1 - m1) / s1;
      px[e].z = ((float)v2 - m2) / s2;
    }
  }
  T* d = dst + (((long long)n * Hp + y) * Wq + j) * 8;
  store4(d, px[0]);
  store4(d + 4, px[1]);
}

// each thread produces 4 consecutive output bytes of a row (one 4-byte store when the row pitch allows it)
__device__ __forceinline__ void put4(uint8_t* d, int xo, int ow, const int (&v)[4]) {
  if (xo + 3 < ow && ((reinterpret_cast<uintptr_t>(d + xo) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(d + xo) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (xo + k < ow) d[xo + k] = (uint8_t)v[k];
  }
}

__global__ void resize_h_kernel(const ResizeSrcs srcs, uint8_t* __restrict__ tmp, int H, int W, int ow, int src_hwc,
                                const int4* __restrict__ xtab, int prec) {
  const uint8_t* __restrict__ src = srcs.p[blockIdx.z];
  uint8_t* __restrict__ timg = tmp + (long long)blockIdx.z * 3 * H * ow;
  if (src_hwc) {
    // interleaved source: one output column per thread, the three channels of its two taps are 3 adjacent bytes each
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xo >= ow || y >= H) return;
    const int4 e = xtab[xo];
    const uint8_t* __restrict__ pa = src + ((long long)y * W + e.x) * 3;
    const uint8_t* __restrict__ pb = src + ((long long)y * W + e.y) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r = (e.z * (int)pa[c] + e.w * (int)pb[c] + (1 << (prec - 1))) >> prec;
      timg[((long long)c * H + y) * ow + xo] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    return;
  }
  const int xo = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (xo >= ow) return;
  const int row = blockIdx.y;          // c * H + y
  const uint8_t* __restrict__ line = src + (long long)row * W;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int4 e = xtab[xo + k < ow ? xo + k : ow - 1];
    const int r = (e.z * (int)line[e.x] + e.w * (int)line[e.y] + (1 << (prec - 1))) >> prec;
    v[k] = r < 0 ? 0 : (r > 255 ? 255 : r);
  }
  put4(timg + (long long)row * ow, xo, ow, v);
}
__global__ void resize_v_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst, int H, int oh, int ow,
                                const int4* __restrict__ ytab, int prec) {
  const int xo = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (xo >= ow) return;
  const int row = blockIdx.y;          // c * oh + yo
  const int c = row / oh, yo = row - c * oh;
  const int4 e = ytab[yo];
  const uint8_t* __restrict__ t = tmp + ((long long)blockIdx.z * 3 + c) * H * ow;
  const uint8_t* __restrict__ la = t + (long long)e.x * ow;
  const uint8_t* __restrict__ lb = t + (long long)e.y * ow;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = xo + k < ow ? xo + k : ow - 1;
    const int r = (e.z * (int)la[x] + e.w * (int)lb[x] + (1 << (prec - 1))) >> prec;
    v[k] = r < 0 ? 0 : (r > 255 ? 255 : r);
  }
  put4(dst + ((long long)blockIdx.z * 3 * oh + row) * ow, xo, ow, v);
}

// The vertical pass FUSED with rcnn.py:156-181 (normalise, zero-pad to a multiple of 32) and the paired-pixel layout the stem reads
// (dp_ops.hip preprocess_paired_kernel): cell (n, y, j) of [n][Hp][Wq][8] = the 4-channel pixels 2j - 3 and 2j - 2 of row y. Same integer
// expression as resize_v_kernel, same float expression as the preprocess kernel: bit-identical to the two launches it replaces, and the
// resized uint8 batch [n][3][oh][ow] is neither written nor read back (frames whose scale is not 1, i.e. every video).
template <typename T>
__global__ void resize_v_preprocess_paired_kernel(const uint8_t* __restrict__ tmp, T* __restrict__ dst, int H, int oh, int ow, int Hp, int Wq,
                                                  const int4* __restrict__ ytab, int prec, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
  if (j >= Wq) return;
  float4 px[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (y < oh) {
    const int4 e = ytab[y];
    const uint8_t* __restrict__ t = tmp + (long long)n * 3 * H * ow;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int x = 2 * j + k - 3;
      if (x >= 0 && x < ow) {
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint8_t* __restrict__ pl = t + (long long)c * H * ow;
          const int r = (e.z * (int)pl[(long long)e.x * ow + x] + e.w * (int)pl[(long long)e.y * ow + x] + (1 << (prec - 1))) >> prec;
          v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
        px[k].x = ((float)v[0] - m0) / s0;
        px[k].y = ((float)v[1] - m1) / s1;
        px[k].z = ((float)v[2] - m2) / s2;
      }
    }
  }
  T* d = dst + (((long long)n * Hp + y) * Wq + j) * 8;
  store4(d, px[0]);
  store4(d + 4, px[1]);
}

// Bit-exact restatement of ATen's CPU upsample_bilinear2d (size= given, align_corners=False) as the reference's visualiser
// calls it (visualizer.py:14-16,24-25): the vectorised CPU kernel evaluates
//     src = fma(scale, dst + 0.5, -0.5) clamped at 0, scale = in / out in float;  i0 = min(int(src), n - 1);  l = src - i0
//     value = fma(wy0, fma(wx0, a, wx1 * b), wy1 * fma(wx0, c, wx1 * d)),  w0 = 1 - l, w1 = l
// (the products wx1 * b, wx1 * d and wy1 * t1 are rounded on their own). Checked bit for bit against torch's CPU kernel
// in tests/test_oracle_ops.py::test_bilinear_restatement_is_bit_exact; contraction is switched off so that hipcc emits
// exactly these operations and no others. (torch's kernel changes its own contraction pattern with the tensor shape -
// outputs narrower than 63 columns, some channel counts - so on such shapes the resampled values can differ by one ulp;
// the part index, an argmax over them, is compared bit-exactly in the tests and does not move.)
__device__ __forceinline__ void src_index(int o, float scale, int n, int& i0, int& i1, float& l) {
  float s = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
  l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
}

__global__ void iuv_extract_kernel(const dp_iuv_extract_params p) {
  const int r = blockIdx.y;
  const int w = p.box_xywh[r * 4 + 2], h = p.box_xywh[r * 4 + 3];
  const int S = p.S;
  const float sy = (float)S / (float)h, sx = (float)S / (float)w;
  const long long off = p.out_offset[r];
  const long long plane = (long long)S * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
    const int y = i / w, x = i - y * w;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(y, sy, S, y0, y1, ly);
    src_index(x, sx, S, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long long o00 = (long long)y0 * S + x0, o01 = (long long)y0 * S + x1, o10 = (long long)y1 * S + x0, o11 = (long long)y1 * S + x1;
    auto sample = [&](const float* base) {
      const float t0 = __builtin_fmaf(hx, base[o00], lx * base[o01]);
      const float t1 = __builtin_fmaf(hx, base[o10], lx * base[o11]);
      return __builtin_fmaf(hy, t0, ly * t1);
    };
    // coarse argmax > 0 ?
    int cbest = 0;
    float cval = sample(p.coarse + (long long)r * p.n_coarse * plane);
    for (int c = 1; c < p.n_coarse; ++c) {
      const float v = sample(p.coarse + ((long long)r * p.n_coarse + c) * plane);
      if (v > cval) { cval = v; cbest = c; }
    }
    int fbest = 0;
    float fval = sample(p.fine + (long long)r * p.n_fine * plane);
    for (int c = 1; c < p.n_fine; ++c) {
      const float v = sample(p.fine + ((long long)r * p.n_fine + c) * plane);
      if (v > fval) { fval = v; fbest = c; }
    }
    const int label = cbest > 0 ? fbest : 0;
    float uu = 0.f, vv = 0.f;
    if (label > 0) {
      uu = sample(p.u + ((long long)r * p.n_fine + label) * plane);
      vv = sample(p.v + ((long long)r * p.n_fine + label) * plane);
    }
    p.labels[off + i] = (uint8_t)label;
    p.uv[2 * off + i] = uu;
    p.uv[2 * off + (long long)h * w + i] = vv;
  }
}

}  // namespace

static int resize_launch(const dp_resize_params* p, const void* const* srcs, int n, hipStream_t s) {
  DP_REQUIRE(3ll * p->H < 65536 && 3ll * p->oh < 65536, "dp_resize_u8_bilinear: more than 21845 rows");
  for (int first = 0; first < n; first += kMaxResizeBatch) {
    const int cnt = n - first < kMaxResizeBatch ? n - first : kMaxResizeBatch;
    ResizeSrcs a;
    for (int i = 0; i < cnt; ++i) {
      DP_REQUIRE(srcs[first + i], "dp_resize_u8_bilinear: null frame %d", first + i);
      a.p[i] = static_cast<const uint8_t*>(srcs[first + i]);
    }
    uint8_t* tmp = p->tmp + (long long)first * 3 * p->H * p->ow;
    uint8_t* dst = p->dst + (long long)first * 3 * p->oh * p->ow;
    const int gx = (p->ow + 4 * 256 - 1) / (4 * 256);   // 4 output bytes per thread
    const dim3 gh = p->src_hwc ? dim3((p->ow + 255) / 256, p->H, cnt) : dim3(gx, 3 * p->H, cnt);
    hipLaunchKernelGGL(resize_h_kernel, gh, dim3(256), 0, s, a, tmp, p->H, p->W, p->ow, p->src_hwc,
                       reinterpret_cast<const int4*>(p->xtab), p->xprec);
    hipLaunchKernelGGL(resize_v_kernel, dim3(gx, 3 * p->oh, cnt), dim3(256), 0, s, tmp, dst, p->H, p->oh, p->ow,
                       reinterpret_cast<const int4*>(p->ytab), p->yprec);
  }
  return dp_check_launch("resize kernels");
}

static int resize_check(const dp_resize_params* p) {
  DP_REQUIRE(p && p->tmp && p->dst && p->xtab && p->ytab, "dp_resize_u8_bilinear: null pointer");
  DP_REQUIRE(p->H > 0 && p->W > 0 && p->oh > 0 && p->ow > 0, "dp_resize_u8_bilinear: bad shape");
  DP_REQUIRE(p->xprec > 0 && p->xprec < 23 && p->yprec > 0 && p->yprec < 23, "dp_resize_u8_bilinear: precision");
  return DP_OK;
}

extern "C" int dp_resize_u8_bilinear(const dp_resize_params* p, dp_stream_t stream) {
  if (int rc = resize_check(p)) return rc;
  const void* src = p->src;
  return resize_launch(p, &src, 1, as_stream(stream));
}

extern "C" int dp_resize_u8_bilinear_batch(const dp_resize_params* p, const void* const* srcs, int n, dp_stream_t stream) {
  if (int rc = resize_check(p)) return rc;
  DP_REQUIRE(srcs && n >= 0, "dp_resize_u8_bilinear_batch: bad frame list");
  if (n == 0) return DP_OK;
  return resize_launch(p, srcs, n, as_stream(stream));
}

extern "C" int dp_resize_preprocess_u8_batch(const dp_resize_params* p, const void* const* srcs, int n, const dp_preprocess_params* q,
                                             dp_stream_t stream) {
  DP_REQUIRE(p && p->tmp && p->xtab && p->ytab && q && q->dst, "dp_resize_preprocess_u8_batch: null pointer");
  DP_REQUIRE(p->H > 0 && p->W > 0 && p->oh > 0 && p->ow > 0 && p->xprec > 0 && p->xprec < 23 && p->yprec > 0 && p->yprec < 23,
             "dp_resize_preprocess_u8_batch: bad resize parameters");
  DP_REQUIRE(srcs && n > 0 && n <= kMaxResizeBatch, "dp_resize_preprocess_u8_batch: 1 .. %d frames per call", kMaxResizeBatch);
  DP_REQUIRE(q->paired == 1 && q->n_img == n && q->h == p->oh && q->w == p->ow && q->Hp >= p->oh && q->Wp >= p->ow && q->Wp % 2 == 0 && q->Hp < 65536,
             "dp_resize_preprocess_u8_batch: the preprocess parameters must describe the resized frames in the paired layout");
  DP_REQUIRE(q->dtype == DP_F32 || q->dtype == DP_BF16 || q->dtype == DP_F16, "dp_resize_preprocess_u8_batch: bad dtype");
  DP_REQUIRE(3ll * p->H < 65536, "dp_resize_preprocess_u8_batch: more than 21845 rows");
  hipStream_t s = as_stream(stream);
  ResizeSrcs a;
  for (int i = 0; i < n; ++i) {
    DP_REQUIRE(srcs[i], "dp_resize_preprocess_u8_batch: null frame %d", i);
    a.p[i] = static_cast<const uint8_t*>(srcs[i]);
  }
  const int gx = (p->ow + 4 * 256 - 1) / (4 * 256);
  const dim3 gh = p->src_hwc ? dim3((p->ow + 255) / 256, p->H, n) : dim3(gx, 3 * p->H, n);
  hipLaunchKernelGGL(resize_h_kernel, gh, dim3(256), 0, s, a, p->tmp, p->H, p->W, p->ow, p->src_hwc, reinterpret_cast<const int4*>(p->xtab), p->xprec);
  const int Wq = q->Wp / 2 + 3;
  const dim3 gv((Wq + 255) / 256, q->Hp, n);
  const int4* yt = reinterpret_cast<const int4*>(p->ytab);
  switch (q->dtype) {
    case DP_F32:
      hipLaunchKernelGGL(resize_v_preprocess_paired_kernel<float>, gv, dim3(256), 0, s, p->tmp, (float*)q->dst, p->H, p->oh, p->ow, q->Hp, Wq, yt, p->yprec,
                         q->mean[0], q->mean[1], q->mean[2], q->std[0], q->std[1], q->std[2]);
      break;
    case DP_BF16:
      hipLaunchKernelGGL(resize_v_preprocess_paired_kernel<uint16_t>, gv, dim3(256), 0, s, p->tmp, (uint16_t*)q->dst, p->H, p->oh, p->ow, q->Hp, Wq, yt,
                         p->yprec, q->mean[0], q->mean[1], q->mean[2], q->std[0], q->std[1], q->std[2]);
      break;
    default:
      hipLaunchKernelGGL(resize_v_preprocess_paired_kernel<f16_t>, gv, dim3(256), 0, s, p->tmp, (f16_t*)q->dst, p->H, p->oh, p->ow, q->Hp, Wq, yt, p->yprec,
                         q->mean[0], q->mean[1], q->mean[2], q->std[0], q->std[1], q->std[2]);
  }
  return dp_check_launch("resize_v_preprocess_paired_kernel");
}

extern "C" int dp_preprocess_u8_frames(const dp_preprocess_params* q, const void* const* srcs, int n, dp_stream_t stream) {
  DP_REQUIRE(q && q->dst && srcs, "dp_preprocess_u8_frames: null pointer");
  DP_REQUIRE(n > 0 && n <= kMaxResizeBatch, "dp_preprocess_u8_frames: 1 .. %d frames per call", kMaxResizeBatch);
  DP_REQUIRE(q->paired == 1 && q->n_img == n && q->h > 0 && q->w > 0 && q->Hp >= q->h && q->Wp >= q->w && q->Wp % 2 == 0 && q->Hp < 65536,
             "dp_preprocess_u8_frames: the parameters must describe n frames in the paired layout");
  DP_REQUIRE(q->dtype == DP_F32 || q->dtype == DP_BF16 || q->dtype == DP_F16, "dp_preprocess_u8_frames: bad dtype");
  hipStream_t s = as_stream(stream);
  ResizeSrcs a;
  for (int i = 0; i < n; ++i) {
    DP_REQUIRE(srcs[i], "dp_preprocess_u8_frames: null frame %d", i);
    a.p[i] = static_cast<const uint8_t*>(srcs[i]);
  }
  const int Wq = q->Wp / 2 + 3;
  const dim3 g((Wq + 255) / 256, q->Hp, n);
#define DP_PPF(T, HWC)                                                                                                                        \
  hipLaunchKernelGGL((preprocess_paired_frames_kernel<T, HWC>), g, dim3(256), 0, s, a, (T*)q->dst, q->h, q->w, q->Hp, Wq, q->mean[0], q->mean[1], \
                     q->mean[2], q->std[0], q->std[1], q->std[2])
  if (q->src_hwc) {
    if (q->dtype == DP_F32) DP_PPF(float, true); else if (q->dtype == DP_BF16) DP_PPF(uint16_t, true); else DP_PPF(f16_t, true);
  } else {
    if (q->dtype == DP_F32) DP_PPF(float, false); else if (q->dtype == DP_BF16) DP_PPF(uint16_t, false); else DP_PPF(f16_t, false);
  }
#undef DP_PPF
  return dp_check_launch("preprocess_paired_frames_kernel");
}

extern "C" int dp_iuv_extract(const dp_iuv_extract_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_iuv_extract: null params");
  if (p->R == 0) return DP_OK;
  DP_REQUIRE(p->coarse && p->fine && p->u && p->v && p->box_xywh && p->out_offset && p->labels && p->uv, "dp_iuv_extract: null pointer");
  DP_REQUIRE(p->R > 0 && p->S > 0 && p->n_coarse > 0 && p->n_fine > 0 && p->max_hw > 0, "dp_iuv_extract: bad shape");
  int gx = (p->max_hw + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(iuv_extract_kernel, dim3(gx, p->R), dim3(256), 0, as_stream(stream), *p);
  return dp_check_launch("iuv_extract_kernel");
}
