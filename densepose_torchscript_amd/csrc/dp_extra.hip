// "Next" rows of SURVEY §8(f): GPU-side uint8 resize (defaults.py:89) and the visualiser's IUV
// extraction (visualizer.py:10-30). Both are HBM/latency-bound byte work.
#include "dp_common.h"

#pragma clang fp contract(off)

namespace {

// one separable pass of ATen's uint8 bilinear resize: out = clip((W0*x[i0] + W1*x[i1] + 2^(p-1)) >> p)
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, int H, int W, int ow, int src_hwc,
                                const int4* __restrict__ xtab, int prec) {
  const long long total = 3ll * H * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % ow);
    const long long t = i / ow;
    const int y = (int)(t % H);
    const int c = (int)(t / H);
    const int4 e = xtab[xo];
    int a, b;
    if (src_hwc) {
      a = src[((long long)y * W + e.x) * 3 + c];
      b = src[((long long)y * W + e.y) * 3 + c];
    } else {
      a = src[((long long)c * H + y) * W + e.x];
      b = src[((long long)c * H + y) * W + e.y];
    }
    int v = (e.z * a + e.w * b + (1 << (prec - 1))) >> prec;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    tmp[i] = (uint8_t)v;
  }
}
__global__ void resize_v_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst, int H, int oh, int ow,
                                const int4* __restrict__ ytab, int prec) {
  const long long total = 3ll * oh * ow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % ow);
    const long long t = i / ow;
    const int yo = (int)(t % oh);
    const int c = (int)(t / oh);
    const int4 e = ytab[yo];
    const int a = tmp[((long long)c * H + e.x) * ow + xo];
    const int b = tmp[((long long)c * H + e.y) * ow + xo];
    int v = (e.z * a + e.w * b + (1 << (prec - 1))) >> prec;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[i] = (uint8_t)v;
  }
}

__device__ __forceinline__ void src_index(int o, float scale, int n, int& i0, int& i1, float& l) {
  // upsample_bilinear2d with size= given: src = max(scale * (o + 0.5) - 0.5, 0), scale = in / out (float)
  float s = scale * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
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
      return hy * (hx * base[o00] + lx * base[o01]) + ly * (hx * base[o10] + lx * base[o11]);
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

extern "C" int dp_resize_u8_bilinear(const dp_resize_params* p, dp_stream_t stream) {
  DP_REQUIRE(p && p->src && p->tmp && p->dst && p->xtab && p->ytab, "dp_resize_u8_bilinear: null pointer");
  DP_REQUIRE(p->H > 0 && p->W > 0 && p->oh > 0 && p->ow > 0, "dp_resize_u8_bilinear: bad shape");
  DP_REQUIRE(p->xprec > 0 && p->xprec < 23 && p->yprec > 0 && p->yprec < 23, "dp_resize_u8_bilinear: precision");
  hipStream_t s = as_stream(stream);
  long long t1 = 3ll * p->H * p->ow, t2 = 3ll * p->oh * p->ow;
  int g1 = (int)((t1 + 255) / 256), g2 = (int)((t2 + 255) / 256);
  if (g1 > 4096) g1 = 4096;
  if (g2 > 4096) g2 = 4096;
  hipLaunchKernelGGL(resize_h_kernel, dim3(g1), dim3(256), 0, s, p->src, p->tmp, p->H, p->W, p->ow, p->src_hwc,
                     reinterpret_cast<const int4*>(p->xtab), p->xprec);
  hipLaunchKernelGGL(resize_v_kernel, dim3(g2), dim3(256), 0, s, p->tmp, p->dst, p->H, p->oh, p->ow,
                     reinterpret_cast<const int4*>(p->ytab), p->yprec);
  return dp_check_launch("resize kernels");
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
