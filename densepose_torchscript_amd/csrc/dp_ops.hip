// Memory-bound NHWC helper kernels of the DensePose path (HBM-bound: coalesced 8/16-byte accesses,
// grid-stride, no LDS needed except the GroupNorm reduction).
#include "dp_common.h"
#include <stdlib.h>

#pragma clang fp contract(off)  // keep mul/add un-fused: these ops are compared with the CPU oracle

thread_local char dp_err_buf[512] = {0};

extern "C" int dp_abi_version(void) { return DP_ABI_VERSION; }
extern "C" const char* dp_last_error(void) { return dp_err_buf; }

namespace {

constexpr int kBlock = 256;
inline int grid_for(long long work) {
  long long g = (work + kBlock - 1) / kBlock;
  if (g > 256 * 16) g = 256 * 16;  // ~16 workgroups per CU, grid-stride beyond that
  if (g < 1) g = 1;
  return (int)g;
}

// ---- K2 preprocess: planar uint8 [n][3][h][w] -> NHWC8 (x - mean)/std, zero padded ------------------
template <typename T>
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst, int n_img, int h, int w, int Hp,
                                  int Wp, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long total = (long long)n_img * Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long long t = i / Wp;
    const int y = (int)(t % Hp);
    const int n = (int)(t / Hp);
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < h && x < w) {
      const uint8_t* s = src + ((long long)n * 3 * h + y) * w + x;
      lo.x = ((float)s[0] - m0) / s0;
      lo.y = ((float)s[(long long)h * w] - m1) / s1;
      lo.z = ((float)s[2ll * h * w] - m2) / s2;
    }
    T* d = dst + i * 8;
    store4(d, lo);
    store4(d + 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// paired layout (dp_preprocess_params.paired): cell j of a row holds the 4-channel pixels 2j - 3 and 2j - 2.
// HWC = the source frames are interleaved [n][h][w][3] (a frame as the caller hands it over, defaults.py:76-78) instead of the
// planar [n][3][h][w] output of the resize: at scale 1 the resize is the identity and the frames are read directly.
template <typename T, bool HWC>
__global__ void preprocess_paired_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst, int n_img, int h, int w, int Hp,
                                         int Wq, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long total = (long long)n_img * Hp * Wq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Wq);
    const long long t = i / Wq;
    const int y = (int)(t % Hp);
    const int n = (int)(t / Hp);
    float4 px[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int x = 2 * j + e - 3;
      px[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < h && x >= 0 && x < w) {
        int v0, v1, v2;
        if constexpr (HWC) {
          const uint8_t* s = src + (((long long)n * h + y) * w + x) * 3;
          v0 = s[0]; v1 = s[1]; v2 = s[2];
        } else {
          const uint8_t* s = src + ((long long)n * 3 * h + y) * w + x;
          v0 = s[0]; v1 = s[(long long)h * w]; v2 = s[2ll * h * w];
        }
        px[e].x = ((float)v0 - m0) / s0;
        px[e].y = ((float)v1 - m1) / s1;
        px[e].z = ((float)v2 - m2) / s2;
      }
    }
    T* d = dst + i * 8;
    store4(d, px[0]);
    store4(d + 4, px[1]);
  }
}

// ---- 3x3 stride-2 pad-1 max pool -------------------------------------------------------------------
// one workgroup per output row: 32-bit index math, 16-byte (8-channel) accesses
template <typename T>
__global__ __launch_bounds__(kBlock) void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C, int Ho,
                                                              int Wo) {
  const int C8 = C >> 3;
  const int row = blockIdx.x;  // n * Ho + ho
  const int n = row / Ho, ho = row - n * Ho;
  const T* img = in + (long long)n * H * W * C;
  T* orow = out + (long long)row * Wo * C;
  const int items = Wo * C8;
  for (int i = threadIdx.x; i < items; i += kBlock) {
    const int wo = i / C8, c = (i - wo * C8) * 8;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = ho * 2 - 1 + dy;
      if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = wo * 2 - 1 + dx;
        if ((unsigned)x >= (unsigned)W) continue;
        float v[8];
        load8(img + ((long long)y * W + x) * C + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
      }
    }
    store8(orow + wo * C + c, m);
  }
}

template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    store4(out + i * 4, load4(in + (((long long)n * H + 2 * ho) * W + 2 * wo) * C + c4 * 4));
  }
}

// ---- bilinear x2, align_corners=False: out[2i] = .75 in[i] + .25 in[i-1], out[2i+1] = .75 in[i] + .25 in[i+1]
//      computed exactly as ATen's upsample_bilinear2d: separable weights (1-l, l) with l in {0.25, 0.75, 0}
__device__ __forceinline__ void bil_src(int o, int n, int& i0, int& i1, float& l) {
  // src = max((o + 0.5) * 0.5 - 0.5, 0)
  float s = ((float)o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = s - (float)i0;
}
__device__ __forceinline__ float lerp2d1(float a, float b, float c, float d, float lx, float ly) {
  // ATen order: h0lambda * (w0lambda * p00 + w1lambda * p01) + h1lambda * (w0lambda * p10 + w1lambda * p11)
  const float hx = 1.f - lx, hy = 1.f - ly;
  return hy * (hx * a + lx * b) + ly * (hx * c + lx * d);
}

// one workgroup per output row (n, yo): the vertical taps are uniform, 32-bit index math, 16-byte accesses
template <typename T>
__global__ __launch_bounds__(kBlock) void upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int C,
                                                            int accumulate) {
  const int C8 = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const int row = blockIdx.x;
  const int n = row / Ho, yo = row - n * Ho;
  int y0, y1;
  float ly;
  bil_src(yo, H, y0, y1, ly);
  const T* r0 = in + ((long long)n * H + y0) * W * C;
  const T* r1 = in + ((long long)n * H + y1) * W * C;
  T* orow = out + (long long)row * Wo * C;
  const int items = Wo * C8;
  for (int i = threadIdx.x; i < items; i += kBlock) {
    const int xo = i / C8, c = (i - xo * C8) * 8;
    int x0, x1;
    float lx;
    bil_src(xo, W, x0, x1, lx);
    float a[8], b[8], cc[8], d[8], r[8];
    load8(r0 + x0 * C + c, a);
    load8(r0 + x1 * C + c, b);
    load8(r1 + x0 * C + c, cc);
    load8(r1 + x1 * C + c, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = lerp2d1(a[k], b[k], cc[k], d[k], lx, ly);
    if (accumulate) {
      float o[8];
      load8(orow + xo * C + c, o);
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = o[k] + r[k];
    }
    store8(orow + xo * C + c, r);
  }
}


// ---- K18: bilinear x2 + channel split + NHWC -> NCHW (fp32) --------------------------------------------
// One workgroup per (roi, source row pair i, i+1): the two low-res rows are brought into LDS with coalesced
// channel-contiguous loads, then the (up to) two output rows 2i+1, 2i+2 that interpolate between them are
// produced with lanes along x, so every NCHW store instruction writes 256 contiguous bytes.
constexpr int kIuvMaxRowFloats = 56 * 96;  // Ws * in_c of the largest supported map (S1x: 56 x 80, legacy 28 x 96)

__global__ __launch_bounds__(256) void iuv_upsample_split_kernel(const float* __restrict__ in, int R, int Hs, int Ws, int in_c, int n_coarse,
                                                                 int n_fine, float* __restrict__ coarse, float* __restrict__ fine,
                                                                 float* __restrict__ u, float* __restrict__ v, int xslots,
                                                                 const int* __restrict__ r_dev) {
  // LDS image: [2 rows][in_c][Ws + 1] - channel-major, so that the lanes of the interpolation loop (consecutive x of one
  // channel) read consecutive words (pixel-major staging put them in_c words apart: in_c = 80 -> 8-way bank conflicts)
  extern __shared__ __attribute__((aligned(16))) float rows[];
  const int r = blockIdx.y;
  if (r_dev != nullptr && r >= *r_dev) return;  // a box slot behind the live ones (dp_iuv_params.r_dev)
  const int i = (int)blockIdx.x - 1;            // source row pair (i, i+1), i in [-1, Hs-1]
  const int y0 = i < 0 ? 0 : i;
  const int y1 = i + 1 > Hs - 1 ? Hs - 1 : i + 1;
  const int rowf = Ws * in_c, pitch = Ws + 1, plane = in_c * pitch;
  const float* src = in + (long long)r * Hs * rowf;
  for (int t = threadIdx.x * 4; t < rowf; t += 256 * 4) {
    const float4 a = *reinterpret_cast<const float4*>(src + (long long)y0 * rowf + t);
    const float4 b = *reinterpret_cast<const float4*>(src + (long long)y1 * rowf + t);
    const int x = t / in_c, c = t - x * in_c;   // in_c % 4 == 0: the four channels belong to one pixel
    float* d = rows + c * pitch + x;
    d[0] = a.x; d[pitch] = a.y; d[2 * pitch] = a.z; d[3 * pitch] = a.w;
    d += plane;
    d[0] = b.x; d[pitch] = b.y; d[2 * pitch] = b.z; d[3 * pitch] = b.w;
  }
  __syncthreads();
  const int Ho = 2 * Hs, Wo = 2 * Ws, Ctot = n_coarse + 3 * n_fine;
  const long long hw = (long long)Ho * Wo;
  if (xslots < 0) {
    // FOUR consecutive outputs of a row per thread (Wo % 4 == 0: one 16-byte store per thread, a channel's row segment of a wave is
    // contiguous): outputs 4q .. 4q + 3 interpolate between the source columns 2q - 1 .. 2q + 2 - eight LDS reads instead of sixteen.
    // Every output is still computed by the expression of the one-output form below, on the same four samples: bit-identical.
    const int qslots = -xslots;                      // power of two >= Wo / 4
    const int q = threadIdx.x & (qslots - 1), cl = threadIdx.x / qslots, n_cl = 256 / qslots;
    if (4 * q >= Wo) return;
    int xa[4], xb[4];
    float lx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bil_src(4 * q + e, Ws, xa[e], xb[e], lx[e]);
    for (int k = 0; k < 2; ++k) {
      const int yo = 2 * i + 1 + k;
      if (yo < 0 || yo >= Ho) continue;
      int yy0, yy1;
      float ly;
      bil_src(yo, Hs, yy0, yy1, ly);
      const float hy = 1.f - ly;
      const long long pix = (long long)yo * Wo + 4 * q;
      for (int c = cl; c < Ctot; c += n_cl) {
        const float* ra = rows + c * pitch;
        // source columns xa[0] = max(2q - 1, 0) .. xb[3] = min(2q + 2, Ws - 1): read once, picked per output
        const int c0 = xa[0];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = min(c0 + j, Ws - 1);
          t0[j] = ra[col];
          t1[j] = ra[plane + col];
        }
        float4 val;
        float* vp = &val.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ia = xa[e] - c0, ib = xb[e] - c0;       // 0 .. 3
          const float a = ia == 0 ? t0[0] : ia == 1 ? t0[1] : ia == 2 ? t0[2] : t0[3];
          const float b = ib == 0 ? t0[0] : ib == 1 ? t0[1] : ib == 2 ? t0[2] : t0[3];
          const float cc = ia == 0 ? t1[0] : ia == 1 ? t1[1] : ia == 2 ? t1[2] : t1[3];
          const float d = ib == 0 ? t1[0] : ib == 1 ? t1[1] : ib == 2 ? t1[2] : t1[3];
          const float hx = 1.f - lx[e];
          vp[e] = hy * (hx * a + lx[e] * b) + ly * (hx * cc + lx[e] * d);
        }
        float* dst;
        if (c < n_coarse) dst = coarse + ((long long)r * n_coarse + c) * hw + pix;
        else if (c < n_coarse + n_fine) dst = fine + ((long long)r * n_fine + (c - n_coarse)) * hw + pix;
        else if (c < n_coarse + 2 * n_fine) dst = u + ((long long)r * n_fine + (c - n_coarse - n_fine)) * hw + pix;
        else dst = v + ((long long)r * n_fine + (c - n_coarse - 2 * n_fine)) * hw + pix;
        *reinterpret_cast<float4*>(dst) = val;
      }
    }
    return;
  }
  // thread = (x slot, channel lane): the horizontal taps are computed once per thread, the channel is wave-uniform (the
  // x slots of a channel lane span whole waves), so the loop body is 4 LDS reads + the lerp + one coalesced store
  const int xo = threadIdx.x & (xslots - 1), cl = threadIdx.x / xslots, n_cl = 256 / xslots;
  if (xo >= Wo) return;
  int x0, x1;
  float lx;
  bil_src(xo, Ws, x0, x1, lx);
  const float hx = 1.f - lx;
  // output rows whose bilinear source rows are exactly (y0, y1): yo = 2i+1 (ly = .25) and 2i+2 (ly = .75);
  // the clamped border rows: yo = 0 <- pair i = -1 (y0 = y1 = 0, ly = 0), yo = Ho-1 <- pair i = Hs-1 (y0 = y1, ly irrelevant)
  for (int k = 0; k < 2; ++k) {
    const int yo = 2 * i + 1 + k;
    if (yo < 0 || yo >= Ho) continue;
    int yy0, yy1;
    float ly;
    bil_src(yo, Hs, yy0, yy1, ly);   // rows[] holds (y0, y1) = (yy0, yy1) by construction
    const float hy = 1.f - ly;
    const long long pix = (long long)yo * Wo + xo;
    for (int c = cl; c < Ctot; c += n_cl) {
      const float* ra = rows + c * pitch;
      const float a = ra[x0], b = ra[x1], cc = ra[plane + x0], d = ra[plane + x1];
      const float val = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
      if (c < n_coarse) coarse[((long long)r * n_coarse + c) * hw + pix] = val;
      else if (c < n_coarse + n_fine) fine[((long long)r * n_fine + (c - n_coarse)) * hw + pix] = val;
      else if (c < n_coarse + 2 * n_fine) u[((long long)r * n_fine + (c - n_coarse - n_fine)) * hw + pix] = val;
      else v[((long long)r * n_fine + (c - n_coarse - 2 * n_fine)) * hw + pix] = val;
    }
  }
}

// ---- decoder level merge: out = base + sum_k bilinear_x2(ups[k]) in ONE pass (roi_head.py:71-79) -------------
template <typename T>
__global__ __launch_bounds__(kBlock) void merge_up2x_kernel(const T* __restrict__ base, const T* __restrict__ u0, const T* __restrict__ u1,
                                                            const T* __restrict__ u2, int n_ups, T* __restrict__ out, int N, int H, int W,
                                                            int C) {
  // H, W are the LOW-res dims of the ups; base/out are [N, 2H, 2W, C].
  // One workgroup per output ROW PAIR (2p-1, 2p), p = 0..H, one thread per column pair (2q-1, 2q), q = 0..W, and 8
  // channels: the four outputs of such a 2x2 block interpolate between the SAME four low-res samples (rows p-1, p and
  // columns q-1, q), so each up map costs 4 loads per 4 outputs instead of 16. Taps and weights still come from bil_src
  // per output row / column, so every output is computed by exactly the expression of the unblocked form.
  const int C8 = C >> 3, Ho = 2 * H, Wo = 2 * W;
  const int n = blockIdx.x / (H + 1), p = blockIdx.x - n * (H + 1);
  const int ya = 2 * p - 1, yb = 2 * p;          // output rows of this pair (ya < 0 / yb >= Ho: absent)
  const bool has_a = ya >= 0, has_b = yb < Ho;
  int y0, y1, t0, t1;
  float lya = 0.f, lyb = 0.f;
  if (has_a) bil_src(ya, H, y0, y1, lya);
  if (has_b) bil_src(yb, H, has_a ? t0 : y0, has_a ? t1 : y1, lyb);   // same taps as row ya when both exist
  const long long o0 = ((long long)n * H + y0) * W * C, o1 = ((long long)n * H + y1) * W * C;
  const long long ra = ((long long)n * Ho + ya) * Wo * C, rb = ((long long)n * Ho + yb) * Wo * C;
  const T* ups[3] = {u0, u1, u2};
  const int items = (W + 1) * C8;
  for (int i = threadIdx.x; i < items; i += kBlock) {
    const int q = i / C8, c = (i - q * C8) * 8;
    const int xa = 2 * q - 1, xb = 2 * q;
    const bool has_xa = xa >= 0, has_xb = xb < Wo;
    int x0, x1, s0, s1;
    float lxa = 0.f, lxb = 0.f;
    if (has_xa) bil_src(xa, W, x0, x1, lxa);
    if (has_xb) bil_src(xb, W, has_xa ? s0 : x0, has_xa ? s1 : x1, lxb);
    float acc[2][2][8] = {};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool ok = (r ? has_b : has_a) && (k ? has_xb : has_xa);
        if (ok) load8(base + (r ? rb : ra) + (long long)(k ? xb : xa) * C + c, acc[r][k]);
      }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (m < n_ups) {
        float a[8], b[8], cc[8], d[8];
        load8(ups[m] + o0 + x0 * C + c, a);
        load8(ups[m] + o0 + x1 * C + c, b);
        load8(ups[m] + o1 + x0 * C + c, cc);
        load8(ups[m] + o1 + x1 * C + c, d);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float ly = r ? lyb : lya, lx = k ? lxb : lxa;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][k][e] = acc[r][k][e] + lerp2d1(a[e], b[e], cc[e], d[e], lx, ly);
          }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool ok = (r ? has_b : has_a) && (k ? has_xb : has_xa);
        if (ok) store8(out + (r ? rb : ra) + (long long)(k ? xb : xa) * C + c, acc[r][k]);
      }
  }
}

// ---- K17: GroupNorm(+ReLU) per ROI, NHWC slice; one workgroup per (roi, group) ------------------------------
template <typename T>
__global__ void groupnorm_kernel(T* __restrict__ x, int HW, int C, int c_stride, int c_off, int groups, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, int relu, const int* __restrict__ r_dev) {
  const int r = blockIdx.x / groups, g = blockIdx.x % groups;
  if (r_dev != nullptr && r >= *r_dev) return;
  const int cg = C / groups;
  T* base = x + (long long)r * HW * c_stride + c_off + g * cg;
  const int n = HW * cg;
  // two-pass (mean, then centred variance) in fp32 with a fixed reduction tree -> deterministic
  __shared__ float red[kBlock];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) s += Elem<T>::load(base + (long long)(i / cg) * c_stride + (i % cg));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = red[0] / (float)n;
  __syncthreads();
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const float d = Elem<T>::load(base + (long long)(i / cg) * c_stride + (i % cg)) - mean;
    q += d * d;
  }
  red[threadIdx.x] = q;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float rstd = 1.0f / sqrtf(red[0] / (float)n + eps);
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const int c = i % cg;
    T* p = base + (long long)(i / cg) * c_stride + c;
    float v = (Elem<T>::load(p) - mean) * rstd * gamma[g * cg + c] + beta[g * cg + c];
    if (relu) v = fmaxf(v, 0.f);
    Elem<T>::store(p, v);
  }
}

// The same normalisation with the group held in REGISTERS: one 16-byte read and one 16-byte write per 8 (bf16 / fp16) or 4 (fp32)
// elements instead of three element-wise sweeps over global memory (a 28 x 28 x 16-channel group of the DeepLab head is 25 KB).
// Two-pass statistics (mean, then centred variance) in fp32 with fixed per-thread order + reduction tree: deterministic.
// Legal when a 16-byte vector does not straddle groups (cg * sizeof(T) % 16 == 0) and the group fits VPT vectors per thread.
template <typename T, int VPT>
__global__ __launch_bounds__(kBlock) void groupnorm_reg_kernel(T* __restrict__ x, int HW, int C, int c_stride, int c_off, int groups,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              int relu, const int* __restrict__ r_dev) {
  constexpr int VE = 16 / sizeof(T);            // elements per vector
  const int r = blockIdx.x / groups, g = blockIdx.x % groups;
  if (r_dev != nullptr && r >= *r_dev) return;
  const int cg = C / groups, vpp = cg / VE;     // vectors per pixel of this group
  T* base = x + (long long)r * HW * c_stride + c_off + g * cg;
  const int nv = HW * vpp, n = HW * cg;
  __shared__ float red[kBlock];
  float v[VPT][VE];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * kBlock;
    if (i < nv) {
      const int px = i / vpp, part = i - px * vpp;
      float t[8];
      if constexpr (VE == 8) load8(base + (long long)px * c_stride + part * VE, t);
      else { const float4 f = load4(base + (long long)px * c_stride + part * VE); t[0] = f.x; t[1] = f.y; t[2] = f.z; t[3] = f.w; }
#pragma unroll
      for (int e = 0; e < VE; ++e) { v[k][e] = t[e]; s += t[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[k][e] = 0.f;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = red[0] / (float)n;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if ((int)threadIdx.x + k * kBlock < nv) {
#pragma unroll
      for (int e = 0; e < VE; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  red[threadIdx.x] = q;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float rstd = 1.0f / sqrtf(red[0] / (float)n + eps);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * kBlock;
    if (i < nv) {
      const int px = i / vpp, part = i - px * vpp;
      float o[8];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int c = g * cg + part * VE + e;
        float w = (v[k][e] - mean) * rstd * gamma[c] + beta[c];
        o[e] = relu ? fmaxf(w, 0.f) : w;
      }
      T* dst = base + (long long)px * c_stride + part * VE;
      if constexpr (VE == 8) store8(dst, o);
      else store4(dst, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

// ... and with whole 128-byte lines: one 1024-thread workgroup takes the 128 B / (cg * sizeof(T)) adjacent groups that share a cache
// line of every pixel (4 groups of 16 bf16 channels, 8 groups of 8), eight 16-byte vectors per pixel. A thread's vectors all belong
// to one group (1024 % 8 == 0), so the per-group statistics are a strided tree reduction over threads with equal (thread & 7).
// One group per workgroup reads 32-byte pieces of 128-byte lines: 2 TB/s on the DeepLab head; this form moves the same bytes in full lines.
constexpr int kGnLineThreads = 1024, kGnLineVpt = 7;
template <typename T>
__global__ __launch_bounds__(kGnLineThreads) void groupnorm_line_kernel(T* __restrict__ x, int HW, int C, int c_stride, int c_off, int groups,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        float eps, int relu, const int* __restrict__ r_dev) {
  constexpr int VE = 16 / sizeof(T);
  const int cg = C / groups, vpp = cg / VE;      // vectors per pixel and group: 1, 2, 4 or 8
  const int gw = 8 / vpp;                        // groups of this workgroup (one 128-byte line of every pixel)
  const int wpr = groups / gw;                   // workgroups per roi
  const int r = blockIdx.x / wpr, g0 = (blockIdx.x - r * wpr) * gw;
  if (r_dev != nullptr && r >= *r_dev) return;
  T* base = x + (long long)r * HW * c_stride + c_off + g0 * cg;
  const int part = threadIdx.x & 7;              // 16-byte vector of the line: fixed per thread
  const int gl = part / vpp;                     // group of this thread inside the workgroup
  const int nv = HW * 8, n = HW * cg;
  __shared__ float red[kGnLineThreads];
  __shared__ float stat[8];
  float v[kGnLineVpt][VE];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kGnLineVpt; ++k) {
    const int i = threadIdx.x + k * kGnLineThreads;
    if (i < nv) {
      const int px = i >> 3;
      float t[8];
      if constexpr (VE == 8) load8(base + (long long)px * c_stride + part * VE, t);
      else { const float4 f = load4(base + (long long)px * c_stride + part * VE); t[0] = f.x; t[1] = f.y; t[2] = f.z; t[3] = f.w; }
#pragma unroll
      for (int e = 0; e < VE; ++e) { v[k][e] = t[e]; s += t[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[k][e] = 0.f;
    }
  }
  auto group_reduce = [&](float mine) -> float {   // sum over the threads of this thread's group (fixed order)
    red[threadIdx.x] = mine;
    __syncthreads();
    for (int o = kGnLineThreads / 2; o >= 8; o >>= 1) {      // keeps (thread & 7): the 8 vector lanes stay apart
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x < 8) {
      float t = 0.f;
      const int first = (threadIdx.x / vpp) * vpp;
      for (int q = 0; q < vpp; ++q) t += red[first + q];   // the vpp vector lanes of one group
      stat[threadIdx.x] = t;
    }
    __syncthreads();
    const float out = stat[gl * vpp];
    __syncthreads();
    return out;
  };
  const float mean = group_reduce(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < kGnLineVpt; ++k) {
    if ((int)threadIdx.x + k * kGnLineThreads < nv) {
#pragma unroll
      for (int e = 0; e < VE; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(group_reduce(q) / (float)n + eps);
  const int c0 = (g0 + gl) * cg + (part - gl * vpp) * VE;   // first channel of this thread's vectors
  float gm[VE], bt[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; }
#pragma unroll
  for (int k = 0; k < kGnLineVpt; ++k) {
    const int i = threadIdx.x + k * kGnLineThreads;
    if (i < nv) {
      float o[8];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float w = (v[k][e] - mean) * rstd * gm[e] + bt[e];
        o[e] = relu ? fmaxf(w, 0.f) : w;
      }
      T* dst = base + (long long)(i >> 3) * c_stride + part * VE;
      if constexpr (VE == 8) store8(dst, o);
      else store4(dst, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gap_kernel(const T* __restrict__ in, T* __restrict__ out, int HW, int C, const int* __restrict__ r_dev) {
  // one workgroup per roi. C % 8 == 0 and C / 8 divides the block: thread = (pixel lane pl, 8-channel chunk cc) reads whole 16-byte
  // (16-bit) / 32-byte vectors, sums its pixels pl, pl + L, ... in order, and the L pixel lanes of a chunk are added in lane order
  // through LDS - a fixed summation order, full-line reads (the first version walked 784 pixels with one 2-byte load each: 0.19 ms
  // per step on the DeepLab head). Other channel counts: thread t owns channels t, t + 256, ..., sequential sum over the pixels.
  const int r = blockIdx.x;
  if (r_dev != nullptr && r >= *r_dev) return;
  const int C8 = C >> 3;
  if ((C & 7) == 0 && C8 <= kBlock && kBlock % C8 == 0) {
    __shared__ float part[kBlock][8];
    const int L = kBlock / C8, cc = threadIdx.x % C8, pl = threadIdx.x / C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* p = in + (long long)r * HW * C + cc * 8;
    for (int i = pl; i < HW; i += L) {
      float t[8];
      load8(p + (long long)i * C, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (pl == 0) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += part[l * C8 + cc][e];
        o[e] = s / (float)HW;
      }
      store8(out + (long long)r * C + cc * 8, o);
    }
    return;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    const T* p = in + (long long)r * HW * C + c;
    for (int i = 0; i < HW; ++i) s += Elem<T>::load(p + (long long)i * C);
    Elem<T>::store(out + (long long)r * C + c, s / (float)HW);
  }
}

template <typename T>
__global__ void broadcast_hw_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int HW, int C, int ocs, int oco,
                                    const int* __restrict__ r_dev) {
  const int C4 = C >> 2;
  if (r_dev != nullptr) R = min(R, *r_dev);
  const long long total = (long long)R * HW * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long long t = i / C4;
    const int r = (int)(t / HW);
    store4(out + t * ocs + oco + c4 * 4, load4(in + (long long)r * C + c4 * 4));
  }
}

// exclusive prefix sum of the per-image detection counts (n_img is a batch size: one thread walks it)
__global__ void count_offsets_kernel(const int* __restrict__ counts, int n, int* __restrict__ offsets, int* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int s = 0;
  for (int i = 0; i < n; ++i) {
    offsets[i] = s;
    const int c = counts[i];
    s += c > 0 ? c : 0;
  }
  total[0] = s;
}

// ... with the rows capped at `limit` slots: image i keeps min(counts[i], limit - rows before it) boxes
__global__ void count_offsets_limited_kernel(const int* __restrict__ counts, int n, int limit, int* __restrict__ counts_out, int* __restrict__ offsets,
                                             int* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int s = 0;
  for (int i = 0; i < n; ++i) {
    offsets[i] = s;
    int c = counts[i];
    c = c > 0 ? c : 0;
    if (c > limit - s) c = limit - s;
    counts_out[i] = c;
    s += c;
  }
  total[0] = s;
}

}  // namespace

// runs CALL with the storage type bound to T (float | uint16_t = bf16 bits | _Float16)
#define DISPATCH_DTYPE(dtype, ...)                                                  \
  if ((dtype) == DP_F32) { using T = float; __VA_ARGS__; }                          \
  else if ((dtype) == DP_BF16) { using T = uint16_t; __VA_ARGS__; }                 \
  else if ((dtype) == DP_F16) { using T = f16_t; __VA_ARGS__; }                     \
  else return dp_fail(DP_ERR_BAD_ARG, "bad dtype %d", (int)(dtype));

extern "C" int dp_preprocess_u8(const dp_preprocess_params* p, dp_stream_t stream) {
  DP_REQUIRE(p && p->src && p->dst, "dp_preprocess_u8: null pointer");
  DP_REQUIRE(p->n_img > 0 && p->h > 0 && p->w > 0 && p->Hp >= p->h && p->Wp >= p->w, "dp_preprocess_u8: bad shape");
  hipStream_t s = as_stream(stream);
  if (p->paired) {
    DP_REQUIRE(p->Wp % 2 == 0, "dp_preprocess_u8: the paired layout needs an even padded width");
    const int Wq = p->Wp / 2 + 3;
    const long long cells = (long long)p->n_img * p->Hp * Wq;
    if (p->src_hwc) {
      DISPATCH_DTYPE(p->dtype,
                     hipLaunchKernelGGL((preprocess_paired_kernel<T, true>), dim3(grid_for(cells)), dim3(kBlock), 0, s, p->src, (T*)p->dst,
                                        p->n_img, p->h, p->w, p->Hp, Wq, p->mean[0], p->mean[1], p->mean[2], p->std[0], p->std[1], p->std[2]));
    } else {
      DISPATCH_DTYPE(p->dtype,
                     hipLaunchKernelGGL((preprocess_paired_kernel<T, false>), dim3(grid_for(cells)), dim3(kBlock), 0, s, p->src, (T*)p->dst,
                                        p->n_img, p->h, p->w, p->Hp, Wq, p->mean[0], p->mean[1], p->mean[2], p->std[0], p->std[1], p->std[2]));
    }
    return dp_check_launch("preprocess_paired_kernel");
  }
  DP_REQUIRE(!p->src_hwc, "dp_preprocess_u8: interleaved (HWC) source frames need the paired layout");
  const long long total = (long long)p->n_img * p->Hp * p->Wp;
  DISPATCH_DTYPE(p->dtype,
                 hipLaunchKernelGGL(preprocess_kernel<T>, dim3(grid_for(total)), dim3(kBlock), 0, s, p->src, (T*)p->dst,
                                    p->n_img, p->h, p->w, p->Hp, p->Wp, p->mean[0], p->mean[1], p->mean[2], p->std[0], p->std[1], p->std[2]));
  return dp_check_launch("preprocess_kernel");
}

extern "C" int dp_maxpool3x3s2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, dp_stream_t stream) {
  DP_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dp_maxpool3x3s2_nhwc: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  DP_REQUIRE((long long)H * W * C < (1ll << 31) && (long long)N * Ho < (1ll << 31), "dp_maxpool3x3s2_nhwc: image too large");
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3(N * Ho), dim3(kBlock), 0, s, (const T*)in, (T*)out, N, H, W, C, Ho, Wo));
  return dp_check_launch("maxpool3x3s2_kernel");
}

extern "C" int dp_subsample2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, dp_stream_t stream) {
  DP_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dp_subsample2_nhwc: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(subsample2_kernel<T>, dim3(grid_for(total)), dim3(kBlock), 0, s, (const T*)in, (T*)out, N, H, W, C, Ho, Wo));
  return dp_check_launch("subsample2_kernel");
}

extern "C" int dp_upsample_bilinear2x_nhwc(const void* in, void* out, int N, int H, int W, int C, int accumulate, int dtype,
                                           dp_stream_t stream) {
  DP_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dp_upsample_bilinear2x_nhwc: bad args");
  DP_REQUIRE(2ll * W * C < (1ll << 31) && 2ll * N * H < (1ll << 31), "dp_upsample_bilinear2x_nhwc: image too large");
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(upsample2x_kernel<T>, dim3(2 * N * H), dim3(kBlock), 0, s, (const T*)in, (T*)out, N, H, W, C, accumulate));
  return dp_check_launch("upsample2x_kernel");
}


extern "C" int dp_iuv_upsample_split(const dp_iuv_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_iuv_upsample_split: null params");
  if (p->R == 0) return DP_OK;
  DP_REQUIRE(p->in && p->coarse && p->fine && p->u && p->v, "dp_iuv_upsample_split: null pointer");
  DP_REQUIRE(p->R > 0 && p->Hs > 0 && p->Ws > 0 && p->n_coarse > 0 && p->n_fine > 0 && p->in_c >= p->n_coarse + 3 * p->n_fine,
             "dp_iuv_upsample_split: bad shape");
  DP_REQUIRE(p->in_c % 4 == 0 && p->Ws * p->in_c <= kIuvMaxRowFloats, "dp_iuv_upsample_split: row of %d x %d floats exceeds the LDS staging",
             p->Ws, p->in_c);
  DP_REQUIRE(p->R <= 65535, "dp_iuv_upsample_split: R too large");
  const int lds = 2 * (p->Ws + 1) * p->in_c * (int)sizeof(float);
  DP_REQUIRE(2 * p->Ws <= 256, "dp_iuv_upsample_split: output rows wider than 256");
  int xslots = 64;   // power of two >= the output width: a channel lane covers whole waves
  while (xslots < 2 * p->Ws) xslots *= 2;
  // four outputs per thread and 16-byte stores when every output row starts on a 16-byte boundary (policy key iuv_quad = 0: A/B);
  // passed as a negative slot count = - (power of two >= output width / 4)
  const bool al16 = ((reinterpret_cast<uintptr_t>(p->coarse) | reinterpret_cast<uintptr_t>(p->fine) | reinterpret_cast<uintptr_t>(p->u) |
                      reinterpret_cast<uintptr_t>(p->v)) & 15) == 0;
  if ((2 * p->Ws) % 4 == 0 && al16 && dp_policy().iuv_quad != 0) {
    int qs = 16;
    while (qs < (2 * p->Ws) / 4) qs *= 2;
    xslots = -qs;
  }
  hipLaunchKernelGGL(iuv_upsample_split_kernel, dim3(p->Hs + 1, p->R), dim3(256), lds, as_stream(stream), p->in, p->R, p->Hs, p->Ws,
                     p->in_c, p->n_coarse, p->n_fine, p->coarse, p->fine, p->u, p->v, xslots, p->r_dev);
  return dp_check_launch("iuv_upsample_split_kernel");
}

extern "C" int dp_count_offsets(const int32_t* counts, int n_img, int32_t* offsets, int32_t* total, dp_stream_t stream) {
  DP_REQUIRE(counts && offsets && total && n_img > 0, "dp_count_offsets: bad args");
  hipLaunchKernelGGL(count_offsets_kernel, dim3(1), dim3(64), 0, as_stream(stream), counts, n_img, offsets, total);
  return dp_check_launch("count_offsets_kernel");
}

extern "C" int dp_count_offsets_limited(const int32_t* counts, int n_img, int limit, int32_t* counts_out, int32_t* offsets, int32_t* total,
                                        dp_stream_t stream) {
  DP_REQUIRE(counts && counts_out && offsets && total && n_img > 0 && limit >= 0, "dp_count_offsets_limited: bad args");
  hipLaunchKernelGGL(count_offsets_limited_kernel, dim3(1), dim3(64), 0, as_stream(stream), counts, n_img, limit, counts_out, offsets, total);
  return dp_check_launch("count_offsets_limited_kernel");
}

extern "C" int dp_merge_upsample2x_nhwc(const void* base, const void* const* ups, int n_ups, void* out, int N, int H, int W, int C,
                                        int dtype, dp_stream_t stream) {
  DP_REQUIRE(base && ups && out && n_ups >= 1 && n_ups <= 3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dp_merge_upsample2x_nhwc: bad args");
  DP_REQUIRE(2ll * W * C < (1ll << 31) && 2ll * N * H < (1ll << 31), "dp_merge_upsample2x_nhwc: image too large");
  for (int k = 0; k < n_ups; ++k) DP_REQUIRE(ups[k], "dp_merge_upsample2x_nhwc: null map %d", k);
  const void* u0 = ups[0];
  const void* u1 = n_ups > 1 ? ups[1] : nullptr;
  const void* u2 = n_ups > 2 ? ups[2] : nullptr;
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(merge_up2x_kernel<T>, dim3(N * (H + 1)), dim3(kBlock), 0, s, (const T*)base, (const T*)u0, (const T*)u1, (const T*)u2, n_ups, (T*)out, N, H, W, C));
  return dp_check_launch("merge_up2x_kernel");
}

extern "C" int dp_groupnorm_relu_nhwc(const dp_groupnorm_params* p, dp_stream_t stream) {
  DP_REQUIRE(p, "dp_groupnorm_relu_nhwc: null params");
  if (p->R == 0) return DP_OK;
  DP_REQUIRE(p->x && p->gamma && p->beta && p->R > 0 && p->HW > 0 && p->groups > 0 && p->C % p->groups == 0 &&
                 p->c_off >= 0 && p->c_off + p->C <= p->c_stride,
             "dp_groupnorm_relu_nhwc: bad args");
  hipStream_t s = as_stream(stream);
  const dim3 g(p->R * p->groups), b(kBlock);
  {
    // the register-resident form: 16-byte vectors inside one group (and 16-byte aligned rows), at most 8 (16-bit) / 16 (fp32) vectors per thread
    const int es = p->dtype == DP_F32 ? 4 : 2, ve = 16 / es, cg = p->C / p->groups;
    const long long nv = (long long)p->HW * (cg / ve);
    const bool vec_ok = cg % ve == 0 && p->c_stride % ve == 0 && p->c_off % ve == 0 && (reinterpret_cast<uintptr_t>(p->x) & 15) == 0;
    const int reg_on = (int)dp_policy().gn_reg;     // policy key gn_reg: 0 = the element-wise three-sweep kernel, 1 = one group per workgroup only
    // whole-line form: the groups sharing a 128-byte line go to one workgroup
    const int line_groups = (cg * es <= 128 && 128 % (cg * es) == 0) ? 128 / (cg * es) : 0;
    if (reg_on >= 2 && vec_ok && line_groups >= 1 && p->groups % line_groups == 0 && (p->c_stride * es) % 128 == 0 && (p->c_off * es) % 128 == 0 &&
        (reinterpret_cast<uintptr_t>(p->x) & 127) == 0 && (long long)p->HW * 8 <= (long long)kGnLineThreads * kGnLineVpt && p->HW * 8 >= kGnLineThreads) {
      const dim3 gl(p->R * (p->groups / line_groups)), bl(kGnLineThreads);
      if (p->dtype == DP_F32) hipLaunchKernelGGL(groupnorm_line_kernel<float>, gl, bl, 0, s, (float*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      else if (p->dtype == DP_BF16) hipLaunchKernelGGL(groupnorm_line_kernel<uint16_t>, gl, bl, 0, s, (uint16_t*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      else hipLaunchKernelGGL(groupnorm_line_kernel<f16_t>, gl, bl, 0, s, (f16_t*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      return dp_check_launch("groupnorm_line_kernel");
    }
    if (reg_on && vec_ok && nv <= (long long)kBlock * (es == 4 ? 16 : 8)) {
      if (p->dtype == DP_F32) hipLaunchKernelGGL((groupnorm_reg_kernel<float, 16>), g, b, 0, s, (float*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      else if (p->dtype == DP_BF16) hipLaunchKernelGGL((groupnorm_reg_kernel<uint16_t, 8>), g, b, 0, s, (uint16_t*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      else hipLaunchKernelGGL((groupnorm_reg_kernel<f16_t, 8>), g, b, 0, s, (f16_t*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev);
      return dp_check_launch("groupnorm_reg_kernel");
    }
  }
  DISPATCH_DTYPE(p->dtype,
                 hipLaunchKernelGGL(groupnorm_kernel<T>, g, b, 0, s, (T*)p->x, p->HW, p->C, p->c_stride, p->c_off, p->groups, p->gamma, p->beta, p->eps, p->relu, p->r_dev));
  return dp_check_launch("groupnorm_kernel");
}

extern "C" int dp_global_avgpool_nhwc(const void* in, void* out, int R, int HW, int C, int dtype, const int32_t* r_dev, dp_stream_t stream) {
  if (R == 0) return DP_OK;
  DP_REQUIRE(in && out && R > 0 && HW > 0 && C > 0, "dp_global_avgpool_nhwc: bad args");
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(gap_kernel<T>, dim3(R), dim3(kBlock), 0, s, (const T*)in, (T*)out, HW, C, r_dev));
  return dp_check_launch("gap_kernel");
}

extern "C" int dp_broadcast_hw_nhwc(const void* in, void* out, int R, int HW, int C, int out_c_stride, int out_c_off, int dtype,
                                    const int32_t* r_dev, dp_stream_t stream) {
  if (R == 0) return DP_OK;
  DP_REQUIRE(in && out && R > 0 && HW > 0 && C > 0 && C % 4 == 0 && out_c_off % 4 == 0 && out_c_off + C <= out_c_stride,
             "dp_broadcast_hw_nhwc: bad args");
  const long long total = (long long)R * HW * (C / 4);
  hipStream_t s = as_stream(stream);
  DISPATCH_DTYPE(dtype,
                 hipLaunchKernelGGL(broadcast_hw_kernel<T>, dim3(grid_for(total)), dim3(kBlock), 0, s, (const T*)in, (T*)out, R, HW, C, out_c_stride, out_c_off, r_dev));
  return dp_check_launch("broadcast_hw_kernel");
}
