/* ORACLE (test infrastructure): plain-C restatement of torchvision 0.16.2's CPU roi_align and nms kernels
 * (torchvision/csrc/ops/cpu/roi_align_kernel.cpp, roi_align_common.h, nms_kernel.cpp - a third-party dependency of
 * the reference that is neither vendored in /root/reference nor installed here; algorithm restated from its
 * published source). Same arithmetic, same operation order as oracle/ops_ref.py's numpy/torch version, which it is
 * tested against bit for bit (tests/test_oracle_ops.py). Used so that the CPU baseline is not dominated by Python.
 * Build: make -C oracle   (gcc -O2 -fopenmp, NO -ffast-math, -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* input [N][C][H][W], rois [K][5] = (batch, x1, y1, x2, y2), out [K][C][P][P] */
void oracle_roi_align_f32(const float* input, int N, int C, int H, int W, const float* rois, int K, int P, float spatial_scale,
                          int sampling, int aligned, float* out) {
  (void)N;
#pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < K; ++k) {
    const float* r = rois + (size_t)k * 5;
    const int b = (int)r[0];
    const float off = aligned ? 0.5f : 0.0f;
    const float rsw = r[1] * spatial_scale - off, rsh = r[2] * spatial_scale - off;
    const float rew = r[3] * spatial_scale - off, reh = r[4] * spatial_scale - off;
    float rw = rew - rsw, rh = reh - rsh;
    if (!aligned) {
      rw = rw > 1.0f ? rw : 1.0f;
      rh = rh > 1.0f ? rh : 1.0f;
    }
    const float bin_h = rh / (float)P, bin_w = rw / (float)P;
    const int g = sampling;
    const float count = (float)(g * g > 1 ? g * g : 1);
    /* pre-calc sample positions / weights for this roi (as pre_calc_for_bilinear_interpolate does) */
    const int ns = P * g;
    int* ylo = (int*)malloc(sizeof(int) * ns * 4);
    int *yhi = ylo + ns, *xlo = yhi + ns, *xhi = xlo + ns;
    float* wy = (float*)malloc(sizeof(float) * ns * 4);
    float *hyv = wy + ns, *wx = hyv + ns, *hxv = wx + ns;
    char* bad = (char*)malloc(2 * ns);
    for (int p = 0; p < P; ++p)
      for (int i = 0; i < g; ++i) {
        const int s = p * g + i;
        float y = rsh + (float)p * bin_h + ((float)i + 0.5f) * bin_h / (float)g;
        float x = rsw + (float)p * bin_w + ((float)i + 0.5f) * bin_w / (float)g;
        bad[s] = (y < -1.0f || y > (float)H);
        bad[ns + s] = (x < -1.0f || x > (float)W);
        if (y <= 0) y = 0;
        if (x <= 0) x = 0;
        int yl = (int)y, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        ylo[s] = yl; yhi[s] = yh; xlo[s] = xl; xhi[s] = xh;
        wy[s] = y - (float)yl; hyv[s] = 1.0f - wy[s];
        wx[s] = x - (float)xl; hxv[s] = 1.0f - wx[s];
      }
    for (int c = 0; c < C; ++c) {
      const float* f = input + ((size_t)b * C + c) * H * W;
      float* o = out + ((size_t)k * C + c) * P * P;
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          float acc = 0.0f;
          for (int iy = 0; iy < g; ++iy) {
            const int sy = ph * g + iy;
            for (int ix = 0; ix < g; ++ix) {
              const int sx = pw * g + ix;
              if (bad[sy] || bad[ns + sx]) continue;
              const float w1 = hyv[sy] * hxv[sx], w2 = hyv[sy] * wx[sx], w3 = wy[sy] * hxv[sx], w4 = wy[sy] * wx[sx];
              acc += w1 * f[ylo[sy] * W + xlo[sx]] + w2 * f[ylo[sy] * W + xhi[sx]] + w3 * f[yhi[sy] * W + xlo[sx]] +
                     w4 * f[yhi[sy] * W + xhi[sx]];
            }
          }
          o[ph * P + pw] = acc / count;
        }
    }
    free(ylo); free(wy); free(bad);
  }
}

/* boxes [n][4], order [n] = indices sorted by score descending (stable); keep_out [n]; returns number kept */
int oracle_nms_f32(const float* boxes, const int64_t* order, int n, float thr, int64_t* keep_out) {
  unsigned char* sup = (unsigned char*)calloc(n > 0 ? n : 1, 1);
  float* areas = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) areas[i] = (boxes[i * 4 + 2] - boxes[i * 4 + 0]) * (boxes[i * 4 + 3] - boxes[i * 4 + 1]);
  int nk = 0;
  for (int _i = 0; _i < n; ++_i) {
    const int64_t i = order[_i];
    if (sup[i]) continue;
    keep_out[nk++] = i;
    const float ix1 = boxes[i * 4], iy1 = boxes[i * 4 + 1], ix2 = boxes[i * 4 + 2], iy2 = boxes[i * 4 + 3], iarea = areas[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      const int64_t j = order[_j];
      if (sup[j]) continue;
      const float xx1 = ix1 > boxes[j * 4] ? ix1 : boxes[j * 4];
      const float yy1 = iy1 > boxes[j * 4 + 1] ? iy1 : boxes[j * 4 + 1];
      const float xx2 = ix2 < boxes[j * 4 + 2] ? ix2 : boxes[j * 4 + 2];
      const float yy2 = iy2 < boxes[j * 4 + 3] ? iy2 : boxes[j * 4 + 3];
      const float w = (xx2 - xx1) > 0.0f ? (xx2 - xx1) : 0.0f;
      const float h = (yy2 - yy1) > 0.0f ? (yy2 - yy1) : 0.0f;
      const float inter = w * h;
      const float ovr = inter / (iarea + areas[j] - inter);
      if (ovr > thr) sup[j] = 1;
    }
  }
  free(sup); free(areas);
  return nk;
}
