// Host-side weight pipeline of the C ABI (no GPU involved): canonical fp32 weights -> the packed operands dp_conv2d_nhwc /
// dp_bottleneck_tail_nhwc consume. A non-Python host binds these instead of re-implementing the row permutation and the
// tap table. Replaces what the reference does implicitly when it hands nn.Conv2d / FrozenBatchNorm2d parameters to ATen
// (/root/reference/detectron2/layers/wrappers.py:104-112, batch_norm.py:31,54-62).
//
//   * dp_fold_frozen_bn     y = (x - mean) * rsqrt(var + eps) * gamma + beta folded into (weights, per-cout shift)
//   * dp_conv_taps          the (dy, dx) tap list of an R x S kernel with dilation; taps that can never land inside the map for
//                           ANY output pixel are dropped (deeplab.py:33: dilation 56 on a 28x28 ROI map keeps the centre tap)
//   * dp_pack_conv_info /   [Cout][ntaps][Cin] fp32 -> [Cout_pad128][Kpad] storage type (rows of every 64-cout block permuted
//     dp_pack_conv_weights  for the register epilogue, K = (channel block, tap, channel) or (tap, channel), zero padded to the
//                           128-byte K step) + the per-16-byte-chunk tap table + the fp32 bias, zero padded
//
// densepose_torchscript_amd/pack.py is a thin caller of these; tests/test_pack.py holds an independent numpy restatement.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "dp_common.h"

#pragma clang fp contract(off)   // the fold is compared bit for bit with its numpy restatement: no fused multiply-adds

namespace {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// physical row i*16 + q*4 + e of a 64-cout block carries logical cout (i>>1)*32 + q*8 + (i&1)*4 + e (dp_conv.hip, store_tile)
inline int row_perm(int r) {
  const int i = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return (i >> 1) * 32 + q * 8 + (i & 1) * 4 + e;
}

inline uint16_t f32_to_bf16(float f) {   // round to nearest even, NaN stays NaN (what torch's .to(bfloat16) does)
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

inline uint16_t f32_to_f16(float f) {
  const _Float16 h = (_Float16)f;        // IEEE round to nearest even, overflow -> inf (torch's .half())
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

struct Geometry {
  int es, ch, pe, cout, cout_w, k, kpad, n_ktab, plane_major;
};

int geometry(const dp_pack_params* p, Geometry* g) {
  DP_REQUIRE(p != nullptr, "dp_pack_conv: null params");
  DP_REQUIRE(p->dtype == DP_F32 || p->dtype == DP_BF16 || p->dtype == DP_F16, "dp_pack_conv: bad dtype %d", p->dtype);
  DP_REQUIRE(p->Cout > 0 && p->ntaps > 0 && p->Cin > 0, "dp_pack_conv: bad shape [%d][%d][%d]", p->Cout, p->ntaps, p->Cin);
  DP_REQUIRE(p->cin_alloc >= p->Cin && p->cin_alloc % 8 == 0, "dp_pack_conv: cin_alloc=%d must be a multiple of 8 and >= Cin=%d",
             p->cin_alloc, p->Cin);
  DP_REQUIRE(p->ntaps < (1 << 23), "dp_pack_conv: too many taps");
  DP_REQUIRE((long long)p->ntaps * p->cin_alloc < (1ll << 30), "dp_pack_conv: K too large");
  g->es = p->dtype == DP_F32 ? 4 : 2;
  g->ch = 16 / g->es;
  g->pe = 64 / g->es;
  g->cout = round_up(p->Cout, 8);
  g->cout_w = round_up(p->Cout, 128);
  g->k = p->ntaps * p->cin_alloc;
  g->kpad = round_up(g->k, 128 / g->es);
  g->n_ktab = g->kpad / g->ch;
  g->plane_major = (!p->tap_major && p->ntaps > 1 && p->cin_alloc % g->pe == 0) ? 1 : 0;
  DP_REQUIRE(!p->tap_major || p->Cout <= 64 || p->ntaps == 1,
             "dp_pack_conv: tap-major K is only legal for layers on the table-driven generic kernel (Cout <= 64)");
  return DP_OK;
}

// position of (tap, channel) on the K axis
inline int k_index(const Geometry& g, int ntaps, int cin_alloc, int tap, int c) {
  if (!g.plane_major) return tap * cin_alloc + c;
  const int cb = c / g.pe, within = c - cb * g.pe;
  return (cb * ntaps + tap) * g.pe + within;
}

}  // namespace

extern "C" int dp_fold_frozen_bn(const float* w, int Cout, int per_cout, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, float* w_out, float* shift_out) {
  DP_REQUIRE(w && gamma && beta && mean && var && w_out && shift_out && Cout > 0 && per_cout > 0, "dp_fold_frozen_bn: bad args");
  for (int co = 0; co < Cout; ++co) {
    const float scale = gamma[co] * (1.0f / sqrtf(var[co] + eps));
    shift_out[co] = beta[co] - mean[co] * scale;
    const float* src = w + (long long)co * per_cout;
    float* dst = w_out + (long long)co * per_cout;
    for (int i = 0; i < per_cout; ++i) dst[i] = src[i] * scale;
  }
  return DP_OK;
}

extern "C" int dp_conv_taps(int R, int S, int pad, int dilation, int stride, int in_h, int in_w, int32_t* taps_out,
                            int32_t* kernel_pos_out) {
  if (!(R > 0 && S > 0 && dilation > 0 && stride > 0 && pad >= 0 && taps_out && kernel_pos_out)) {
    dp_fail(DP_ERR_BAD_ARG, "dp_conv_taps: bad args");
    return DP_ERR_BAD_ARG;
  }
  int n = 0;
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) {
      const int dy = r * dilation, dx = s * dilation;
      if (in_h > 0 && in_w > 0 && stride == 1 &&
          (dy - pad >= in_h || dy - pad <= -in_h || dx - pad >= in_w || dx - pad <= -in_w))
        continue;   // contributes exactly 0 to every output pixel
      taps_out[2 * n] = dy;
      taps_out[2 * n + 1] = dx;
      kernel_pos_out[n] = r * S + s;
      ++n;
    }
  return n;
}

extern "C" int dp_pack_conv_info(const dp_pack_params* p, dp_pack_info* info) {
  Geometry g;
  if (int rc = geometry(p, &g)) return rc;
  DP_REQUIRE(info != nullptr, "dp_pack_conv_info: null info");
  info->cout = g.cout;
  info->cout_w = g.cout_w;
  info->kpad = g.kpad;
  info->n_ktab = g.n_ktab;
  info->plane_major = g.plane_major;
  return DP_OK;
}

extern "C" int dp_pack_conv_weights(const dp_pack_params* p, const float* wmat, const int32_t* taps, const float* bias, void* w_out,
                                    int32_t* ktab_out, float* bias_out) {
  Geometry g;
  if (int rc = geometry(p, &g)) return rc;
  DP_REQUIRE(wmat && taps && w_out && ktab_out && bias_out, "dp_pack_conv_weights: null pointer");
  const int nt = p->ntaps, ci = p->Cin, ca = p->cin_alloc;
  // ---- weights: logical [cout_w][kpad] fp32, then rows permuted inside every 64-cout block and converted
  std::vector<float> row((size_t)g.kpad);
  for (int pr = 0; pr < g.cout_w; ++pr) {   // physical row
    const int co = (pr & ~63) + row_perm(pr & 63);
    std::fill(row.begin(), row.end(), 0.0f);
    if (co < p->Cout) {
      const float* src = wmat + (long long)co * nt * ci;
      for (int t = 0; t < nt; ++t)
        for (int c = 0; c < ci; ++c) row[(size_t)k_index(g, nt, ca, t, c)] = src[t * ci + c];
    }
    // stored in 1 KiB tiles of 16 rows x 64 bytes of K (dp_wtile_off, dp_common.h): element k of row pr = byte k * es of the row
    const int n_planes = g.kpad * g.es / 64;
    unsigned char* wb = static_cast<unsigned char*>(w_out);
    for (int i = 0; i < g.kpad; ++i) {
      const int kb = i * g.es;
      unsigned char* dst = wb + dp_wtile_off(pr, kb / 64, (kb % 64) / 16, n_planes) + kb % 16;
      if (p->dtype == DP_F32) {
        memcpy(dst, &row[(size_t)i], sizeof(float));
      } else {
        const uint16_t h = p->dtype == DP_BF16 ? f32_to_bf16(row[(size_t)i]) : f32_to_f16(row[(size_t)i]);
        memcpy(dst, &h, sizeof(h));
      }
    }
  }
  // ---- tap table: one {dy, dx, c0, valid | tap << 8} per 16-byte chunk of K
  for (int kc = 0; kc < g.n_ktab; ++kc) {
    int32_t* e = ktab_out + 4 * kc;
    const int k0 = kc * g.ch;
    e[0] = e[1] = e[2] = e[3] = 0;
    if (k0 >= g.k) continue;
    int tap, c0;
    if (g.plane_major) {
      const int plane = k0 / g.pe, within = k0 - plane * g.pe;
      const int cb = plane / nt;
      tap = plane - cb * nt;
      c0 = cb * g.pe + within;
    } else {
      tap = k0 / ca;
      c0 = k0 - tap * ca;
    }
    e[0] = taps[2 * tap];
    e[1] = taps[2 * tap + 1];
    e[2] = c0;
    e[3] = 1 | (tap << 8);
  }
  for (int i = 0; i < g.cout_w; ++i) bias_out[i] = (bias && i < p->Cout) ? bias[i] : 0.0f;
  return DP_OK;
}
