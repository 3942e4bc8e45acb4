/*
 * densepose_hip.h - C ABI of the MI355X (gfx950) DensePose inference kernels.
 *
 * The reference (dajes/DensePose-TorchScript) has no FFI of its own: every arithmetic step of
 * DefaultPredictor.forward is a torch ATen / torchvision op call. This header declares one entry
 * point per such call site; each comment cites the reference line(s) it
 * replaces (paths relative to the reference root). INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch types; every function enqueues work on the
 *     given hipStream_t and returns immediately: 0 = ok, <0 = dp_status error (nothing launched).
 *   - functions never allocate; workspaces are caller-provided.
 *   - activations are NHWC ("pixels x channels"), channel count padded to a multiple of 8;
 *     dtype is DP_F32 (parity mode, exact fp32 MFMA), DP_BF16 (throughput mode, fp32 accumulate) or DP_F16
 *     (IEEE half storage + fp32 accumulate: the reference's `.half()` export, /root/reference/export.py:36-37,
 *     run.py:26, with everything that is not a GEMM operand kept in fp32).
 *   - boxes, scores, anchors, decode, IoU, softmax are always fp32.
 */
#ifndef DENSEPOSE_HIP_H
#define DENSEPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dp_stream_t; /* hipStream_t */

enum dp_dtype { DP_F32 = 0, DP_BF16 = 1, DP_F16 = 2 };

enum dp_status {
  DP_OK = 0,
  DP_ERR_BAD_ARG = -1,      /* null pointer / inconsistent shape */
  DP_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels were built for */
  DP_ERR_LAUNCH = -3        /* hipGetLastError() after launch */
};

#define DP_ABI_VERSION 8
int dp_abi_version(void);
/* human-readable reason of the last non-zero return on this thread */
const char* dp_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Kernel-policy overrides (ABI 6). The reference has no counterpart: ATen picks its kernels by itself
 * (wrappers.py:105 F.conv2d). Here the choice of a kernel fixes a layer's summation order, hence its bits, so it must
 * not depend on anything but the layer and this table. NO entry point reads the process environment; the table holds
 * the defaults of csrc/dp_policy.h until a caller changes it:
 *   tests pin a kernel class ("conv_big", "conv_ws", "conv_rows", "conv_pws" ...), tools/ A/B a schedule.
 * Process-wide, not thread-safe against concurrent launches (set it before launching). Unknown key: DP_ERR_BAD_ARG.
 * ------------------------------------------------------------------------------------------- */
int dp_set_policy(const char* key, int64_t value);
int dp_get_policy(const char* key, int64_t* value);
void dp_reset_policy(void);              /* every key back to its default */
int dp_policy_num_keys(void);
const char* dp_policy_key(int index);    /* NULL past the end */

/* ---------------------------------------------------------------------------------------------
 * K2  rcnn.py:162,180  (x - pixel_mean) / pixel_std ; F.pad(right/bottom, 0) ; + NCHW->NHWC, C 3->8
 *     src: uint8 [3,h,w] planar (the output of the uint8 resize, defaults.py:89); dst: [Hp,Wp,8]
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* src; /* [n_img][3][h][w] */
  void* dst;          /* [n_img][Hp][Wp][8] dtype */
  int32_t n_img, h, w, Hp, Wp;
  int32_t dtype;
  float mean[3], inv_std_unused[3], std[3];
  int32_t paired;     /* 0: dst = [n_img][Hp][Wp][8] (3 real channels + 5 zeros per pixel).
                         1: dst = [n_img][Hp][Wq][8], Wq = Wp / 2 + 3: the image as 4-channel pixels (3 real + 1 zero), shifted
                            right by 3 pixels, TWO pixels per 8-channel cell - cell j holds pixels 2j - 3 and 2j - 2. The 7x7
                            stride-2 stem (resnet.py:350-353) then is a 7 x 4-tap convolution with stride (2, 1) over cells:
                            K = 7 * 4 * 8 = 224 instead of 49 * 8 = 392 (3 of 8 channels used -> 6 of 8). */
  int32_t src_hwc;    /* 1 (paired layout only): src is [n_img][h][w][3], the frames as the caller hands them over (defaults.py:76-78).
                         At scale 1 - a frame that already has the test size - the uint8 resize of defaults.py:89 is the identity, and
                         the two resize passes + the planar intermediate are skipped. */
} dp_preprocess_params;
int dp_preprocess_u8(const dp_preprocess_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3-K7, K12, K14, K16-K18  wrappers.py:105-111 F.conv2d (+FrozenBN batch_norm.py:54-62 folded,
 * + F.relu / residual add resnet.py:203-204 / FPN top-down add fpn.py:152-155), nn.Linear
 * box_head.py:71-73 (H=W=1), ConvTranspose2d chart.py:45-59 (4 sub-pixel 2x2 convolutions).
 *
 * Implicit GEMM: out[m][co] = sum_k A[m][k] * Wt[co][k], m = (n,ho,wo), k = (tap, c).
 * The K axis is described by `ktab`: one int4 {dy, dx, c0, valid} per 16-byte chunk of K, so any
 * tap set (3x3, dilated, 7x7 stem, 2x2 sub-pixel) uses the same kernel.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* in;        /* [N][H][W][Cin] dtype, Cin % 8 == 0 */
  const void* weight;    /* [Cout_w][Kpad] dtype, K-contiguous; Cout_w % 128 == 0, zero padded */
  const int32_t* ktab;   /* [Kpad / chunk][4] = {dy, dx, c0, valid | tap << 8} */
  const float* bias;     /* [Cout_w] fp32 (BN shift or conv bias), zero padded */
  const void* residual;  /* optional, dtype; element (n,ho,wo,c) at n*rsN + (ho>>rshift)*rsH + (wo>>rshift)*rsW + c */
  void* out;             /* element (n,ho,wo,c) at n*osN + ho*osH + wo*osW + c */
  int32_t N, H, W, Cin;
  int32_t Ho, Wo, Cout;  /* Cout = channels stored (multiple of 8, <= Cout_w) */
  int32_t Cout_w, Kpad;
  int32_t stride, ntaps;  /* ntaps = number of (dy,dx) taps in ktab; ktab[k].w = valid | tap_index << 8 */
  int64_t osN, osH, osW;
  int64_t rsN, rsH, rsW;
  int32_t rshift;
  int32_t relu;
  int32_t dtype;         /* dtype of in / weight / residual */
  int32_t out_f32;       /* 1: out is fp32 regardless of dtype */
  int32_t hi_off, wi_off; /* input pixel of output (ho,wo), tap (dy,dx): (ho*stride + hi_off + dy, wo*stride_w + wi_off + dx) */
  int32_t stride_w;       /* horizontal stride; 0 = same as `stride` (only the paired-pixel stem uses stride 2 x 1) */
  /* Optional fused 1x1 head on the ReLU output of this convolution (rpn.py:168-171: objectness_logits + anchor_deltas on
   * relu(conv(x))): head_out[m][0..15] = head_w[16][Cout] . relu(conv + bias)[m] + head_b, the hidden tensor rounded to the
   * storage type as if it had been stored. Needs dp_conv2d_kernel_class() == 2 with Cout == 256, 16-bit storage, relu = 1,
   * no residual (else DP_ERR_UNSUPPORTED); `out` may then be NULL: the hidden tensor is never written. */
  const void* head_w;     /* [16][Cout] dtype, plain row-major (rows = head channels, zero rows for unused ones) */
  const float* head_b;    /* [16] */
  float* head_out;        /* [N*Ho*Wo][16] fp32 */
  /* Scheduling hint from the host, which knows its stream graph: 0 = this launch has the chip (more or less) to itself,
   * 1 = it runs beside other large launches on other streams. The weight-stationary 3x3 kernel is a persistent launch with a
   * static split of the work over its workgroups (weights resident in registers): alone on the chip it runs one workgroup per
   * CU; with hint 1 it is split over twice as many, shorter workgroups so that the dispatcher can rebalance when some CUs are
   * held by another stream (a one-per-CU split cannot, and loses its advantage over the tiled kernels);
   * 2 = it runs beside a chain of small latency-bound launches on another stream (top-k / NMS): one workgroup per CU again, but
   * on 7/8 of the CUs, so that the chain's workgroups find a free CU at once instead of waiting for one of this launch to end.
   * The result is bit-identical in all three cases. */
  int32_t shared_chip;
  /* A tensor added AFTER the activation: out = act(conv + bias) + post (weight-stationary 3x3 kernel only, i.e. kernel class 6;
   * DP_ERR_UNSUPPORTED elsewhere). The DensePose decoder (roi_head.py:71-79) sums its four scale heads after their ReLUs:
   *   post_mode 1: post_res is a tensor of the OUTPUT's geometry [N, H, W, Cout] (the running sum of the low-resolution heads),
   *   post_mode 2: post_res is [N, H / 2, W / 2, Cout] and is added through the bilinear x2 up-sampling (align_corners = False)
   *                the heads end in (roi_head.py:63): out = relu(conv(p2)) + up2(sum of the low heads), H and W even.
   * Both in fp32 before the one rounding to the storage type. */
  int32_t post_mode;
  const void* post_res;
  /* Device-side image count: when non-NULL, only the first *n_dev of the N images hold data and tiles that start behind them are
   * skipped (N keeps sizing the launch and the tensors). The DensePose head runs on R detected boxes (roi_head.py:126-158); R is
   * known on the device only - sizing its launches on the host costs a device -> host round trip in the middle of every step.
   * Rows behind *n_dev inside the last live tile are computed on whatever the input holds: every image (ROI) is independent,
   * their outputs are never read. Honoured by the tiled kernels (classes 0 - 4) and classes 7 / 8 / 10 (which size their work from the live
   * count and write nothing behind it); a launch with n_dev is never given to the persistent kernels that ignore it (classes 5 and 6: they
   * would do the full work on all N images). */
  const int32_t* n_dev;
  /* Second source of a pointwise (1 tap, stride 1) layer: K = Cin channels of `in` followed by Cin2 channels of `in2`, an
   * [N, H2, W2, Cin2] tensor read at pixel (ho * stride2, wo * stride2); Kpad = Cin + Cin2, both multiples of 64 bytes; the
   * weight matrix is the one of a (Cin + Cin2)-channel 1x1 convolution. The projection shortcut of the first block of a ResNet
   * stage (resnet.py:189-205: out = relu(conv3(t2) + shortcut(x))) is then part of conv3's K axis - the shortcut tensor is neither
   * written nor read back. LDS-ring kernels only (DP_ERR_UNSUPPORTED otherwise). */
  const void* in2;
  int32_t H2, W2, Cin2, stride2;
  /* Split-K (ABI 4): split_k > 1 cuts the K axis into split_k segments of whole 64-byte planes; workgroup (tile, segment) writes
   * fp32 partial sums to split_ws ([split_k][N*Ho*Wo][Cout] floats, caller-owned scratch) and a second pass adds the segments in
   * index order, then bias and activation. For layers with a long K and few pixel tiles (the box head's fc1: K = 12544 over 1000
   * rows per image, box_head.py:60-67; res5's 3x3): one workgroup per tile streams megabytes of weights while most CUs idle. The
   * value is a property of the LAYER: pass the same split_k whatever the batch, and a row's summation order - hence its bits -
   * does not depend on what else is in the batch. Honoured for 16-bit storage, layers the LDS-ring kernels take (64-byte K planes inside
   * one tap, tensors below 2 GiB), plain NHWC output, at least two planes per segment, no residual / head / second source / post_res /
   * n_dev; anything else runs UNSPLIT on the kernel it would get without the field (no error). 0 or 1 = off. */
  int32_t split_k;
  void* split_ws;
  /* Grouped launch (ABI 6): n_groups = 2 .. 4 convolutions that share the input, the bias, the geometry, the output strides and every
   * other field run in ONE launch; group g multiplies by weight_g[g] (with tap table ktab_g[g]: same Kpad / ntaps) and stores at
   * out_g[g] (`weight`, `ktab`, `out` are ignored). The chart predictor's ConvTranspose2d(k4, s2, p1) layers (chart.py:45-60) are four
   * sub-pixel 2x2 convolutions - output pixel (2i + a, 2j + b) for (a, b) in {0, 1}^2 - whose separate launches each fill 1.5 rounds of
   * the chip; as one launch the tiles of the four parity classes of a pixel tile sit side by side (same input rows, same XCD).
   * Per-output arithmetic is that of the separate launch (same kernel, same K order): bit-identical. Kernel classes 3 / 4 (the 128-cout
   * LDS-ring tiles) only, no residual / head / second source / post_res / split_k: DP_ERR_UNSUPPORTED otherwise. 0 or 1 = off. */
  int32_t n_groups;
  const void* weight_g[4];
  const int32_t* ktab_g[4];
  void* out_g[4];
  /* Summation-order pin (ABI 8): 1 = this launch must produce the bits of the LDS-ring kernel family (classes 1 - 6 share one K
   * order: 64-byte planes, channel block x tap, one chain). Kernel classes with an order of their own (10: the one-wave-per-SIMD
   * weight-stationary 3x3 kernel) are then not chosen. For a caller that runs the SAME layer with a fused head - ring kernels only -
   * whenever the launch is large enough and as a plain convolution otherwise (the RPN's 3x3, rpn.py:166-172): the fused / unfused
   * choice depends on the batch, the bits of a frame must not. 0 = the library's choice. */
  int32_t ring_order;
} dp_conv_params;
int dp_conv2d_nhwc(const dp_conv_params* p, dp_stream_t stream);
/* which kernel dp_conv2d_nhwc will launch for these parameters: 0 = generic 128x64 tile, 1 = generic 128x128 tile, 2 = 256x256 LDS-ring tile,
 * 3 = 128x128 LDS-ring tile, 4 = 256x128 two-workgroup ring tile, 5 = streaming 1x1 (resident weights), 6 = weight-stationary 3x3
 * (128 -> 128 / 256 -> 256 channels, weights in registers), 7 = row-streaming K-split weight-stationary 3x3 (512 input channels, or
 * 256 -> 512: the DensePose head v1convx.py:44-59 / deeplab.py:64-74 and res5's conv2 resnet.py:195-197; it honours n_dev by sizing
 * its work from the live image count, and its per-pixel summation order - fixed, but not the other kernels' - is why a layer it
 * takes runs on it for EVERY batch size), 8 = the same with 32 pixels per step and the waves split 4 K quarters x 2 cout halves (512
 * input channels on maps whose width gives strip groups of at most 8 images, e.g. the DensePose head's 28-wide ROI maps, with or
 * without n_dev, and the 256 -> 512 first layer of that head on a 64-cout instance; chosen by the geometry alone, its summation order
 * differs from class 7's; classes 7 / 8 cut a launch whose tensors exceed 2 GiB into several launches over chunks of whole strip groups
 * instead of handing it to another class), 9 = weight-stationary pointwise kernel (1 tap, stride 1, K = 512 / 1024 / 2048 channels:
 * conv1 of res4 / res5 and conv3 of res5 resnet.py:189-205, the FPN laterals fpn.py:140-157, fc2 box_head.py:71-73; a 256 KiB slice
 * of the weights in registers, pixels once through LDS, K slices added in a fixed order of its own - chosen by the channel counts
 * alone), 10 = weight-stationary 3x3 kernel with ONE wave per SIMD (256 -> 256 channels: res4 conv2 resnet.py:195-197, the FPN output
 * convolutions fpn.py:134-157, the DensePose decoder's scale heads roi_head.py:48-68; 32 couts x 1152 K of the weights in each wave's 512
 * registers, four output rows of a 16-pixel strip per step, every bit of step bookkeeping in the MFMAs' shadow; policy key wsq_shape picks
 * the v_mfma_f32_16x16x32 form (default) or the 32x32x16 form; a summation order of its own - chosen by the per-image geometry and the
 * channel counts alone, never by N; honours post_res; a launch with ring_order = 1 is never given to it) - profiling / roofline bookkeeping only */
int dp_conv2d_kernel_class(const dp_conv_params* p);
/* pixel rows of the tile dp_conv2d_nhwc will use for these parameters (the 256-cout ring kernel picks 128 .. 256 rows in
 * steps of 32 to fit the launch into whole rounds of the chip) - profiling / roofline bookkeeping only */
int dp_conv2d_tile_rows(const dp_conv_params* p);

/* ---------------------------------------------------------------------------------------------
 * K4  resnet.py:189-205  BottleneckBlock.forward, everything after conv1, in ONE launch:
 *       t2  = relu(conv2(t1) + b2)                 3x3, Cmid -> Cmid         (resnet.py:195-197)
 *       out = relu(conv3(t2) + b3 + residual)      1x1, Cmid -> Cout         (resnet.py:199-205)
 *       next_t1 = relu(conv1'(out) + b1')          1x1, Cout -> Cmid_next: conv1 of the NEXT block
 *                                                  (resnet.py:192-193), optional (next_t1 == NULL: skipped)
 *     First block of a stage (resnet.py:189-190, the projection shortcut), sc_in != NULL:
 *       out = relu([W3 | Ws][t2 ; sc_in] + b3 + bs)  w3 = the dual-source matrix over Cmid + Csc channels (Kpad3 = their sum),
 *                                                  b3 = the summed shifts; residual and next_t1 must be NULL. Same bits as
 *                                                  dp_conv2d_nhwc on conv2, then on that matrix with in2 = sc_in.
 * Weights, tap table and biases are the packed operands dp_conv2d_nhwc takes for the same layers
 * (FrozenBN folded, batch_norm.py:54-62); results are bit-identical to three dp_conv2d_nhwc calls.
 * Fused shapes: dp_bottleneck_tail_supported() (16-bit storage, Cmid 64, Cout 256 = the res2 blocks);
 * anything else returns DP_ERR_UNSUPPORTED and the caller runs the layers one by one.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* t1;        /* [N][H][W][Cmid] dtype: conv1 output of this block */
  const void* residual;  /* [N][H][W][Cout] dtype: block input or its shortcut projection */
  void* out;             /* [N][H][W][Cout] dtype */
  void* next_t1;         /* optional [N][H][W][Cmid_next] dtype */
  const void* w2; const void* w3; const void* w1n;   /* packed weights ([Cout_w][Kpad], see dp_conv_params) */
  const int32_t* ktab2;  /* conv2's tap table */
  const float* b2; const float* b3; const float* b1n;
  int32_t N, H, W;
  int32_t Cmid, Cout, Cmid_next;
  int32_t Kpad2, Kpad3, Kpad1n;
  int32_t ntaps2, hi_off2, wi_off2;
  int32_t k_order2;      /* K order of conv2's packed weights: 0 = channel-block major (64-byte planes outer, taps inner),
                            1 = tap major (K = tap * Cmid + channel); the fused kernel takes 1 */
  int32_t dtype;
  /* ABI 7 */
  int32_t Csc;           /* channels of sc_in (64) */
  const void* sc_in;     /* optional [N][H][W][Csc] dtype: the block's input, second source of conv3's K axis (see above) */
} dp_bottleneck_params;
int dp_bottleneck_tail_supported(const dp_bottleneck_params* p);
int dp_bottleneck_tail_nhwc(const dp_bottleneck_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  resnet.py:199-205 of one BottleneckBlock + resnet.py:192-193 of the NEXT block, in ONE launch (ABI 6):
 *       out     = relu(conv3(t2) + b3 + residual)      1x1, Cmid -> Cout
 *       next_t1 = relu(conv1'(out) + b1')              1x1, Cout -> Cmid_next
 * for the plain blocks of res3 (Cmid 128, Cout 512, Cmid_next 128, 16-bit storage): both weight matrices stay in the register file, the
 * block output is written once and never read back. `out` is bit-identical to dp_conv2d_nhwc's; next_t1 adds eight 64-channel partial
 * sums in a fixed order (its own summation order: a call site uses this entry point for every batch size or never). Pointwise layers:
 * M = N * H * W pixels, plain NHWC tensors. Anything else: DP_ERR_UNSUPPORTED (dp_bottleneck_pair_supported() == 0), the caller runs
 * the two layers one by one.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* t2;        /* [M][Cmid] dtype: conv2 output of this block */
  const void* residual;  /* [M][Cout] dtype: block input */
  void* out;             /* [M][Cout] dtype */
  void* next_t1;         /* [M][Cmid_next] dtype */
  const void* w3; const void* w1n;      /* packed weights ([Cout_w][Kpad], see dp_conv_params) */
  const float* b3; const float* b1n;
  int64_t M;
  int32_t Cmid, Cout, Cmid_next, Kpad3, Kpad1n, dtype;
} dp_pair_params;
int dp_bottleneck_pair_supported(const dp_pair_params* p);
int dp_bottleneck_pair_nhwc(const dp_pair_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  resnet.py:350-354  BasicStem.forward in ONE launch: 7x7 stride-2 pad-3 conv + FrozenBN + ReLU + F.max_pool2d(3, 2, 1).
 * in: the paired-pixel image of dp_preprocess_u8 (paired = 1); weight / bias: the stem packed over that layout (K = 7 kernel
 * rows x 4 cells x 8 = 224, Kpad 256: pack.stem_paired_conv / dp_pack_conv_weights with taps (dy, dxp)). The conv output is
 * never written. Bit-identical to dp_conv2d_nhwc + dp_maxpool3x3s2_nhwc. Fused shapes: dp_stem_pool_supported() (16-bit
 * storage, 64 output channels); anything else returns DP_ERR_UNSUPPORTED and the caller runs the two layers.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* in;        /* [N][Hp][Wp / 2 + 3][8] dtype */
  const void* weight;    /* [Cout_w][Kpad] dtype */
  const float* bias;     /* [Cout_w] */
  void* out;             /* [N][Hp / 4][Wp / 4][Cout] dtype */
  int32_t N, Hp, Wp;     /* padded image size (multiples of 4) */
  int32_t Cout, Kpad;
  int32_t dtype;
} dp_stem_pool_params;
int dp_stem_pool_supported(const dp_stem_pool_params* p);
int dp_stem_pool_nhwc(const dp_stem_pool_params* p, dp_stream_t stream);

/* K3  resnet.py:353  F.max_pool2d(k=3, s=2, p=1), NHWC */
int dp_maxpool3x3s2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, dp_stream_t stream);

/* K6  fpn.py:199  F.max_pool2d(k=1, s=2) == x[:, ::2, ::2, :] */
int dp_subsample2_nhwc(const void* in, void* out, int N, int H, int W, int C, int dtype, dp_stream_t stream);

/* K14 roi_head.py:63,71-79  out = (accumulate ? out : 0) + bilinear_x2(in), align_corners=False */
int dp_upsample_bilinear2x_nhwc(const void* in, void* out, int N, int H, int W, int C, int accumulate, int dtype,
                                dp_stream_t stream);
/* K14 roi_head.py:71-79  out = base + sum_k bilinear_x2(ups[k]), k < n_ups <= 3, summed in list order in fp32:
 * the decoder's level sum in one pass. ups are [N,H,W,C], base/out [N,2H,2W,C] (out may alias base). */
int dp_merge_upsample2x_nhwc(const void* base, const void* const* ups, int n_ups, void* out, int N, int H, int W, int C,
                             int dtype, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9   proposal_utils.py:76   logits_i.topk(min(HiWiA, k)) per image and level (sorted, desc)
 * K8   rpn.py:375-394 + box_regression.py:74-112 + anchor_generator.py:165-179  decode of the survivors
 *      proposal_utils.py:102-116  isfinite filter, clip_boxes (Q1: x to size[1], y to size[0]), nonempty >= 0
 * head: [n_img][Hi][Wi][16] fp32: channel a = objectness of anchor a (a<3), 3 + 4a + c = delta c.
 * Output per image: L*kmax candidate slots: boxes [.,4], scores, level id, valid flag.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* head;      /* level map */
  int32_t n_img, Hi, Wi, A, head_c; /* A anchors per cell (3), head_c channel stride (16) */
  int32_t stride_px;      /* 4..64 */
  float cell_anchors[3][4];
  int32_t level, kmax, slot_off, slots_per_img; /* candidates of this level go to slots [slot_off, slot_off + min(n,kmax)) */
  float clip_x, clip_y;   /* Q1: clip_x = H_pad, clip_y = W_pad */
  float* cand_boxes;      /* [n_img][slots_per_img][4] */
  float* cand_scores;     /* [n_img][slots_per_img] */
  int32_t* cand_level;    /* [n_img][slots_per_img] */
  int32_t* cand_valid;    /* [n_img][slots_per_img] */
  void* workspace;        /* dp_rpn_topk_workspace_bytes */
} dp_rpn_level_params;
int64_t dp_rpn_topk_workspace_bytes(int n_img, int Hi, int Wi, int A);
int dp_rpn_topk_decode(const dp_rpn_level_params* p, dp_stream_t stream);
/* all (<= 5) FPN levels of a batch in one select launch; every level needs its OWN workspace */
int dp_rpn_topk_decode_levels(const dp_rpn_level_params* levels, int n_levels, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K10/K13  nms.py:20 -> torchvision batched_nms + keep[:post_topk]  (proposal_utils.py:118-126,
 *          fast_rcnn.py:129-132). Greedy, IoU > thr strict, fp32, stable score order, groups never
 *          suppress each other; reproduces torchvision's "coordinate trick" rounding when
 *          4*n_valid <= trick_max_numel (4000 on the reference's CPU path).
 * in : per image n_slots candidates (boxes, scores, group, valid) in any order
 * out: kept boxes/scores in descending score order, at most max_out, + count; also source slot idx.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* boxes; const float* scores; const int32_t* group; const int32_t* valid;
  int32_t n_img, n_slots;
  float iou_thr;        /* suppressed iff iou > iou_thr, in float. torchvision's CPU kernel compares the float IoU with its DOUBLE threshold
                           (nms_kernel.cpp): a caller holding a double passes the largest float not above it - the same predicate */
  int32_t max_out;
  int32_t trick_max_numel;
  float* out_boxes;     /* [n_img][max_out][4] */
  float* out_scores;    /* [n_img][max_out] */
  int32_t* out_index;   /* [n_img][max_out] slot index of each kept box */
  int32_t* out_count;   /* [n_img] */
  void* workspace;      /* dp_nms_workspace_bytes */
} dp_nms_params;
int64_t dp_nms_workspace_bytes(int n_img, int n_slots);
int dp_batched_nms(const dp_nms_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K11/K15  poolers.py:187-227 ROIPooler (level = clamp(floor(4 + log2(sqrt(area)/224 + 1e-8)))) +
 *          roi_align.py:58 torchvision roi_align(aligned=False, sampling_ratio=2), NHWC.
 * boxes [n_img][max_rois][4] with counts[n_img]; output row r = img*max_rois + j -> [P][P][C].
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* feat[4];     /* level maps [n_img][Hl][Wl][C] dtype; n_levels = 1 or 4 */
  int32_t Hl[4], Wl[4];
  float scale[4];
  int32_t n_levels, min_level; /* min_level = 2 for p2..p5; single map: n_levels=1 */
  int32_t C, P, sampling;
  const float* boxes; const int32_t* counts;
  int32_t n_img, max_rois;
  void* out;               /* [n_img*max_rois][P][P][C] dtype (rows >= count untouched) */
  int32_t dtype;
  int32_t compact;         /* 1: output rows are compacted (row = prefix(counts)[img] + j) using roi_offsets */
  const int32_t* roi_offsets; /* [n_img] exclusive prefix of counts (compact mode) */
} dp_roi_align_params;
int dp_roi_align_nhwc(const dp_roi_align_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K13  fast_rcnn.py:257-326,105-127  apply_deltas(weights 10,10,5,5) + softmax + finite filter +
 *      score > thresh (Q2: no clip). Produces NMS candidates per image.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* logits;   /* [n_img*max_rois][ld] : col 0 = person, col 1 = background, cols 2..5 = deltas */
  int32_t ld;
  const float* prop_boxes; const int32_t* prop_counts; /* [n_img][max_rois][4], [n_img] */
  int32_t n_img, max_rois;
  float wx, wy, ww, wh, score_thresh;
  float* cand_boxes; float* cand_scores; int32_t* cand_group; int32_t* cand_valid; /* [n_img][max_rois] */
} dp_box_decode_params;
int dp_box_decode_score(const dp_box_decode_params* p, dp_stream_t stream);

/* K19 postprocessing.py:43-54  scale_boxes, nonempty, clip to (H,W) of the original frame */
typedef struct {
  const float* boxes; const int32_t* counts; int32_t n_img, max_dets;
  const float* scale_xy;    /* [n_img][2] */
  const float* out_hw;      /* [n_img][2] original (H, W) as float */
  float* out_boxes; int32_t* keep; /* [n_img][max_dets][4], [n_img][max_dets] */
} dp_postprocess_params;
int dp_postprocess_boxes(const dp_postprocess_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K18  chart.py:72-74,86-89  F.interpolate(bilinear x2) of the 4 deconv outputs + split + NHWC->NCHW
 * in : [R][Hs][Ws][Cin_stride] fp32 (channels: coarse | fine | u | v), out: 4 NCHW fp32 tensors
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* in; int32_t R, Hs, Ws, in_c;
  int32_t n_coarse, n_fine;   /* 2|15, 25 */
  float* coarse; float* fine; float* u; float* v;  /* [R][C][2Hs][2Ws] */
  const int32_t* r_dev;       /* when non-NULL: only the first *r_dev of the R boxes are processed (see dp_conv_params.n_dev) */
} dp_iuv_params;
int dp_iuv_upsample_split(const dp_iuv_params* p, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K17  deeplab.py:45,90,101,119 nn.GroupNorm(32, C) (+ReLU), deeplab.py:97 AdaptiveAvgPool2d(1),
 *      deeplab.py:109 bilinear 1x1 -> HxW (== broadcast)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  void* x;               /* in/out [R][HW][c_stride] dtype, channels [c_off, c_off + C) */
  int32_t R, HW, C, c_stride, c_off, groups;
  const float* gamma; const float* beta; float eps; int32_t relu; int32_t dtype;
  const int32_t* r_dev;  /* when non-NULL: only the first *r_dev of the R boxes are processed (see dp_conv_params.n_dev) */
} dp_groupnorm_params;
int dp_groupnorm_relu_nhwc(const dp_groupnorm_params* p, dp_stream_t stream);
/* r_dev (may be NULL): device-side count of live boxes, as above */
int dp_global_avgpool_nhwc(const void* in, void* out, int R, int HW, int C, int dtype, const int32_t* r_dev, dp_stream_t stream);
int dp_broadcast_hw_nhwc(const void* in, void* out, int R, int HW, int C, int out_c_stride, int out_c_off, int dtype,
                         const int32_t* r_dev, dp_stream_t stream);
/* exclusive prefix sum of the per-image detection counts (fast_rcnn.py:86-140 keeps at most max_dets per image): offsets[i] = the
 * first row of image i in the compact ROI list, total[0] = R. One tiny launch instead of a read-back + host cumsum + upload. */
int dp_count_offsets(const int32_t* counts, int n_img, int32_t* offsets, int32_t* total, dp_stream_t stream);
/* (ABI 5) ... with the compact ROI list capped at `limit` rows: counts_out[i] = min(counts[i], limit - offsets[i]) (>= 0), offsets and
 * total follow the capped counts. The DensePose branch (roi_head.py:126-158) is launched before the host knows R; its buffers are
 * sized for a high-water mark of recent steps instead of n x DETECTIONS_PER_IMAGE slots, this cap keeps the device inside them, and the
 * host - which reads the TRUE counts one step later anyway - runs the branch again with more slots in the rare step that overflowed. */
int dp_count_offsets_limited(const int32_t* counts, int n_img, int limit, int32_t* counts_out, int32_t* offsets, int32_t* total, dp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Host-side weight pipeline (CPU code, no stream): canonical fp32 parameters -> the packed operands of
 * dp_conv2d_nhwc / dp_bottleneck_tail_nhwc. What the reference does implicitly by handing nn.Conv2d /
 * FrozenBatchNorm2d parameters to ATen (wrappers.py:104-112, batch_norm.py:31,54-62).
 * ------------------------------------------------------------------------------------------- */
/* FrozenBN fold: w_out[co][..] = w[co][..] * scale[co], shift[co] = beta - mean * scale, scale = gamma * (1 / sqrt(var + eps)),
 * all in fp32 (per_cout = Cin * R * S elements per output channel; w_out may alias w) */
int dp_fold_frozen_bn(const float* w, int Cout, int per_cout, const float* gamma, const float* beta, const float* mean,
                      const float* var, float eps, float* w_out, float* shift_out);
/* tap list of an R x S kernel: taps_out[n] = {r * dilation, s * dilation}, kernel_pos_out[n] = r * S + s, row-major; with
 * stride 1 and in_h, in_w > 0 the taps that cannot land inside an in_h x in_w map for any output pixel are dropped
 * (deeplab.py:33: dilation 56 on a 28x28 map). Returns the number of taps kept (< 0: error); hi_off = wi_off = -pad. */
int dp_conv_taps(int R, int S, int pad, int dilation, int stride, int in_h, int in_w, int32_t* taps_out, int32_t* kernel_pos_out);
typedef struct {
  int32_t Cout, ntaps, Cin;   /* wmat: [Cout][ntaps][Cin] fp32 (tap t of an OIHW kernel = w[:, :, r, s] of dp_conv_taps) */
  int32_t cin_alloc;          /* channels of the NHWC tensor the layer reads (multiple of 8, >= Cin; extra channels get zero weights) */
  int32_t dtype;              /* storage type of the packed weights */
  int32_t tap_major;          /* 0: multi-tap layers whose cin_alloc is whole 64-byte planes are packed channel-block major, taps
                                 inner (what the LDS-ring kernels expect); 1: K = tap * cin_alloc + channel (generic kernel only:
                                 Cout <= 64; the order dp_bottleneck_tail_nhwc takes for conv2) */
} dp_pack_params;
typedef struct {
  int32_t cout;      /* channels the layer stores (Cout rounded up to 8) */
  int32_t cout_w;    /* rows of the packed matrix (Cout rounded up to 128) */
  int32_t kpad;      /* elements per row (K rounded up to 128 bytes) */
  int32_t n_ktab;    /* tap table entries (int32[4] each) = kpad / elements per 16-byte chunk */
  int32_t plane_major; /* 1: channel-block-major K order was used */
} dp_pack_info;
int dp_pack_conv_info(const dp_pack_params* p, dp_pack_info* info);
/* wmat as above, taps [ntaps][2] = {dy, dx}, bias [Cout] or NULL; outputs are HOST buffers sized by dp_pack_conv_info:
 * w_out [cout_w][kpad] dtype, ktab_out [n_ktab][4], bias_out [cout_w] fp32 */
int dp_pack_conv_weights(const dp_pack_params* p, const float* wmat, const int32_t* taps, const float* bias, void* w_out,
                         int32_t* ktab_out, float* bias_out);

/* ---------------------------------------------------------------------------------------------
 * "next" rows
 * f1  defaults.py:89  F.interpolate(uint8, scale_factor=k, bilinear) - bit-exact fixed-point resize
 * f2  visualizer.py:10-30  per-detection resample to the box + coarse-mask * fine-argmax + UV gather
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t* src; /* [H][W][3] interleaved (src_hwc=1) or [3][H][W] planar (src_hwc=0) */
  uint8_t* tmp;       /* [3][H][ow] horizontal-pass result (uint8, like ATen's two-pass kernel) */
  uint8_t* dst;       /* [3][oh][ow] planar */
  int32_t H, W, oh, ow, src_hwc;
  const int32_t* xtab; /* [ow][4] = {i0, i1, W0, W1} fixed-point weights of the horizontal pass */
  const int32_t* ytab; /* [oh][4] vertical pass */
  int32_t xprec, yprec; /* weight precision bits of each pass */
} dp_resize_params;
int dp_resize_u8_bilinear(const dp_resize_params* p, dp_stream_t stream);
/* n frames of ONE geometry (a video, run.py:42-57) in one launch per pass: srcs = host array of n device pointers
 * (p->src is ignored), p->tmp = [n][3][H][ow], p->dst = [n][3][oh][ow] - the batch tensor the engine consumes. */
int dp_resize_u8_bilinear_batch(const dp_resize_params* p, const void* const* srcs, int n, dp_stream_t stream);
/* (ABI 5) defaults.py:84-89 + rcnn.py:156-181 for frames whose scale is not 1: the horizontal pass into p->tmp, then ONE launch that does
 * the vertical pass, (x - mean) / std, the zero padding to q->Hp x q->Wp and the paired-pixel layout the stem reads (q->paired must be 1,
 * q->h / q->w = p->oh / p->ow, q->n_img = n <= 64; q->src and p->dst are ignored): the resized uint8 batch is never written. Bit-identical
 * to dp_resize_u8_bilinear_batch followed by dp_preprocess_u8. */
int dp_resize_preprocess_u8_batch(const dp_resize_params* p, const void* const* srcs, int n, const dp_preprocess_params* q, dp_stream_t stream);
/* (ABI 5) rcnn.py:156-181 for n <= 64 frames that already have the test size (defaults.py:84-89 is the identity at scale 1), each in its own
 * allocation: dp_preprocess_u8 in the paired layout with srcs = host array of n device pointers instead of one stacked tensor (q->src is
 * ignored, q->paired must be 1, q->n_img = n, q->src_hwc says whether a frame is [h][w][3] or [3][h][w]). Bit-identical to stacking the
 * frames and calling dp_preprocess_u8 - without the 3 * h * w * n byte copy in front of every batch. */
int dp_preprocess_u8_frames(const dp_preprocess_params* q, const void* const* srcs, int n, dp_stream_t stream);

typedef struct {
  const float* coarse; const float* fine; const float* u; const float* v; /* [R][C][S][S] */
  int32_t R, S, n_coarse, n_fine;
  const int32_t* box_xywh;   /* [R][4] int (truncated), w,h >= 1 */
  const int64_t* out_offset; /* [R] element offset of detection r in labels / (2 planes) uv */
  uint8_t* labels;           /* sum(h*w) */
  float* uv;                 /* 2 * sum(h*w): per detection [2][h][w] */
  int32_t max_hw;            /* max h*w over detections (grid sizing) */
} dp_iuv_extract_params;
int dp_iuv_extract(const dp_iuv_extract_params* p, dp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DENSEPOSE_HIP_H */
