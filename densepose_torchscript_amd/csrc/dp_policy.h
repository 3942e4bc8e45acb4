// Kernel-policy overrides (tests, calibration tools): ONE process-wide table behind dp_set_policy() / dp_get_policy() of
// include/densepose_hip.h. No entry point reads the process environment: which kernel a layer runs on - hence its summation order,
// hence its bits - is a property of the layer and of this table's DEFAULTS; the table only moves when a caller says so
// (tests pin a kernel class, tools/ A/B a schedule; densepose_torchscript_amd/lib.py applies DP_<KEY> environment variables ONCE at load
// for the command-line tools).
#pragma once
#include <stdint.h>

struct DpPolicy {
  // dp_conv2d_nhwc (dp_conv.hip)
  int64_t conv_big = -1;            // -1 = policy decides; 0 generic only, 1 256x256 ring whenever legal, 2 128x128 ring, 3 256x128 ring, 5 streaming 1x1
  int64_t conv_stream = 1;          // 0 = the streaming 1x1 kernel is never chosen
  int64_t conv_tp = 0;              // 4..8 = tile height (x 32 pixels) of the 256-cout ring kernel, 0 = chosen per launch
  int64_t conv_ring2_m = 230000;    // pixel count from which 128-cout layers take the 256x128 two-workgroup tile (0 = never)
  int64_t conv_policy = 3;          // (DP_EXPERIMENTS builds only) 0 / 1 = the round-1 tile policies
  // weight-stationary 3x3 (dp_conv_ws.hip)
  int64_t conv_ws = 1;              // 0 never, 4 only launches that have the chip to themselves
  int64_t ws_min_m = 256;           // fewest output pixels the kernel takes
  int64_t ws_over_shared = 2, ws_over_alone = 1, ws_reserve = 1;   // workgroups per CU slot with / without a neighbour stream, CU groups left free
  // weight-stationary 3x3 on v_mfma_f32_32x32x16, one wave per SIMD (dp_conv_wq.hip)
  int64_t conv_wsq = 1;             // 0 = never chosen (256 -> 256 layers then run on the kernels above)
  int64_t wsq_shape = 16;           // 16 = conv3x3_ws1_kernel (v_mfma_f32_16x16x32), 32 = conv3x3_wsq_kernel (v_mfma_f32_32x32x16): different summation orders
  int64_t wsq_min_hw = 128;         // fewest output pixels PER IMAGE the kernel takes (never a function of the batch)
  // row kernels (dp_conv_rows.hip)
  int64_t conv_rows = 1;            // 0 never, 2 the 16-pixel form also for n_dev launches and 256 -> 512
  int64_t conv_rows2 = 1;           // 0 = the 32-pixel form is never chosen
  int64_t conv_rows2_256 = -1;      // 0 never, 1 every 256-channel layer, -1 only cout counts other than 256
  int64_t conv_rows2_maxg = 8;      // largest strip group of the 512-channel layers
  int64_t conv_rows2_lockstep = 0;  // 1 = all eight waves on one schedule
  int64_t conv_rows_chain = 0;      // (DP_EXPERIMENTS builds only) 1 = the barrier-free chain form
  int64_t rows_chunk_bytes = (1ll << 31) - 1;   // tensor bytes per launch of the row kernels (tests lower it)
  // weight-stationary pointwise (dp_conv_pw.hip)
  int64_t conv_pws = 1;             // 0 = never chosen
  int64_t pws_skew = 0;             // (DP_EXPERIMENTS builds only) 1 = the halves of the workgroup half a step apart, two barriers per step
  // res4's conv3 -> next-conv1 pair kernel (dp_pair256.hip): built, verified, measured SLOWER than the two launches at the benchmark geometry
  // (profiles/r6_pair256_experiments.txt: the weights of both layers cross L2 -> CU once per 131 pixels): off unless asked for
  int64_t pair256 = 0;
  // the rest
  int64_t tail_kernel = 0;          // dp_bottleneck_tail_nhwc: 1 = the tile kernel instead of the strip walker
  int64_t roi_tab = 1;              // dp_roi_align_nhwc: 0 = the per-sample kernel for every sampling ratio
  int64_t iuv_quad = 1;             // dp_iuv_upsample_split: 0 = one output per thread
  int64_t gn_reg = 2;               // dp_groupnorm_relu_nhwc: 0 element-wise three-sweep kernel, 1 one group per workgroup, 2 whole-line form
};

DpPolicy& dp_policy();
