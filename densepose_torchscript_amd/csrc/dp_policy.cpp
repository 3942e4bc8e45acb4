// dp_set_policy / dp_get_policy / dp_reset_policy (include/densepose_hip.h): the one table of kernel-policy overrides.
#include "dp_policy.h"
#include <string.h>
#include "../../include/densepose_hip.h"

namespace {
DpPolicy g_policy;

struct Key {
  const char* name;
  int64_t DpPolicy::*field;
};
const Key kKeys[] = {
    {"conv_big", &DpPolicy::conv_big},
    {"conv_stream", &DpPolicy::conv_stream},
    {"conv_tp", &DpPolicy::conv_tp},
    {"conv_ring2_m", &DpPolicy::conv_ring2_m},
    {"conv_policy", &DpPolicy::conv_policy},
    {"conv_ws", &DpPolicy::conv_ws},
    {"ws_min_m", &DpPolicy::ws_min_m},
    {"ws_over_shared", &DpPolicy::ws_over_shared},
    {"ws_over_alone", &DpPolicy::ws_over_alone},
    {"ws_reserve", &DpPolicy::ws_reserve},
    {"conv_wsq", &DpPolicy::conv_wsq},
    {"wsq_min_hw", &DpPolicy::wsq_min_hw},
    {"wsq_shape", &DpPolicy::wsq_shape},
    {"conv_rows", &DpPolicy::conv_rows},
    {"conv_rows2", &DpPolicy::conv_rows2},
    {"conv_rows2_256", &DpPolicy::conv_rows2_256},
    {"conv_rows2_maxg", &DpPolicy::conv_rows2_maxg},
    {"conv_rows2_lockstep", &DpPolicy::conv_rows2_lockstep},
    {"conv_rows_chain", &DpPolicy::conv_rows_chain},
    {"rows_chunk_bytes", &DpPolicy::rows_chunk_bytes},
    {"conv_pws", &DpPolicy::conv_pws},
    {"pws_skew", &DpPolicy::pws_skew},
    {"pair256", &DpPolicy::pair256},
    {"tail_kernel", &DpPolicy::tail_kernel},
    {"roi_tab", &DpPolicy::roi_tab},
    {"iuv_quad", &DpPolicy::iuv_quad},
    {"gn_reg", &DpPolicy::gn_reg},
};
constexpr int kNumKeys = (int)(sizeof(kKeys) / sizeof(kKeys[0]));

const Key* find_key(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumKeys; ++i)
    if (strcmp(kKeys[i].name, name) == 0) return &kKeys[i];
  return nullptr;
}
}  // namespace

DpPolicy& dp_policy() { return g_policy; }

extern "C" int dp_set_policy(const char* key, int64_t value) {
  const Key* k = find_key(key);
  if (!k) return DP_ERR_BAD_ARG;
  g_policy.*(k->field) = value;
  return DP_OK;
}

extern "C" int dp_get_policy(const char* key, int64_t* value) {
  const Key* k = find_key(key);
  if (!k || !value) return DP_ERR_BAD_ARG;
  *value = g_policy.*(k->field);
  return DP_OK;
}

extern "C" void dp_reset_policy(void) { g_policy = DpPolicy(); }

extern "C" int dp_policy_num_keys(void) { return kNumKeys; }

extern "C" const char* dp_policy_key(int index) { return (index >= 0 && index < kNumKeys) ? kKeys[index].name : nullptr; }
