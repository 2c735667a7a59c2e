// placeholder until the MFMA kernel lands
#include "lv_kernels.h"
namespace lv {
int launch_prefill_attn(const PrefillParams&, hipStream_t) {
  return fail(LV_ERR_UNSUPPORTED, "prefill: not built yet");
}
}  // namespace lv
