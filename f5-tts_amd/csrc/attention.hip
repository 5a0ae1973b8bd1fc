// attention.hip — flash-style attention (fp16 MFMA); see DESIGN.md.  (placeholder until the kernel lands)
#include "kernels.h"
bool flash_attn_available() { return false; }
hipError_t launch_flash_attn(const f16*, const f16*, const f16*, int, int, int, const int32_t*, f16*, f16*, hipStream_t) {
  return hipErrorNotSupported;
}
