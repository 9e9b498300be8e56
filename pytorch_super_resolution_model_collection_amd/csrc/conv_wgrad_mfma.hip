// placeholder until the MFMA weight-gradient kernel lands (see below in this round)
#include "srk_common.h"
namespace srk {
bool conv_wgrad_mfma_supported(const srk_conv_desc& d) { (void)d; return false; }
size_t conv_wgrad_mfma_ws(const srk_conv_desc& d) { (void)d; return 0; }
int conv_wgrad_mfma(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                    float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
  (void)d; (void)x; (void)dy; (void)mask; (void)dw; (void)db; (void)beta; (void)ws; (void)ws_bytes; (void)s;
  set_error("conv_wgrad_mfma: not built");
  return SRK_ERR_UNSUPPORTED;
}
}  // namespace srk
