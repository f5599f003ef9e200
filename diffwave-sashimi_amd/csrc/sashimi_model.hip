// SaShiMi backbone behind the C ABI (placeholder until the S4 kernels land).
#include "model.h"
namespace dws {
dws_model* make_sashimi(const dws_model_desc&) {
    set_error(DWS_ERR_UNSUPPORTED, "sashimi backbone not built into this libdws.so yet");
    return nullptr;
}
}  // namespace dws
