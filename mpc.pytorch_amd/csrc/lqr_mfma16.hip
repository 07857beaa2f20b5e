// placeholder until the fused kernel lands
#include "lqr_common.h"
namespace mpclqr {
bool mfma16_supported(const StepParams<float> &) { return false; }
int launch_step_mfma16(const StepParams<float> &, hipStream_t) { set_last_error("mfma16 kernel not built"); return MPC_E_DIMS; }
}
