// wfk_fir.hip -- FIR stage (predistort(ker=...), waveforms/distortion.py:329-337).
// Placeholder until the rocFFT overlap-save path lands in this round.
#include <hip/hip_runtime.h>
#include "wfk.h"
extern "C" {
int wfk_fir_plan_create(const double*, int32_t, int64_t, int32_t, int, wfk_fir_plan** out) {
  if (out) *out = nullptr;
  return WFK_EUNSUP;
}
int wfk_fir_apply(wfk_fir_plan*, const void*, int64_t, void*, int64_t, void*) { return WFK_EUNSUP; }
int wfk_fir_plan_destroy(wfk_fir_plan*) { return WFK_OK; }
}
