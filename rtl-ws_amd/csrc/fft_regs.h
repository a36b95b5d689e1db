// fft_regs.h -- register-resident FFT building blocks shared by the fused kernels
// (spectrum_fused.hip, spectrum_fused_v2.hip): radix-4 / radix-16 butterflies, the
// fused-multiply-add forms of the twiddled butterflies, the last-pass radix-R3, and
// the Hann weights generated from two lane constants.  gfx950, wave64; no state.
#ifndef RTLWS_FFT_REGS_H
#define RTLWS_FFT_REGS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtlws {

#define RTLWS_FR_REAL float
#define RTLWS_FR_C2 float2
#define RTLWS_FR_MAKE make_float2
#define RTLWS_FR_FMA fmaf
#define RTLWS_FR_LIT(x) x##f
#include "fft_regs_impl.h"
#undef RTLWS_FR_REAL
#undef RTLWS_FR_C2
#undef RTLWS_FR_MAKE
#undef RTLWS_FR_FMA
#undef RTLWS_FR_LIT

}  // namespace rtlws
#endif
