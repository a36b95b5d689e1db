// fft_regs_f64.h -- the register-resident FFT building blocks of fft_regs_impl.h in double
// (namespace rtlws::f64; "f2" there is double2), for spectrum_f64_fused.hip.
#ifndef RTLWS_FFT_REGS_F64_H
#define RTLWS_FFT_REGS_F64_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtlws {
namespace f64 {

#define RTLWS_FR_REAL double
#define RTLWS_FR_C2 double2
#define RTLWS_FR_MAKE make_double2
#define RTLWS_FR_FMA fma
#define RTLWS_FR_LIT(x) x
#include "fft_regs_impl.h"
#undef RTLWS_FR_REAL
#undef RTLWS_FR_C2
#undef RTLWS_FR_MAKE
#undef RTLWS_FR_FMA
#undef RTLWS_FR_LIT

}  // namespace f64
}  // namespace rtlws
#endif
