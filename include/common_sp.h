/*
 * common_sp.h -- sample types shared by every entry point of the drop-in.
 *
 * Layout-identical to the reference's wire/ABI types (reference
 * src/common_sp.h:7-20): `cmplx_u8` is two bytes, I then Q, exactly as the
 * RTL2832U delivers them; `cmplx_s32` is eight bytes, re then im, and also
 * addressable as one int64 (`bulk`) for whole-value copies.
 * The include guard matches the reference header on purpose: a translation
 * unit that already pulled in the reference's own common_sp.h keeps it.
 *
 * The accessor macros and atan2_approx carry the reference's names and
 * semantics (src/common_sp.h:22-38 and :40-76), so a reference unit that
 * stays in the build and needs them (src/audio_main.c does) compiles with
 * this directory first on the include path; oracle/Makefile (ref_dropin)
 * compiles the reference's audio_main.c, resample.c and rf_decimator.c
 * against these headers as the check.  The engine itself uses only the
 * layouts.
 */
#ifndef COMMON_SP_H
#define COMMON_SP_H

#include <stdint.h>

typedef struct {
    uint8_t re;     /* I, offset-binary: 128 is zero */
    uint8_t im;     /* Q */
} cmplx_u8;

typedef union {
    int64_t bulk;
    struct p {
        int32_t re;
        int32_t im;
    } p;
} cmplx_s32;

/* Statement macros, as in the reference (used as `macro(...);`).  dst / a / b /
 * result / c are lvalue expressions of the types above. */
#define set_cmplx_u8(dst, real, imag) \
    do { (dst).re = (real); (dst).im = (imag); } while (0)
#define set_cmplx_s32(dst, src) \
    do { (dst).bulk = (src).bulk; } while (0)
/* the centring step of the CIC: transform is -128 (src/resample.c:24-25) */
#define set_cmplx_s32_cmplx_u8(dst, src, transform) \
    do { (dst).p.re = (transform) + (src).re; (dst).p.im = (transform) + (src).im; } while (0)
#define add_cmplx_s32(a, b, result) \
    do { (result).p.re = (a).p.re + (b).p.re; (result).p.im = (a).p.im + (b).p.im; } while (0)
#define sub_cmplx_s32(a, b, result) \
    do { (result).p.re = (a).p.re - (b).p.re; (result).p.im = (a).p.im - (b).p.im; } while (0)
#define real_cmplx_s32(c) ((c).p.re)
#define imag_cmplx_s32(c) ((c).p.im)
#define real_cmplx_u8(c) ((c).re)
#define imag_cmplx_u8(c) ((c).im)

#include <math.h>

/* Rational approximation of atan2 used by the FM demodulator (reference
 * src/common_sp.h:40-76): with q = y/x, q/(1 + 0.28 q^2) inside +-45 degrees of
 * the x axis and pi/2 - q/(q^2 + 0.28) outside, moved to the right quadrant.
 * The quadrant corrections add M_PI in double and round once on return, as the
 * reference's expressions do; rtlws_fm_demod (rtlws_hip.h) evaluates exactly
 * this on the device. */
#define RTLWS_M_PI 3.14159265358979323846   /* M_PI, which strict C99 <math.h> does not provide */
static inline float atan2_approx(float y, float x)
{
    const float half_pi = (float)(RTLWS_M_PI / 2);
    float q, a;
    if (x == 0)
        return y > 0.0f ? half_pi : (y == 0 ? 0.0f : -half_pi);
    q = y / x;
    if (fabs(q) < 1.0f) {
        a = q / (1.0f + 0.28f * q * q);
        if (!(x < 0))
            return a;
        return (float)(y < 0.0f ? a - RTLWS_M_PI : a + RTLWS_M_PI);
    }
    a = half_pi - q / (q * q + 0.28f);
    return y < 0.0f ? (float)(a - RTLWS_M_PI) : a;
}

#endif /* COMMON_SP_H */
