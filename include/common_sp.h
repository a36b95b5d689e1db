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
 * The reference's accessor macros (src/common_sp.h:22-38) and atan2_approx
 * (src/common_sp.h:40-76, audio path) are deliberately not repeated here:
 * the engine only needs the layouts.
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

#endif /* COMMON_SP_H */
