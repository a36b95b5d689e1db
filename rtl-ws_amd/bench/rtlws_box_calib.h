/* rtlws_box_calib.h -- bench.py's box calibration (lib/librtlws_bench.so; measurement plumbing, not product API).
 * Runs a fixed v_fma_f64 instruction stream on every SIMD of `device` for `seconds` (0 < seconds <= 5) and
 * reports the shader clock the governor settled at.  0 on success. */
#ifndef RTLWS_BOX_CALIB_H
#define RTLWS_BOX_CALIB_H
#ifdef __cplusplus
extern "C" {
#endif
struct rtlws_box_calib {
    double sclk_ghz;           /* median over the wavefronts of the last launch: d(s_memtime) / d(s_memrealtime) */
    double gpu_seconds;        /* HIP events around all launches */
    double wall_seconds;
    double wave_instructions;  /* wave64 v_fma_f64 issued in all */
    long launches;
    int simds;
};
int rtlws_box_calib_run(int device, double seconds, struct rtlws_box_calib* res);
#ifdef __cplusplus
}
#endif
#endif
