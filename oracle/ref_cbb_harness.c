/* ref_cbb_harness.c -- TEST INFRASTRUCTURE.  main() for oracle/_ref/rtlws_ref_cbb_on_gpu:
 * the reference's own, unmodified src/cbb_main.c + src/signal_source.c object
 * code (compiled where they lie, oracle/Makefile) linked against THIS repo's
 * librtlws_amd.so -- i.e. the reference's control plane calling spectrum_alloc /
 * spectrum_add_cmplx_u8 / rf_decimator_* of the GPU engine through its own
 * headers -- fed by the synthetic rtl_sensor (rtl-ws_amd/host/synth_sensor.c).
 *
 * Prints, for the first spectrum the reference publishes, the 1024 payload bytes
 * of cbb_get_spectrum_payload for gains 0, 15, -25 as hex lines.  The dB/clamp
 * arithmetic that produced them is the reference's (f64, src/cbb_main.c:121-130);
 * the power sums under it came from the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

/* the reference's own header (found through -I/root/reference/src) */
#include "cbb_main.h"

void cbb_init(int decimated_bw_target_hz);   /* src/cbb_main.c:72 (its header omits the int) */

int main(void)
{
    static char buf[8192];
    const int gains[3] = {0, 15, -25};
    struct timespec nap = {0, 2000000};
    int waited = 0, g, i;

    cbb_init(192000);
    while (!cbb_new_spectrum_available() && waited < 5000) { nanosleep(&nap, NULL); waited += 2; }
    if (!cbb_new_spectrum_available()) { fprintf(stderr, "no spectrum within 5 s\n"); return 2; }
    for (g = 0; g < 3; g++) {
        const int n = cbb_get_spectrum_payload(buf, sizeof buf, gains[g]);
        printf("gain %d len %d ", gains[g], n);
        for (i = 0; i < n; i++) printf("%02x", (unsigned char)buf[i]);
        printf("\n");
    }
    cbb_close();
    return 0;
}
