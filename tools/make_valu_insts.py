#!/usr/bin/env python3
"""profiles/valu_insts.json -- what bench.py's roofline.valu_issue_frac is computed from.

For a workload: the vector-issue cycles one launch needs on a SIMD,
    issue_cycles_per_launch = sum over instruction classes of (wave64 instructions x issue cycles),
from rocprofv3's SQ_INSTS_VALU per launch (profiles/r03_<workload>_pmc.json) split by class with
the kernel's own ISA (hipcc -S; tools/isa_hist.py): plain f32 / integer VALU 2 cycles, conversions
and SDWA 4, every f64 instruction 4 (tools/valubench.hip, measured on gfx950) -- and the shader
clock the workload runs at: measured IN the kernel where the kernel has the stamped diagnostic
build (tools/inkernel_clock.py: d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups
after 2.5 s of back-to-back launches; profiles/r03_inkernel_clock.json), rocm-smi's sclk under
sustained load otherwise (profiles/r03_power_clocks.txt; it can read up to ~10 % high).
bench.py divides by the SIMD-cycles its own launch lasted: 4 SIMDs x CUs x launch time x sclk.

usage: make_valu_insts.py profiles/valu_insts.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(workload):
    """the newest committed counter file of the workload (the kernel a line is priced with)"""
    for tag in ("r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (tag, workload))
        if os.path.exists(f):
            return json.load(open(f))["counters"]
    raise FileNotFoundError(workload)


def entry(workload, frames, four_cycle_per_frame, sclk_ghz, note):
    """four_cycle_per_frame: wave64 instructions per frame that issue in 4 cycles instead of 2."""
    insts = pmc(workload)["SQ_INSTS_VALU"]["mean_per_launch"]
    four = four_cycle_per_frame * frames
    cycles = 2.0 * (insts - four) + 4.0 * four
    return {"sq_insts_valu_per_launch": insts, "four_cycle_insts_per_launch": four,
            "issue_cycles_per_launch": cycles, "sclk_ghz_under_load": sclk_ghz, "frames_per_launch": frames,
            "note": note}


def main():
    out = {
        # 4 wavefronts x 32 v_cvt_f32_ubyteN per frame
        "hann_4096pt_k8_db": entry("hann_4096pt_k8_db", 16384, 128, 2.188,
                                   "spectra_fused<4096, u8, Hann, dB, K>1>: 128 byte conversions per frame at 4 cycles; "
                                   "in-kernel clock 2.188 GHz median (p10-p90 2.163-2.217; profiles/r03_inkernel_clock.json); "
                                   "rocm-smi 2.194-2.202 GHz at 1 354-1 372 W"),
        # 1 wavefront x 32 conversions per frame
        "batched_1024pt_64k_frames": entry("batched_1024pt_64k_frames", 65536, 32, 2.007,
                                           "spectra_fused<1024, u8, rect, sum, K=1>: 32 byte conversions per frame at 4 "
                                           "cycles; in-kernel clock 2.007 GHz median (p10-p90 1.936-2.078; another box's "
                                           "rocm-smi: 1.884-1.892 GHz at 1 394-1 402 W, the cap)"),
        # round 4, spectrum_f64_1024x.hip (one wavefront per frame): 436 f64 add / mul / fma + 32 v_cvt_f64_i32 at 4
        # cycles (+ 16 v_cvt_f32_f64 with f32 rows); the ~115 integer / cross-lane / address instructions at 2
        "batched_1024pt_64k_frames_f64": entry("batched_1024pt_64k_frames_f64", 65536, 436 + 32, 2.03,
                                               "spectra_f64_1024x<sum, K=1, f64 rows>: 436 f64 add / mul / fma + 32 "
                                               "v_cvt_f64_i32 per frame at 4 cycles; rocm-smi sclk 2.02-2.05 GHz at "
                                               "1 375-1 382 W (bench.py measures its own run's clock: roofline.sclk_ghz)"),
        "batched_1024pt_64k_frames_f64c_f32o": entry("batched_1024pt_64k_frames_f64c_f32o", 65536, 436 + 32 + 16, 2.01,
                                                     "spectra_f64_1024x<sum, K=1, f32 rows>: 436 f64 add / mul / fma + 32 "
                                                     "v_cvt_f64_i32 + 16 v_cvt_f32_f64 per frame at 4 cycles; rocm-smi "
                                                     "sclk 2.00-2.02 GHz at 1 380-1 386 W (profiles/r04_power_clocks.txt)"),
    }
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, "%.4g issue cycles per launch" % v["issue_cycles_per_launch"])


if __name__ == "__main__":
    main()
