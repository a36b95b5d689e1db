import numpy as np

# Error metric: per bin |got-ref| / max(|ref|, eps * max_bin(ref_row)).
# f32 cannot hold 1e-4 RELATIVE on bins arbitrarily far below the row maximum:
# next to a strong tone the partial sums reach ~N*amplitude and their f32 ulp
# sets an absolute error floor about 6e-8 of the peak amplitude (DESIGN.md,
# "Error budget").  So a single frame (K=1) is judged with eps = 1e-5 (bins
# more than 50 dB below the row maximum are held to the absolute error of a
# bin at that level); K-frame averages (K >= 6, the product's case) never have
# near-empty bins and are judged with the strict eps = 1e-9.
EPS_K1 = 1e-5
EPS_STRICT = 1e-9
TOL = 1e-4
# The reference-API paths (spectrum.h, cbb_main.h) and rtlws_spectra_batch_f64
# compute in double like the reference: they are held to the strict metric
# (eps = 1e-9) at every K, four orders inside north_star's 1e-4.
TOL_F64 = 1e-10
# Guard rails for the f32 batch kernel under the STRICT metric at K = 1 (what
# DESIGN.md "Error budget" reports: max 1.3e-3..3.5e-3, p99.9 3e-5..1.5e-4): a
# regression in the f32 arithmetic must not hide behind the relaxed floor.
STRICT_K1_P999 = 1e-4
STRICT_K1_MAX = 5e-3


def eps_for(K):
    return EPS_K1 if K == 1 else EPS_STRICT


def rel_err(got, ref, eps=EPS_STRICT):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    floor = eps * np.abs(ref).max(axis=-1, keepdims=True)
    floor = np.where(floor > 0, floor, 1.0)       # all-zero rows: absolute error
    return np.abs(got - ref) / np.maximum(np.abs(ref), floor)


def strict_stats(got, ref):
    """(max, 99.9th percentile) of the strict-floor (eps = 1e-9) relative error."""
    e = rel_err(got, ref, EPS_STRICT)
    return float(e.max()), float(np.percentile(e, 99.9))
