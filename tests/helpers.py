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


def eps_for(K):
    return EPS_K1 if K == 1 else EPS_STRICT


def rel_err(got, ref, eps=EPS_STRICT):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    floor = eps * np.abs(ref).max(axis=-1, keepdims=True)
    floor = np.where(floor > 0, floor, 1.0)       # all-zero rows: absolute error
    return np.abs(got - ref) / np.maximum(np.abs(ref), floor)
