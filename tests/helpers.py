import numpy as np

EPS_FLOOR = 1e-9      # error metric floor relative to the row maximum


def rel_err(got, ref, eps=EPS_FLOOR):
    """Per-bin |got-ref| / max(|ref|, eps*max_bin(ref)), rows = spectra."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    floor = eps * np.abs(ref).max(axis=-1, keepdims=True)
    floor = np.where(floor > 0, floor, 1.0)       # all-zero rows: absolute error
    return np.abs(got - ref) / np.maximum(np.abs(ref), floor)
