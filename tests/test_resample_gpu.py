"""GPU parity for the decimators: bit-exact against the reference-object-code
goldens (tests/golden/*_ref.npz) and against the oracle on fresh inputs."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R", [8, 10, 12])
def test_cic_golden_chained_calls(built, R):
    g = golden("cic_ref.npz")
    src, cuts = g[f"R{R}_src"], g[f"R{R}_cuts"]
    st, outs = None, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, dst, st = built.cic_decimate(R, src[a:b], state=st)
        assert rc == 0
        outs.append(dst)
    assert np.array_equal(np.concatenate(outs), g[f"R{R}_dst"])
    assert np.array_equal(st, g[f"R{R}_state"])


def test_cic_golden_state_cases(built):
    g = golden("cic_ref.npz")
    rc, dst, st = built.cic_decimate(8, g["odd_src"], state=g["odd_state0"])
    assert rc == 0 and np.array_equal(dst, g["odd_dst"]) and np.array_equal(st, g["odd_state1"])
    rc, dst, st = built.cic_decimate(8, g["wrap_src"], state=g["wrap_state0"])
    assert rc == 0 and np.array_equal(dst, g["wrap_dst"]) and np.array_equal(st, g["wrap_state1"])
    rc, _, _ = built.cic_decimate(8, g["wrap_src"][:63], dst_len=8)
    assert rc == int(g["mismatch_rc"]) == -1


@pytest.mark.parametrize("R", [1, 2, 3, 7, 8, 10, 12, 16, 100])
def test_cic_vs_oracle(built, oracle, R):
    rng = np.random.default_rng(R)
    src = rng.integers(0, 256, size=(R * 1000 + 0, 2), dtype=np.uint8)
    rc, dst, st = built.cic_decimate(R, src)
    rc2, dst2, st2 = oracle.cic_decimate(R, src)
    assert rc == rc2 == 0
    assert np.array_equal(dst, dst2) and np.array_equal(st, st2)


def test_cic_large_block_sums(engine, oracle):
    """2^20 outputs at R=8 straight through the batch entry point."""
    rng = np.random.default_rng(0)
    n = 1 << 20
    src = rng.integers(0, 256, size=(n * 8, 2), dtype=np.uint8)
    d_src = engine.upload(src)
    d_dst = engine.alloc(n * 8)
    engine.cic_block_sums(8, d_src, n, d_dst)
    got = engine.download(d_dst, np.int32, (n, 2))
    want = (src.astype(np.int32) - 128).reshape(n, 8, 2).sum(axis=1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("R", [1, 2, 3, 5, 9, 10, 12, 16, 25, 63, 64, 65, 127, 128])
def test_cic_block_sums_every_shape(engine, R):
    """Batch entry point: lengths around the 64-output piece and around a
    wavefront's round of pieces, every alignment class of R (odd R has a
    half-wavefront tail copy and masked half dwords; R > 64 the larger slice)."""
    rng = np.random.default_rng(1000 + R)
    per_round = 64 * max(1, (8192 if R <= 64 else 16384) // (128 * R))
    for n in (1, 63, 64, 65, per_round - 1, per_round + 64, 4 * per_round * 5 + 17, 40000):
        src = rng.integers(0, 256, size=(n * R, 2), dtype=np.uint8)
        d_src = engine.upload(src)
        d_dst = engine.alloc(n * 8)
        engine.cic_block_sums(R, d_src, n, d_dst)
        got = engine.download(d_dst, np.int32, (n, 2))
        want = (src.astype(np.int32) - 128).reshape(n, R, 2).sum(axis=1)
        assert np.array_equal(got, want), (R, n)


def test_cic_empty(built):
    rc, dst, st = built.cic_decimate(8, np.zeros((0, 2), dtype=np.uint8), state=[5, 6, 7, 8])
    assert rc == 0 and dst.shape[0] == 0 and list(st) == [5, 6, 7, 8]


def test_halfband_golden(built):
    g = golden("halfband_ref.npz")
    delay = np.zeros(10, dtype=np.float32)
    y1 = built.halfband_decimate(g["x"][:200], delay)
    assert np.array_equal(delay, g["delay_mid"])
    y2 = built.halfband_decimate(g["x"][200:], delay)
    assert np.array_equal(y1, g["y1"]) and np.array_equal(y2, g["y2"])      # bit-exact f32
    assert np.array_equal(delay, g["delay_end"])


def test_halfband_vs_oracle(built, oracle):
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2 * 50000).astype(np.float32)
    d1 = rng.standard_normal(10).astype(np.float32)
    d2 = d1.copy()
    y = built.halfband_decimate(x, d1)
    y_ref = oracle.halfband_decimate(x, d2)
    assert np.array_equal(y, y_ref) and np.array_equal(d1, d2)


def test_rf_decimator_golden(built):
    g = golden("rfdec_ref.npz")
    d = built.RfDecimator()
    rcs = [d.set_parameters(float(g["fs"]), int(g["R"]))]
    src, cuts = g["src"], g["cuts"]
    for a, b in zip(cuts[:-1], cuts[1:]):
        rcs.append(d.decimate(src[a:b]))
    assert rcs == list(g["rcs"])
    assert len(d.blocks) == g["blocks"].shape[0]
    assert np.array_equal(np.stack(d.blocks), g["blocks"])
    d.free()


def test_rf_decimator_unconfigured_and_bad_params(built):
    d = built.RfDecimator()
    assert d.decimate(np.zeros((10, 2), dtype=np.uint8)) == -1
    assert d.set_parameters(0.0, 8) == -1
    assert d.set_parameters(48000.0, 0) == -1
    assert d.set_parameters(48000.0, 8) == 0
    d.free()
