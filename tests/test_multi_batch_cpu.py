"""rtlws_multi.h on CPU: the K-aligned contiguous partition of a batch over the devices of a node
("contiguous frame ranges [g*B/G, (g+1)*B/G) aligned to K", one host thread and one
engine per device, no collective), through the C-ABI and through the C driver's --plan-only.  The
GPU side is tests/test_multi_batch_gpu.py."""
import json
import os
import subprocess

import pytest


@pytest.mark.parametrize("B", [65536, 65530, 7, 0])
@pytest.mark.parametrize("K", [1, 6, 8])
@pytest.mark.parametrize("D", [1, 2, 8])
def test_partition_is_contiguous_k_aligned_and_complete(built, B, K, D):
    rows = B // K
    nxt, sizes = 0, []
    for g in range(D):
        rc, first, count = built.multi_partition(B, K, D, g)
        assert rc == 0
        assert first == nxt and first % K == 0 and count % K == 0 and count >= 0     # contiguous, ascending, whole K-groups
        assert first == K * (g * rows // D) and count == K * ((g + 1) * rows // D - g * rows // D)
        nxt = first + count
        sizes.append(count // K)
    assert nxt == rows * K                       # every whole K-group owned exactly once; B % K frames by nobody
    assert max(sizes) - min(sizes) <= 1          # balanced to one row
    if B == 65536 and K == 1:
        assert sizes == [65536 // D] * D         # configs[1]: 8 192 frames per GPU at 8 GPUs


def test_partition_rejects_bad_arguments(built):
    assert built.multi_partition(-1, 1, 2, 0)[0] == -1
    assert built.multi_partition(16, 0, 2, 0)[0] == -1
    assert built.multi_partition(16, 1, 0, 0)[0] == -1
    assert built.multi_partition(16, 1, 2, 2)[0] == -1
    assert built.multi_partition(16, 1, 2, -1)[0] == -1
    assert built.multi_partition(1 << 40, 8, 8, 7) == (0, 7 * (1 << 37), 1 << 37)      # no 32-bit arithmetic inside


@pytest.mark.parametrize("B,K,D", [(65536, 1, 8), (65530, 6, 8), (65530, 8, 2), (65536, 8, 1)])
def test_driver_plan_only_needs_no_gpu(built, B, K, D):
    exe = os.path.join(built.LIB_DIR, "rtlws_multi_batch")
    out = subprocess.run([exe, "--plan-only", "--devices", str(D), "--frames", str(B), "--k", str(K)],
                         capture_output=True, text=True, timeout=30)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["plan_only"] and r["devices"] == D and r["frames_used"] == B - B % K
    want = [built.multi_partition(B, K, D, g)[1:] for g in range(D)]
    assert [(s["first_frame"], s["frames"]) for s in r["shards"]] == want
    assert [s["device"] for s in r["shards"]] == list(range(D))
    assert subprocess.run([exe, "--plan-only"], capture_output=True).returncode == 2          # needs --devices


def test_open_without_a_gpu_fails_loudly(built):
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        built.MultiBatch(built.make_desc(1024), 64)
    exe = os.path.join(built.LIB_DIR, "rtlws_multi_batch")
    out = subprocess.run([exe, "--frames", "64", "--launches", "1", "--warmup", "0"], capture_output=True, text=True)
    assert out.returncode == 2 and "no HIP device" in out.stderr
