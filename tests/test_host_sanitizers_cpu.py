"""The threaded host layer under sanitizers, on the CPU.

tests/fake_hip/ holds a TEST-ONLY implementation of include/rtlws_hip.h (streams = worker threads with random
delays, events that complete when their queue reaches them, the f64 oracle as the transform) and a stress
driver; build.sh compiles the PRODUCT's host sources (rtl-ws_amd/host/*.c, unmodified) against them with
-fsanitize=thread or -fsanitize=address,undefined.  What then runs is the real ring walk / condition variables
of stream_gpu.c (ring full + non-blocking drop, close with chunks in flight, eight producers on eight streams,
four producers on one stream, an injected launch failure), the shard threads and mailbox of multi_batch.c, the
two-slot hand-off of cbb_gpu.c with rf_decimator_set_parameters called from a second thread (reference
src/main.c:154), and spectrum.h from two threads -- with every row checked against a direct oracle call.
Zero sanitizer reports and zero failed checks, or the test fails.  Nothing of tests/fake_hip/ is ever built
into rtl-ws_amd/lib.  The last test runs tests/tools/asan_host_cpu.sh (the device-less failure paths of the
real library under ASan + UBSan), which used to be run by hand."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path, sanitizer, env_extra):
    out = str(tmp_path / "hs")
    b = subprocess.run(["bash", os.path.join(ROOT, "tests", "fake_hip", "build.sh"), out, sanitizer],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RTLWS_STREAM_QUEUES", "RTLWS_STREAM_ZEROCOPY_OUT", "RTLWS_STREAM_ZEROCOPY_IN", "RTLWS_CBB_ALL_FRAMES")}
    env.update(env_extra)
    r = subprocess.run([os.path.join(out, "host_stress")], capture_output=True, text=True, timeout=600, env=env)
    return r


@pytest.mark.parametrize("zero_copy_out", ["1", "0"])
def test_threaded_host_layer_under_thread_sanitizer(tmp_path, zero_copy_out):
    r = _build_and_run(tmp_path, "thread", {"TSAN_OPTIONS": "halt_on_error=0 exitcode=66",
                                            "RTLWS_STREAM_ZEROCOPY_OUT": zero_copy_out})
    text = r.stdout + r.stderr
    assert "ThreadSanitizer" not in text, text[-6000:]
    assert r.returncode == 0 and "host_stress: 0 failure(s)" in r.stdout, text[-3000:]


def test_threaded_host_layer_under_address_and_ub_sanitizers(tmp_path):
    r = _build_and_run(tmp_path, "address,undefined",
                       {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                        "RTLWS_CBB_ALL_FRAMES": "2", "FAKE_HIP_DEVICES": "8"})   # the Welch mode of cbb_gpu.c, eight fake devices
    text = r.stdout + r.stderr
    assert not re.search(r"AddressSanitizer|LeakSanitizer|runtime error", text), text[-6000:]
    assert r.returncode == 0 and "host_stress: 0 failure(s)" in r.stdout, text[-3000:]


def test_the_fake_shim_is_not_part_of_the_product():
    lib = os.path.join(ROOT, "rtl-ws_amd", "lib")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rtl-ws_amd")):
        for f in files:
            if f.endswith((".c", ".h", ".hip", ".cpp", ".py")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "fake_hip" not in text and "fake_rtlws" not in text, os.path.join(dirpath, f)
    if os.path.isdir(lib):
        assert not [f for f in os.listdir(lib) if "fake" in f or "host_stress" in f]


def test_device_less_failure_paths_under_asan(built):
    """tests/tools/asan_host_cpu.sh: librtlws_amd built with ASan + UBSan and driven through the CPU tests
    (no device: every entry point takes its failure path, partially built handles are torn down)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the script walks the no-device paths; this host has a device")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "tools", "asan_host_cpu.sh")], capture_output=True, text=True, timeout=900)
    text = r.stdout + r.stderr
    assert r.returncode == 0 and not re.search(r"AddressSanitizer|runtime error", text), text[-5000:]
