"""CPU suite: the C-ABI libraries load and export every symbol include/*.h
declares (no compute calls: there is no GPU here), and fail loudly without one."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def _declared_functions(header):
    txt = open(os.path.join(INCLUDE, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    txt = re.sub(r'extern\s+"C"\s*\{', "", txt)            # keep what the linkage block holds
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    for _ in range(3):                                   # drop struct/enum bodies
        txt = re.sub(r"\{[^{}]*\}", "", txt)
    txt = re.sub(r"typedef[^;]*;", "", txt)              # function-pointer typedefs
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", txt)
    return [n for n in names if n not in ("defined", "sizeof")]


# which headers each library implements -- and, since round 6, ALL it exports
LIB_HEADERS = {
    "HIP_LIB": ("rtlws_hip.h",),
    "AMD_LIB": ("spectrum.h", "resample.h", "rf_decimator.h", "rtlws_stream.h", "audio_main.h", "rtlws_host.h",
                "rtlws_multi.h", "rtlws_topo.h"),
    "CBB_LIB": ("cbb_main.h", "rtlws_cbb.h"),
    "SYNTH_LIB": ("rtl_sensor.h", "signal_source.h"),
}


def _declared_by_lib():
    out = {}
    for lib, headers in LIB_HEADERS.items():
        names = []
        for h in headers:
            fns = _declared_functions(h)
            assert fns, h
            names += fns
        out[lib] = names
    return out


def _exported(path):
    """Defined dynamic symbols of a shared object (nm -D --defined-only)."""
    import subprocess
    txt = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1].split("@")[0] for ln in txt.splitlines() if ln.strip()}


def test_every_declared_symbol_is_exported(built):
    hip = ctypes.CDLL(built.HIP_LIB, mode=ctypes.RTLD_GLOBAL)
    amd = ctypes.CDLL(built.AMD_LIB)
    declared = _declared_by_lib()
    assert len(declared["HIP_LIB"]) >= 20
    for name in declared["HIP_LIB"]:
        assert hasattr(hip, name), "librtlws_hip.so lacks " + name
    for name in declared["AMD_LIB"]:
        assert hasattr(amd, name), "librtlws_amd.so lacks " + name
    # boundary #2 and the synthetic seam live in their own libraries
    synth = ctypes.CDLL(built.SYNTH_LIB, mode=ctypes.RTLD_GLOBAL)
    cbb = ctypes.CDLL(built.CBB_LIB)
    for name in declared["CBB_LIB"]:
        assert hasattr(cbb, name), "librtlws_cbb.so lacks " + name
    for name in declared["SYNTH_LIB"]:
        assert hasattr(synth, name), "librtlws_synth.so lacks " + name
    # the binding's own lists agree with the headers
    assert set(built.HIP_SYMBOLS) == set(declared["HIP_LIB"])
    assert (set(built.AMD_SYMBOLS) | set(built.STREAM_SYMBOLS) | set(built.AUDIO_SYMBOLS) | set(built.HOST_SYMBOLS)
            | set(built.MULTI_SYMBOLS) | set(built.TOPO_SYMBOLS) == set(declared["AMD_LIB"]))
    assert set(built.CBB_SYMBOLS) == set(declared["CBB_LIB"])
    assert set(built.SYNTH_SYMBOLS) == set(declared["SYNTH_LIB"])


def test_every_exported_symbol_is_declared(built):
    """The converse: a library a C server links exports its headers and NOTHING else -- no
    kernel stubs, launchers, std:: instantiations or cross-file helpers (librtlws_hip.so used to leak 1 360 of
    them).  -fvisibility=hidden + the version scripts of rtl-ws_amd/exports/."""
    declared = _declared_by_lib()
    for lib, names in declared.items():
        got = _exported(getattr(built, lib))
        assert got == set(names), "%s: exported but not declared %s; declared but not exported %s" % (
            lib, sorted(got - set(names)), sorted(set(names) - got))
    assert len(_exported(built.HIP_LIB)) == len(built.HIP_SYMBOLS)


def test_hooks_patch_applies():
    """tools/variants/csrc_hooks.patch (the measurement hooks, kept OUT of the product sources) still applies to
    rtl-ws_amd/csrc, and the product sources carry none of its switches."""
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "rtl-ws_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "csrc")
        shutil.copytree(src, dst)
        r = subprocess.run(["patch", "-s", "-p3", "--dry-run", "-i", os.path.join(ROOT, "tools", "variants", "csrc_hooks.patch")],
                           cwd=dst, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    for f in os.listdir(src):
        txt = open(os.path.join(src, f)).read()
        for sw in ("RTLWS_F64_ABL_", "RTLWS_ABL_", "RTLWS_X_STAMP", "RTLWS_STAMP", "RTLWS_NO_NT", "RTLWS_F64_PLAIN_STORE"):
            assert sw not in txt, "%s still carries %s" % (f, sw)
        # no preprocessor conditional other than include guards and the per-size compile (-DRTLWS_N)
        for m in re.finditer(r"^\s*#\s*(if|ifdef|ifndef|elif)\b(.*)$", txt, flags=re.M):
            arg = m.group(2).strip()
            assert arg.endswith("_H") or arg == "RTLWS_N", "%s: conditional on %r" % (f, arg)


def test_abi_struct_layouts(built):
    assert ctypes.sizeof(built.SpectraDesc) == 32
    assert ctypes.sizeof(built.CicDelayLine) == 16        # two cmplx_s32, src/resample.h:8-12
    assert ctypes.sizeof(built.CmplxS32) == 8             # src/common_sp.h:13-20


def test_descriptor_validation_needs_no_gpu(built):
    L = built.hip_lib()
    kind = lambda **kw: L.rtlws_spectra_kernel_kind(ctypes.byref(built.make_desc(**kw)))
    assert kind(n_fft=1024) == 1 and kind(n_fft=2048) == 1 and kind(n_fft=4096) == 1
    assert kind(n_fft=1000) == 2 and kind(n_fft=512) == 2
    assert kind(n_fft=1) == 0 and kind(n_fft=1024, k_avg=0) == 0
    assert kind(n_fft=1024, input="cs32", cic_r=8) == 0
    assert kind(n_fft=100000) == 0


def test_no_gpu_fails_loudly(built):
    """Without a HIP device the product path must refuse, never fall back."""
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        built.Engine(0)
    with pytest.raises(RuntimeError):
        built.Spectrum(1024)
    rc, _, _ = built.cic_decimate(8, np.zeros((64, 2), dtype=np.uint8))
    assert rc == -3
    A = built.amd_lib()
    A.rtlws_stream_open.restype = ctypes.c_void_p
    A.rtlws_stream_open.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    d = built.make_desc(1024)
    cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_double, ctypes.c_void_p)(lambda *a: None)
    assert not A.rtlws_stream_open(0, ctypes.byref(d), 128, 3, ctypes.cast(cb, ctypes.c_void_p), None)
    with pytest.raises(RuntimeError):
        built.MultiBatch(d, 1024, device_ids=[0, 0])


def test_void_entry_points_do_not_kill_the_host_without_a_gpu(built):
    """halfband_decimate / audio_* return void (src/resample.h:17, src/audio_main.c:53,106): with no
    device they must not compute on the CPU -- and must not abort() the server they were dropped
    into either: zeros out, delay line advanced as src/resample.c:66 would, failure recorded
    (include/rtlws_host.h)."""
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    built.host_error_clear()
    assert built.host_error() == (0, "")
    x = np.arange(1, 41, dtype=np.float32)
    delay = np.full(10, 7.0, dtype=np.float32)
    y = built.halfband_decimate(x, delay)
    assert y.shape == (20,) and not y.any()                       # silence, not garbage, not a CPU filter
    assert np.array_equal(delay, x[-10:])                         # the delay line still advances
    short = np.array([1.0, 2.0], dtype=np.float32)
    d2 = np.arange(10, dtype=np.float32)
    built.halfband_decimate(short, d2)
    assert np.array_equal(d2, np.array([2, 3, 4, 5, 6, 7, 8, 9, 1, 2], dtype=np.float32))
    n, msg = built.host_error()
    assert n == 2 and msg.startswith("halfband_decimate: ")       # sticky: the FIRST message, every failure counted
    A = built.amd_lib()
    A.audio_init()
    buf = np.zeros((64, 2), dtype=np.int32)
    A.audio_fm_demodulator(buf.ctypes.data_as(ctypes.c_void_p), 64)
    assert A.audio_new_audio_available() == 0                     # nothing queued
    A.audio_close()
    n2, msg2 = built.host_error()
    assert n2 == 4 and msg2 == msg
    built.host_error_clear()
    assert built.host_error() == (0, "")


def _own_call_graph(lib):
    """{function: set(call targets)} of a shared library's own .text (objdump -d; direct calls and
    tail jumps to named symbols; PLT stubs keep their `name@plt` spelling)."""
    import subprocess
    txt = subprocess.run(["objdump", "-d", "--no-show-raw-insn", "-j", ".text", lib], capture_output=True, text=True,
                         check=True).stdout
    graph, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = m.group(1)
            graph[cur] = set()
            continue
        if cur is None:
            continue
        m = re.search(r"\b(?:call|jmp|j[a-z]+)\s+[0-9a-f]+ <([^>+]+)(?:\+0x[0-9a-f]+)?>", line)
        if m and m.group(1) != cur:
            graph[cur].add(m.group(1))
    return graph


def _reaches(graph, start, target):
    seen, todo = set(), [start]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for g in graph.get(f, ()):
            if g == target:
                return True
            todo.append(g)
    return False


def test_no_launch_path_reads_the_environment(built):
    """RTLWS_V2 / RTLWS_*BLOCKS_PER_CU / RTLWS_F64_FUSED / RTLWS_CIC_* were read
    with getenv() on every launch.  They are now read ONCE, in rtlws_engine_create: in the built
    library no call path from a launch entry point reaches getenv (call graph of the library's own
    code, from its disassembly), while the one from rtlws_engine_create does (positive control)."""
    g = _own_call_graph(built.HIP_LIB)
    assert "rtlws_spectra_batch" in g and "rtlws_engine_create" in g
    assert _reaches(g, "rtlws_engine_create", "getenv@plt")
    for entry in ("rtlws_spectra_batch", "rtlws_spectra_batch_f64", "rtlws_cic_block_sums", "rtlws_halfband",
                  "rtlws_fm_demod", "rtlws_payload_from_sums", "rtlws_payload_from_sums_f64",
                  "rtlws_welch_accumulate_f64", "rtlws_welch_finish_f64", "rtlws_spectra_grid",
                  "rtlws_copy_h2d", "rtlws_copy_d2h", "rtlws_stream_sync", "rtlws_event_record"):
        assert entry in g and g[entry], entry                     # the function exists and calls something
        assert not _reaches(g, entry, "getenv@plt"), entry + " can reach getenv()"
    # ... and in the C host layer the drop-in calls do not either (spectrum_alloc / the lazily
    # created context read their switches once)
    h = _own_call_graph(built.AMD_LIB)
    assert _reaches(h, "spectrum_alloc", "getenv@plt")
    for entry in ("spectrum_add_cmplx_u8", "spectrum_add_cmplx_s32", "spectrum_add_real_f32", "rtlws_stream_push"):
        assert entry in h and not _reaches(h, entry, "getenv@plt"), entry


def test_product_never_touches_the_oracle():
    """No file under rtl-ws_amd/ or include/ mentions the oracle library."""
    bad = []
    for base in ("rtl-ws_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "build" in dp or dp.endswith("lib"):
                continue
            for f in fn:
                if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pyoracle|rtlws_oracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_no_kernel_spills_registers(built):
    """Code-object metadata of librtlws_hip.so (NT_AMDGPU_METADATA, read with
    llvm-readelf; no GPU needed): no kernel of the product may spill VGPRs or
    SGPRs or use scratch.  A spilling instantiation writes its registers to HBM
    on every frame (round 1: 1.33x write amplification on the 4096-point Hann
    kernel), so the occupancy each instantiation is built for
    (rtlws_internal.h, fused_waves_per_simd) must leave it enough registers."""
    from rtlws import codeobj
    ks = codeobj.kernels(built.HIP_LIB)
    fused = [k for k in ks if "spectra_fused" in k["name"]]
    assert len(fused) >= 200                       # every (N, input, window, output, K==1) instantiation
    names = " ".join(k.get("demangled", "") for k in ks)
    for want in ("spectra_f64", "spectra_f64_fused", "spectra_fused_v2", "spectra_direct", "cic8_kernel", "cicr_kernel", "halfband_kernel",
                 "fm_demod_kernel", "payload_kernel", "payload_f64_kernel"):
        assert want in names, want
    # (the f64 kernel parks scalar lane masks of its 32 unrolled slots in a VGPR --
    # an SGPR "spill" that never leaves the register file; it is not a throughput kernel)
    bad = [(k.get("demangled", k["name"]), k["vgpr_spill_count"], k.get("sgpr_spill_count", 0),
            k["private_segment_fixed_size"]) for k in ks
           if k["vgpr_spill_count"] or k["private_segment_fixed_size"]
           or (k.get("sgpr_spill_count", 0) and "spectra_f64" not in k["name"])]
    assert not bad, bad
    # the headline kernel keeps its 4 waves/SIMD (<= 128 VGPRs)
    head = [k for k in fused if "spectra_fused<1024, 0, false, 0, true>" in k.get("demangled", "")]
    assert len(head) == 1 and head[0]["vgpr_count"] <= 128
    # the two-wavefronts-per-SIMD kernels stay inside their 256 registers without scratch
    for k in ks:
        if "spectra_fused_v2" in k["name"] or "spectra_f64_fused" in k["name"]:
            assert k["vgpr_count"] <= 256 and not k.get("sgpr_spill_count", 0), k.get("demangled")


def test_v2_pass3_lane_map_is_a_conflict_free_permutation():
    """spectrum_fused_v2.hip pair_of_lane(): a permutation of the 128 pairs, and with the
    transposition-2 strides (290, 18) every ds_read_b128 lane group covers the 64 banks once
    (tools/lds_sim.py's bank model, MI355X_MICROARCH.md LDS table)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    argv = sys.argv
    sys.argv = ["lds_sim"]
    try:
        import lds_sim
    finally:
        sys.argv = argv

    def pair_of_lane(t):
        l = t & 31
        s = l if l < 4 else l + 12 if l < 12 else l - 8 if l < 16 else l + 8 if l < 20 else l - 12 if l < 28 else l
        return (t & ~31) | s

    assert sorted(pair_of_lane(t) for t in range(128)) == list(range(128))
    assert pair_of_lane(0) == 0 and pair_of_lane(127) == 127          # the DC-slot hand-off relies on both
    total = 0
    for wave in range(2):
        lanes = [64 * wave + l for l in range(64)]
        for h in range(2):
            for i in range(8):
                addr = [(pair_of_lane(t) >> 3) * 290 + (2 * (pair_of_lane(t) & 7) + h) * 18 + 2 * i for t in lanes]
                assert all(a % 2 == 0 for a in addr)                  # 16-byte aligned
                total += lds_sim.cycles(addr, lds_sim.R128, 4, 64)
    assert total == 2 * 16 * 4                                        # 4 LDS cycles per instruction: no conflict
    res = lds_sim.analyse_v2(4096, 272, 290, 18, verbose=False)
    assert all(r[:3] == (128, 64, 128) for r in res.values())         # writes and the pass-2 reads as well

    # N = 2048 (one wavefront per frame): 64 pairs, q2 = pair / 4, strides (146, 18)
    def pair_2048(t):
        l = t & 31
        g1 = (4 <= l < 12) or (16 <= l < 20) or l >= 28
        idx = (l - 4 if l < 12 else l - 8 if l < 20 else l - 16) if g1 else (l if l < 4 else l - 8 if l < 16 else l - 12)
        k = 2 * (t >> 5) + (1 if g1 else 0)
        return 8 * k + idx if idx < 8 else 32 + 8 * k + (idx - 8)

    assert sorted(pair_2048(t) for t in range(64)) == list(range(64))
    assert pair_2048(0) == 0 and pair_2048(63) == 63
    total = 0
    for h in range(2):
        for i in range(8):
            addr = [((2 * pair_2048(t) + h) // 8) * 146 + ((2 * pair_2048(t) + h) % 8) * 18 + 2 * i for t in range(64)]
            assert all(a % 2 == 0 for a in addr)
            total += lds_sim.cycles(addr, lds_sim.R128, 4, 64)
    assert total == 16 * 4
    res = lds_sim.analyse_v2(2048, 136, 146, 18, verbose=False)
    assert all(r[:3] == (128, 64, 128) for r in res.values())


def test_f64_fused_lds_layout_is_conflict_free():
    """spectrum_f64_fused.hip: rows of 64 double2 padded to 68, reader groups of 16 padded to 17."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    argv = sys.argv
    sys.argv = ["lds_sim"]
    try:
        import lds_sim
    finally:
        sys.argv = argv
    assert lds_sim.analyse_f64(68, 17, verbose=False) == (128, 64, 128, 64)
    assert lds_sim.analyse_f64(64, 16, verbose=False)[1] > 64          # the unpadded layout conflicts
    # the multi-wavefront sizes: rows of 17*R3, groups of 17 -- every access pattern at its ideal
    for N in (2048, 4096):
        per_wave = lds_sim.analyse_f64_n(N, verbose=False)
        assert all(w == (128, 64, 128, 64) for w in per_wave), (N, per_wave)


def test_cbb_init_without_a_gpu_leaves_the_server_alive(built):
    """cbb_init returns void (src/cbb_main.c:72).  Without a device it used to abort() the process
    it was linked into; now the spectrum side is inert -- the sensor keeps being drained, no spectrum
    is ever announced, the payload call returns 0 bytes -- and the failure is on record.  Run in a
    child process (the synthetic sensor owns a thread)."""
    import subprocess
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    code = (
        "import sys, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import rtlws\n"
        "L = rtlws.cbb_lib(); rtlws.host_error_clear(); L.cbb_init(192000); time.sleep(0.4)\n"
        "n, msg = rtlws.host_error()\n"
        "assert L.cbb_new_spectrum_available() == 0 and len(rtlws.cbb_payload(0)) == 0\n"
        "assert L.rtlws_cbb_samples_seen() > 0 and n == 1 and msg.startswith('cbb_init: '), (n, msg)\n"
        "L.cbb_close(); print('alive')\n" % (ROOT, os.path.join(ROOT, "rtl-ws_amd")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip().endswith("alive"), out.stderr[-2000:]


def test_public_headers_are_strict_c99(tmp_path):
    """The boundary is a C-ABI: every header under include/ compiles on its own as strict C99
    (no C++-isms, no torch / HIP types in a signature)."""
    import subprocess
    for h in sorted(os.listdir(INCLUDE)):
        if not h.endswith(".h"):
            continue
        src = tmp_path / ("t_" + h[:-2] + ".c")
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % h)
        out = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", INCLUDE, "-c", str(src),
                              "-o", os.devnull], capture_output=True, text=True)
        assert out.returncode == 0, (h, out.stderr)
        code = re.sub(r"/\*.*?\*/", "", open(os.path.join(INCLUDE, h)).read(), flags=re.S)
        assert "hipStream_t" not in code and "hipError_t" not in code and "at::" not in code, h
