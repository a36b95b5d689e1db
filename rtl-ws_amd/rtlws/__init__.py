"""ctypes front end of the product libraries (rtl-ws_amd/lib/*.so).

This is plumbing for tests/, bench.py and __graft_entry__: it loads the C-ABI
declared in include/*.h and nothing else.  There is no CPU implementation
behind it -- if the libraries are missing or no HIP device is usable, calls
fail loudly.

  Engine ............ include/rtlws_hip.h (batch API on device buffers)
  Spectrum .......... include/spectrum.h      (reference src/spectrum.h:7-17)
  cic_decimate ...... include/resample.h      (reference src/resample.h:14)
  halfband_decimate . include/resample.h      (reference src/resample.h:17)
  RfDecimator ....... include/rf_decimator.h  (reference src/rf_decimator.h:6-21)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)                       # rtl-ws_amd/
LIB_DIR = os.path.join(ROOT, "lib")
# RTLWS_HIP_LIB selects an experiment build (make variant ...); default: the product library
HIP_LIB = os.environ.get("RTLWS_HIP_LIB") or os.path.join(LIB_DIR, "librtlws_hip.so")
# RTLWS_AMD_LIB: an instrumented build of the C host layer (tests/tools/asan_host_cpu.sh)
AMD_LIB = os.environ.get("RTLWS_AMD_LIB") or os.path.join(LIB_DIR, "librtlws_amd.so")
CBB_LIB = os.path.join(LIB_DIR, "librtlws_cbb.so")       # include/cbb_main.h
SYNTH_LIB = os.path.join(LIB_DIR, "librtlws_synth.so")   # synthetic rtl_sensor.h + signal_source.h

IN_CU8, IN_CS32, IN_RF32 = 0, 1, 2
WIN_RECT, WIN_HANN = 0, 1
OUT_POWER_SUM, OUT_MEAN_DB, OUT_PAYLOAD_U8 = 0, 1, 2
FLAG_ROWS_F32 = 1          # rtlws_spectra_batch_f64: f64 arithmetic, f32 rows
FLAG_F64 = 2               # rtlws_stream.h: this stream computes in f64

_INPUTS = {"cu8": IN_CU8, "cs32": IN_CS32, "rf32": IN_RF32}
_WINDOWS = {"rect": WIN_RECT, "hann": WIN_HANN, None: WIN_RECT}
_OUTPUTS = {"power_sum": OUT_POWER_SUM, "mean_db": OUT_MEAN_DB, "payload_u8": OUT_PAYLOAD_U8}


def build(jobs=8):
    """Compile every HIP/C source for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", ROOT, "-j%d" % jobs], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("rtl-ws_amd build failed:\n" + out.stdout[-4000:] + out.stderr[-4000:])


class SpectraDesc(C.Structure):
    _fields_ = [("n_fft", C.c_int), ("k_avg", C.c_int), ("input", C.c_int), ("window", C.c_int),
                ("output", C.c_int), ("cic_r", C.c_int), ("gain_db", C.c_int), ("flags", C.c_int)]


class CmplxS32(C.Structure):
    _fields_ = [("re", C.c_int32), ("im", C.c_int32)]


class TopoInfo(C.Structure):
    """rtlws_topo_info (include/rtlws_topo.h)"""
    _fields_ = [("device", C.c_int), ("bus_id", C.c_char * 32), ("numa_node", C.c_int), ("ncpus", C.c_int),
                ("cpulist", C.c_char * 512)]


class CicDelayLine(C.Structure):
    _fields_ = [("integrator_prev_out", CmplxS32), ("comb_prev_in", CmplxS32)]


# every symbol include/*.h declares, by library (tests check the exports)
HIP_SYMBOLS = [
    "rtlws_device_count", "rtlws_device_pci_bus_id", "rtlws_engine_create", "rtlws_engine_destroy", "rtlws_engine_device",
    "rtlws_engine_prepare", "rtlws_engine_prepare_f64", "rtlws_engine_set_option", "rtlws_engine_get_option",
    "rtlws_last_error", "rtlws_dev_alloc", "rtlws_dev_free", "rtlws_pinned_alloc",
    "rtlws_pinned_free", "rtlws_copy_h2d", "rtlws_copy_d2h", "rtlws_memset_dev",
    "rtlws_stream_sync", "rtlws_event_create", "rtlws_event_destroy", "rtlws_event_record",
    "rtlws_event_elapsed_ms", "rtlws_event_sync", "rtlws_spectra_batch", "rtlws_spectra_kernel_kind",
    "rtlws_cic_block_sums", "rtlws_halfband", "rtlws_spectra_grid", "rtlws_payload_from_sums",
    "rtlws_fm_demod", "rtlws_copy_d2d", "rtlws_spectra_batch_f64", "rtlws_payload_from_sums_f64",
    "rtlws_welch_accumulate_f64", "rtlws_welch_finish_f64",
    "rtlws_queue_create", "rtlws_queue_destroy", "rtlws_queue_wait_event", "rtlws_event_create_blocking",
    "rtlws_clock_probe_start", "rtlws_clock_probe_signal", "rtlws_clock_probe_signal_on_stream",
    "rtlws_clock_probe_stop", "rtlws_clock_stamp",
]
AUDIO_SYMBOLS = ["audio_init", "audio_new_audio_available", "audio_get_audio_payload",
                 "audio_fm_demodulator", "audio_close"]
STREAM_SYMBOLS = ["rtlws_stream_open", "rtlws_stream_open_q", "rtlws_stream_push", "rtlws_stream_flush",
                  "rtlws_stream_get_stats", "rtlws_stream_close", "rtlws_stream_device_for", "rtlws_stream_topology"]
AMD_SYMBOLS = [
    "spectrum_alloc", "spectrum_add_cmplx_u8", "spectrum_add_cmplx_s32", "spectrum_add_real_f32",
    "spectrum_free", "cic_decimate", "halfband_decimate", "rf_decimator_alloc",
    "rf_decimator_add_callback", "rf_decimator_set_parameters", "rf_decimator_decimate_cmplx_u8",
    "rf_decimator_remove_callbacks", "rf_decimator_free",
]

# include/rtlws_multi.h: one device-resident batch sharded over the devices of a node (librtlws_amd.so)
MULTI_SYMBOLS = ["rtlws_multi_partition", "rtlws_multi_open", "rtlws_multi_shards", "rtlws_multi_frames",
                 "rtlws_multi_frame_bytes", "rtlws_multi_row_bytes", "rtlws_multi_upload", "rtlws_multi_run",
                 "rtlws_multi_download", "rtlws_multi_close", "rtlws_multi_error", "rtlws_multi_shard_topology"]
# include/rtlws_topo.h (librtlws_amd.so)
TOPO_SYMBOLS = ["rtlws_topo_describe", "rtlws_topo_parse_cpulist", "rtlws_topo_pin_thread"]

# include/rtlws_host.h: sticky failure record of the void entry points (librtlws_amd.so)
HOST_SYMBOLS = ["rtlws_host_error", "rtlws_host_error_count", "rtlws_host_error_clear", "rtlws_host_fail",
                "rtlws_host_device"]

CBB_SYMBOLS = ["cbb_init", "cbb_rf_decimator", "cbb_get_rtl_dev", "cbb_new_spectrum_available",
               "cbb_get_spectrum_payload", "cbb_close",
               "rtlws_cbb_published_frames", "rtlws_cbb_samples_seen"]      # include/rtlws_cbb.h
SYNTH_SYMBOLS = ["rtl_init", "rtl_set_frequency", "rtl_set_sample_rate", "rtl_set_gain", "rtl_freq",
                 "rtl_sample_rate", "rtl_gain", "rtl_read_async", "rtl_cancel", "rtl_close",
                 "signal_source_start", "signal_source_add_callback",
                 "signal_source_remove_callbacks", "signal_source_stop"]

_hip = None
_amd = None
_cbb = None


_tried_build = False


def _need(path):
    """The product libraries are built in-tree (make -C rtl-ws_amd).  If one is
    missing, build once (hipcc/gcc are on every box of this image); if it is
    still missing, fail loudly -- there is no CPU implementation to fall back to."""
    global _tried_build
    if not os.path.exists(path) and not _tried_build and not os.environ.get("RTLWS_HIP_LIB"):
        _tried_build = True
        build()
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C rtl-ws_amd). There is no CPU fallback." % path)


def hip_lib():
    global _hip
    if _hip is None:
        _need(HIP_LIB)
        L = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        vp, i, l, sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
        L.rtlws_device_count.restype = i
        L.rtlws_engine_create.argtypes = [i]
        L.rtlws_engine_create.restype = vp
        L.rtlws_engine_destroy.argtypes = [vp]
        L.rtlws_engine_device.argtypes = [vp]
        L.rtlws_engine_prepare.argtypes = [vp, i]
        L.rtlws_engine_prepare_f64.argtypes = [vp, i]
        L.rtlws_engine_set_option.argtypes = [vp, C.c_char_p, i]
        L.rtlws_engine_get_option.argtypes = [vp, C.c_char_p]
        L.rtlws_last_error.restype = C.c_char_p
        L.rtlws_dev_alloc.argtypes = [vp, sz]
        L.rtlws_dev_alloc.restype = vp
        L.rtlws_dev_free.argtypes = [vp, vp]
        L.rtlws_pinned_alloc.argtypes = [sz]
        L.rtlws_pinned_alloc.restype = vp
        L.rtlws_pinned_free.argtypes = [vp]
        L.rtlws_copy_h2d.argtypes = [vp, vp, vp, sz, vp]
        L.rtlws_copy_d2h.argtypes = [vp, vp, vp, sz, vp]
        L.rtlws_memset_dev.argtypes = [vp, vp, i, sz, vp]
        L.rtlws_stream_sync.argtypes = [vp, vp]
        L.rtlws_event_create.restype = vp
        L.rtlws_event_create_blocking.restype = vp
        L.rtlws_queue_create.argtypes = [vp]
        L.rtlws_queue_create.restype = vp
        L.rtlws_queue_destroy.argtypes = [vp, vp]
        L.rtlws_queue_wait_event.argtypes = [vp, vp, vp]
        L.rtlws_event_destroy.argtypes = [vp]
        L.rtlws_event_record.argtypes = [vp, vp, vp]
        L.rtlws_event_elapsed_ms.argtypes = [vp, vp]
        L.rtlws_event_sync.argtypes = [vp]
        L.rtlws_event_elapsed_ms.restype = C.c_float
        L.rtlws_spectra_batch.argtypes = [vp, C.POINTER(SpectraDesc), vp, l, vp, vp]
        L.rtlws_spectra_batch_f64.argtypes = [vp, C.POINTER(SpectraDesc), vp, l, vp, vp]
        L.rtlws_payload_from_sums_f64.argtypes = [vp, vp, i, i, i, vp, vp]
        L.rtlws_welch_accumulate_f64.argtypes = [vp, vp, vp, i, l, vp, vp]
        L.rtlws_welch_finish_f64.argtypes = [vp, vp, i, l, vp, vp]
        L.rtlws_spectra_kernel_kind.argtypes = [C.POINTER(SpectraDesc)]
        L.rtlws_cic_block_sums.argtypes = [vp, i, vp, l, vp, vp]
        L.rtlws_halfband.argtypes = [vp, vp, vp, l, vp]
        L.rtlws_payload_from_sums.argtypes = [vp, vp, i, i, i, vp, vp]
        L.rtlws_fm_demod.argtypes = [vp, vp, l, vp, vp, vp, vp]
        L.rtlws_copy_d2d.argtypes = [vp, vp, vp, sz, vp]
        L.rtlws_clock_stamp.argtypes = [vp, vp, C.c_int, vp]
        L.rtlws_clock_probe_start.argtypes = [vp]
        L.rtlws_clock_probe_start.restype = vp
        L.rtlws_clock_probe_signal.argtypes = [vp]
        L.rtlws_clock_probe_signal.restype = None
        L.rtlws_clock_probe_signal_on_stream.argtypes = [vp, vp]
        L.rtlws_clock_probe_stop.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rtlws_spectra_grid.argtypes = [vp, C.POINTER(SpectraDesc), l, C.POINTER(i),
                                         C.POINTER(i), C.POINTER(i)]
        _hip = L
    return _hip


def amd_lib():
    global _amd
    if _amd is None:
        hip_lib()
        _need(AMD_LIB)
        L = C.CDLL(AMD_LIB)
        vp, i = C.c_void_p, C.c_int
        L.spectrum_alloc.argtypes = [i]
        L.spectrum_alloc.restype = vp
        for n in ("spectrum_add_cmplx_u8", "spectrum_add_cmplx_s32", "spectrum_add_real_f32"):
            f = getattr(L, n)
            f.argtypes = [vp, vp, vp, i]
            f.restype = i
        L.spectrum_free.argtypes = [vp]
        L.cic_decimate.argtypes = [i, vp, i, vp, i, C.POINTER(CicDelayLine)]
        L.cic_decimate.restype = i
        L.halfband_decimate.argtypes = [vp, vp, i, vp]
        L.halfband_decimate.restype = None
        L.rf_decimator_alloc.restype = vp
        L.rf_decimator_add_callback.argtypes = [vp, vp]
        L.rf_decimator_set_parameters.argtypes = [vp, C.c_double, i]
        L.rf_decimator_set_parameters.restype = i
        L.rf_decimator_decimate_cmplx_u8.argtypes = [vp, vp, i]
        L.rf_decimator_decimate_cmplx_u8.restype = i
        L.rf_decimator_remove_callbacks.argtypes = [vp]
        L.rf_decimator_free.argtypes = [vp]
        lp = C.POINTER(C.c_long)
        L.rtlws_multi_partition.argtypes = [C.c_long, i, i, i, lp, lp]
        L.rtlws_multi_open.argtypes = [i, C.POINTER(i), C.POINTER(SpectraDesc), C.c_long, i]
        L.rtlws_multi_open.restype = vp
        L.rtlws_multi_shards.argtypes = [vp]
        L.rtlws_multi_frames.argtypes = [vp]
        L.rtlws_multi_frames.restype = C.c_long
        L.rtlws_multi_frame_bytes.argtypes = [vp]
        L.rtlws_multi_frame_bytes.restype = C.c_size_t
        L.rtlws_multi_row_bytes.argtypes = [vp]
        L.rtlws_multi_row_bytes.restype = C.c_size_t
        L.rtlws_multi_upload.argtypes = [vp, vp]
        L.rtlws_multi_run.argtypes = [vp, i, vp, C.POINTER(C.c_double)]
        L.rtlws_multi_download.argtypes = [vp, vp]
        L.rtlws_multi_close.argtypes = [vp]
        L.rtlws_multi_error.argtypes = [vp]
        L.rtlws_multi_error.restype = C.c_char_p
        L.rtlws_multi_shard_topology.argtypes = [vp, i, C.POINTER(TopoInfo), C.POINTER(i)]
        L.rtlws_topo_describe.argtypes = [i, C.c_char_p, C.c_char_p, C.POINTER(TopoInfo)]
        L.rtlws_topo_parse_cpulist.argtypes = [C.c_char_p, vp, i]
        L.rtlws_topo_pin_thread.argtypes = [C.POINTER(TopoInfo)]
        L.rtlws_host_error.restype = C.c_char_p
        L.rtlws_host_error_count.restype = C.c_long
        L.rtlws_host_error_clear.restype = None
        _amd = L
    return _amd


def last_error():
    return hip_lib().rtlws_last_error().decode()


def host_error():
    """(count, first message) of include/rtlws_host.h's sticky failure record."""
    L = amd_lib()
    return L.rtlws_host_error_count(), L.rtlws_host_error().decode()


def host_error_clear():
    amd_lib().rtlws_host_error_clear()


def device_count():
    return hip_lib().rtlws_device_count()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


STREAM_ENGINE, STREAM_DEFAULT = None, 1     # include/rtlws_hip.h "Streams"


def torch_stream_handle(stream=None):
    """The `stream` argument for a torch caller: torch's default stream has handle 0, which
    the C-ABI reads as "the engine's own non-blocking stream" (NOT ordered against torch's
    kernels) -- so 0 is mapped to RTLWS_STREAM_DEFAULT, HIP's default stream itself."""
    import torch
    h = (torch.cuda.current_stream() if stream is None else stream).cuda_stream
    return h if h else STREAM_DEFAULT


def make_desc(n_fft, k_avg=1, input="cu8", window="rect", output="power_sum", cic_r=0, gain_db=0, flags=0):
    return SpectraDesc(int(n_fft), int(k_avg), _INPUTS.get(input, input), _WINDOWS.get(window, window),
                       _OUTPUTS.get(output, output), int(cic_r), int(gain_db), int(flags))


class DevBuf:
    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, int(nbytes)
        self.ptr = hip_lib().rtlws_dev_alloc(eng.h, self.nbytes)
        if not self.ptr:
            raise RuntimeError("rtlws_dev_alloc(%d) failed: %s" % (nbytes, last_error()))

    def free(self):
        if self.ptr:
            hip_lib().rtlws_dev_free(self.eng.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One engine per device (include/rtlws_hip.h). Fails loudly without a GPU."""

    def __init__(self, device=0):
        self.h = hip_lib().rtlws_engine_create(int(device))
        if not self.h:
            raise RuntimeError("rtlws_engine_create(%d) failed: %s" % (device, last_error()))
        self.device = device

    def close(self):
        if self.h:
            hip_lib().rtlws_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ----------------------------------------------------------
    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        buf = DevBuf(self, arr.nbytes)
        self._chk(hip_lib().rtlws_copy_h2d(self.h, buf.ptr, _p(arr), arr.nbytes, stream), "h2d")
        self.sync(stream)
        return buf

    def download(self, buf, dtype, shape, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= buf.nbytes
        self._chk(hip_lib().rtlws_copy_d2h(self.h, _p(out), buf.ptr, out.nbytes, stream), "d2h")
        self.sync(stream)
        return out

    def sync(self, stream=None):
        self._chk(hip_lib().rtlws_stream_sync(self.h, stream), "sync")

    def clock_stamp(self, d_out, slots, stream=None):
        """`slots` one-wavefront workgroups on `stream` write {shader clocks, 100 MHz ticks, place, magic} to the
        device pointer d_out (slots x 4 64-bit words)."""
        self._chk(hip_lib().rtlws_clock_stamp(self.h, self._ptr(d_out), int(slots), stream), "rtlws_clock_stamp")

    @staticmethod
    def clock_from_stamps(a0, a1):
        """(sclk_ghz, seconds, places) between two stamp launches (arrays of shape [slots, 4]): the records are paired
        by place -- s_memtime is a counter of the place it is read at -- and the median of d(memtime) / d(memrealtime)
        x 100 MHz over the places both launches reached is returned, with the median interval and the number of
        places; (None, 0.0, 0) when no place is common or the interval is empty."""
        first = {}
        for mt, rt, place, magic in np.asarray(a0).astype(np.int64).tolist():
            if magic == 0x5354414d50 and place not in first:
                first[place] = (mt, rt)
        ghz, secs = [], []
        seen = set()
        for mt, rt, place, magic in np.asarray(a1).astype(np.int64).tolist():
            if magic == 0x5354414d50 and place in first and place not in seen:
                seen.add(place)
                dmt, drt = mt - first[place][0], rt - first[place][1]
                if drt > 0 and dmt > 0:
                    ghz.append(dmt / (drt * 10.0))
                    secs.append(drt * 1e-8)
        if not ghz:
            return None, 0.0, 0
        return float(np.median(ghz)), float(np.median(secs)), len(ghz)

    def clock_probe_start(self):
        """A wavefront beside the next launches that measures the shader clock they run at."""
        h = hip_lib().rtlws_clock_probe_start(self.h)
        if not h:
            raise RuntimeError("rtlws_clock_probe_start failed: %s" % last_error())
        return h

    def clock_probe_signal(self, probe):
        hip_lib().rtlws_clock_probe_signal(probe)

    def clock_probe_signal_on_stream(self, probe, stream=None):
        self._chk(hip_lib().rtlws_clock_probe_signal_on_stream(probe, stream), "rtlws_clock_probe_signal_on_stream")

    def clock_probe_stop(self, probe):
        """(sclk_ghz, seconds) of the interval since clock_probe_start."""
        g, s = C.c_double(0.0), C.c_double(0.0)
        self._chk(hip_lib().rtlws_clock_probe_stop(probe, C.byref(g), C.byref(s)), "rtlws_clock_probe_stop")
        return g.value, s.value

    def set_option(self, name, value):
        """Kernel-selection switch of this engine (include/rtlws_hip.h: rtlws_engine_set_option)."""
        self._chk(hip_lib().rtlws_engine_set_option(self.h, name.encode(), int(value)), "set_option(%s)" % name)

    def get_option(self, name):
        return hip_lib().rtlws_engine_get_option(self.h, name.encode())

    def option(self, name, value):
        """Context manager: the option set to `value` inside the block, restored afterwards."""
        eng = self

        class _Scope:
            def __enter__(self_):
                self_.old = eng.get_option(name)
                eng.set_option(name, value)

            def __exit__(self_, *exc):
                eng.set_option(name, self_.old)
        return _Scope()

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, last_error()))

    # -- kernels (device pointers: ints or DevBuf) --------------------------
    @staticmethod
    def _ptr(x):
        return x.ptr if isinstance(x, DevBuf) else int(x)

    def spectra_batch(self, desc, d_in, nframes, d_out, stream=None, check=True):
        rc = hip_lib().rtlws_spectra_batch(self.h, C.byref(desc), self._ptr(d_in), int(nframes),
                                           self._ptr(d_out), stream)
        if check:
            self._chk(rc, "rtlws_spectra_batch")
        return rc

    def spectra_batch_f64(self, desc, d_in, nframes, d_out, stream=None, check=True):
        rc = hip_lib().rtlws_spectra_batch_f64(self.h, C.byref(desc), self._ptr(d_in), int(nframes),
                                               self._ptr(d_out), stream)
        if check:
            self._chk(rc, "rtlws_spectra_batch_f64")
        return rc

    def cic_block_sums(self, R, d_src, dst_len, d_dst, stream=None, check=True):
        rc = hip_lib().rtlws_cic_block_sums(self.h, int(R), self._ptr(d_src), int(dst_len),
                                            self._ptr(d_dst), stream)
        if check:
            self._chk(rc, "rtlws_cic_block_sums")
        return rc

    def halfband(self, d_x, d_y, out_len, stream=None):
        self._chk(hip_lib().rtlws_halfband(self.h, self._ptr(d_x), self._ptr(d_y), int(out_len),
                                           stream), "rtlws_halfband")

    def grid(self, desc, nframes):
        b, t, l = C.c_int(), C.c_int(), C.c_int()
        rc = hip_lib().rtlws_spectra_grid(self.h, C.byref(desc), int(nframes), C.byref(b),
                                          C.byref(t), C.byref(l))
        return rc, b.value, t.value, l.value

    # -- convenience: host arrays in, host arrays out ------------------------
    def spectra(self, data, n_fft, k_avg=1, input="cu8", window="rect", output="power_sum",
                cic_r=0, gain_db=0, f64=False, rows_f32=False):
        """f64=True: rtlws_spectra_batch_f64 (the reference-precision kernels, f64 rows;
        rows_f32=True: the same arithmetic with rows rounded once to f32, RTLWS_FLAG_ROWS_F32)."""
        desc = make_desc(n_fft, k_avg, input, window, output, cic_r, gain_db,
                         FLAG_ROWS_F32 if (f64 and rows_f32) else 0)
        data = np.ascontiguousarray(data)
        per_sample = {"cu8": 2, "cs32": 8, "rf32": 4}[input] * max(int(cic_r), 1)
        nframes = data.nbytes // (per_sample * n_fft)
        assert nframes * per_sample * n_fft == data.nbytes
        rows = nframes // k_avg
        out_dtype = np.uint8 if output == "payload_u8" else (np.float64 if (f64 and not rows_f32) else np.float32)
        d_in = self.upload(data)
        d_out = self.alloc(rows * n_fft * np.dtype(out_dtype).itemsize)
        (self.spectra_batch_f64 if f64 else self.spectra_batch)(desc, d_in, nframes, d_out)
        res = self.download(d_out, out_dtype, (rows, n_fft))
        d_in.free()
        d_out.free()
        return res


# ---- drop-in API (librtlws_amd.so) -----------------------------------------

class Spectrum:
    """struct spectrum* of include/spectrum.h."""

    def __init__(self, N):
        self.N = N
        self.h = amd_lib().spectrum_alloc(int(N))
        if not self.h:
            raise RuntimeError("spectrum_alloc(%d) returned NULL: %s" % (N, last_error()))

    def add_cmplx_u8(self, src, ps, length=None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        return amd_lib().spectrum_add_cmplx_u8(self.h, _p(src), _p(ps),
                                               self.N if length is None else length)

    def add_cmplx_s32(self, src, ps, length=None):
        src = np.ascontiguousarray(src, dtype=np.int32)
        return amd_lib().spectrum_add_cmplx_s32(self.h, _p(src), _p(ps),
                                                self.N if length is None else length)

    def add_real_f32(self, src, ps, length=None):
        src = np.ascontiguousarray(src, dtype=np.float32)
        return amd_lib().spectrum_add_real_f32(self.h, _p(src), _p(ps),
                                               self.N if length is None else length)

    def free(self):
        if self.h:
            amd_lib().spectrum_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def cic_decimate(R, src, state=None, dst_len=None):
    """include/resample.h cic_decimate. Returns (rc, dst int32[dst_len,2], state int32[4])."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1, 2)
    src_len = src.shape[0]
    if dst_len is None:
        dst_len = src_len // R if R > 0 else 0
    dst = np.zeros((max(dst_len, 0), 2), dtype=np.int32)
    st = [0, 0, 0, 0] if state is None else [int(x) for x in state]
    d = CicDelayLine(CmplxS32(st[0], st[1]), CmplxS32(st[2], st[3]))
    rc = amd_lib().cic_decimate(int(R), _p(src), src_len, _p(dst), dst_len, C.byref(d))
    out_state = np.array([d.integrator_prev_out.re, d.integrator_prev_out.im,
                          d.comb_prev_in.re, d.comb_prev_in.im], dtype=np.int32)
    return rc, dst, out_state


def halfband_decimate(inp, delay):
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    assert delay.dtype == np.float32 and delay.size == 10
    out = np.empty(inp.size // 2, dtype=np.float32)
    amd_lib().halfband_decimate(_p(inp), _p(out), out.size, _p(delay))
    return out


_RF_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class RfDecimator:
    """struct rf_decimator* of include/rf_decimator.h; callback blocks are collected."""

    def __init__(self):
        self.h = amd_lib().rf_decimator_alloc()
        self.blocks = []

        def _cb(ptr, n):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 2))
            self.blocks.append(a.copy())

        self._cb = _RF_CB(_cb)
        amd_lib().rf_decimator_add_callback(self.h, C.cast(self._cb, C.c_void_p))

    def set_parameters(self, sample_rate, down_factor):
        return amd_lib().rf_decimator_set_parameters(self.h, float(sample_rate), int(down_factor))

    def decimate(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.uint8).reshape(-1, 2)
        return amd_lib().rf_decimator_decimate_cmplx_u8(self.h, _p(iq), iq.shape[0])

    def free(self):
        if self.h:
            amd_lib().rf_decimator_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- include/rtlws_multi.h ----------------------------------------------------

class MultiShardStats(C.Structure):
    _fields_ = [("device", C.c_int), ("first_frame", C.c_long), ("frames", C.c_long), ("launches", C.c_int),
                ("event_ms", C.c_double), ("wall_ms", C.c_double), ("rc", C.c_int)]


def multi_partition(nframes, k_avg, shards, g):
    """rtlws_multi_partition: (rc, first_frame, frame_count).  No GPU needed."""
    a, b = C.c_long(-1), C.c_long(-1)
    rc = amd_lib().rtlws_multi_partition(int(nframes), int(k_avg), int(shards), int(g), C.byref(a), C.byref(b))
    return rc, a.value, b.value


class MultiBatch:
    """rtlws_multi*: one batch sharded over devices (device_ids: one shard per entry; None = every device)."""

    def __init__(self, desc, nframes, device_ids=None, f64=False):
        L = amd_lib()
        n = 0 if device_ids is None else len(device_ids)
        ids = None if device_ids is None else (C.c_int * n)(*device_ids)
        self.desc, self.f64 = desc, f64
        self.h = L.rtlws_multi_open(n, ids, C.byref(desc), int(nframes), 1 if f64 else 0)
        if not self.h:
            raise RuntimeError("rtlws_multi_open failed: %s" % (L.rtlws_multi_error(None).decode() or last_error()))
        self.shards = L.rtlws_multi_shards(self.h)
        self.frames = L.rtlws_multi_frames(self.h)

    def upload(self, frames):
        frames = np.ascontiguousarray(frames)
        assert frames.nbytes >= self.frames * amd_lib().rtlws_multi_frame_bytes(self.h)
        rc = amd_lib().rtlws_multi_upload(self.h, _p(frames))
        if rc:
            raise RuntimeError("rtlws_multi_upload rc=%d: %s" % (rc, self.error()))

    def run(self, launches=1):
        st = (MultiShardStats * self.shards)()
        wall = C.c_double(0.0)
        rc = amd_lib().rtlws_multi_run(self.h, int(launches), st, C.byref(wall))
        if rc:
            raise RuntimeError("rtlws_multi_run rc=%d: %s" % (rc, self.error()))
        return list(st), wall.value

    def download(self):
        rows = self.frames // self.desc.k_avg
        rb = amd_lib().rtlws_multi_row_bytes(self.h)
        dt = np.uint8 if self.desc.output == OUT_PAYLOAD_U8 else {4: np.float32, 8: np.float64}[rb // self.desc.n_fft]
        out = np.empty((rows, self.desc.n_fft), dtype=dt)
        rc = amd_lib().rtlws_multi_download(self.h, _p(out))
        if rc:
            raise RuntimeError("rtlws_multi_download rc=%d: %s" % (rc, self.error()))
        return out

    def error(self):
        """why the last upload / run / download failed (the failing call ran on a shard's own thread)"""
        return amd_lib().rtlws_multi_error(self.h).decode()

    def topology(self, g):
        """(TopoInfo, cpus the shard's thread pinned itself to) of shard g"""
        t, n = TopoInfo(), C.c_int(0)
        if amd_lib().rtlws_multi_shard_topology(self.h, int(g), C.byref(t), C.byref(n)) != 0:
            raise IndexError(g)
        return t, n.value

    def close(self):
        if self.h:
            amd_lib().rtlws_multi_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def topo_describe(device=-1, bus_id=None, sysfs_root=None):
    """rtlws_topo_describe: TopoInfo of a device (bus_id given: no GPU is asked).  None on bad arguments."""
    t = TopoInfo()
    rc = amd_lib().rtlws_topo_describe(int(device), None if bus_id is None else bus_id.encode(),
                                       None if sysfs_root is None else str(sysfs_root).encode(), C.byref(t))
    return t if rc == 0 else None


# ---- boundary #2 (librtlws_cbb.so over the synthetic sensor) ------------------

def cbb_lib():
    """cbb_main.h entry points.  The sensor / signal-source symbols they need
    come from librtlws_synth.so here (in rtl-ws they come from the server's own
    rtl_sensor.c / signal_source.c)."""
    global _cbb
    if _cbb is None:
        amd_lib()
        _need(SYNTH_LIB)
        _need(CBB_LIB)
        C.CDLL(SYNTH_LIB, mode=C.RTLD_GLOBAL)
        L = C.CDLL(CBB_LIB)
        L.cbb_init.argtypes = [C.c_int]
        L.cbb_init.restype = None
        L.cbb_rf_decimator.restype = C.c_void_p
        L.cbb_get_rtl_dev.restype = C.c_void_p
        L.cbb_new_spectrum_available.restype = C.c_int
        L.cbb_get_spectrum_payload.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cbb_get_spectrum_payload.restype = C.c_int
        L.cbb_close.restype = None
        L.rtlws_cbb_samples_seen.restype = C.c_uint64
        _cbb = L
    return _cbb


def cbb_payload(gain_db, buf_len=8192):
    buf = np.zeros(buf_len, dtype=np.uint8)
    n = cbb_lib().cbb_get_spectrum_payload(_p(buf), buf_len, int(gain_db))
    return buf[:n].copy()
