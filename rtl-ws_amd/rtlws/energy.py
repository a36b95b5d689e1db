"""Package energy accumulator of ONE GPU, found by PCI bus id (measurement plumbing for bench.py and tools/).

rocm_smi enumerates every physical GPU of the host and ignores HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, so a HIP
ordinal is not an rsmi index: the device is looked up by the bus id HIP reports for it (rtlws_device_pci_bus_id,
include/rtlws_hip.h) against rsmi_dev_pci_id_get of every rsmi device.  No match, no library, no counter: None --
the caller drops its energy leg, it never reads another GPU's accumulator.
"""
import ctypes as C

_smi = None        # None: not tried; False: unusable


def _lib():
    global _smi
    if _smi is None:
        try:
            lib = C.CDLL("librocm_smi64.so")
            _smi = lib if lib.rsmi_init(C.c_uint64(0)) == 0 else False
        except OSError:
            _smi = False
    return _smi or None


def parse_bus_id(text):
    """"0000:0d:00.0" -> (domain, bus, device, function), or None."""
    try:
        dom, bus, rest = text.strip().split(":")
        dev, fn = rest.split(".")
        return int(dom, 16), int(bus, 16), int(dev, 16), int(fn, 16)
    except (ValueError, AttributeError):
        return None


def bdf_fields(bdfid):
    """rsmi's 64-bit BDFID -> (domain, bus, device, function); bits 28-31 (a partition id on some
    releases) are not part of the address."""
    return (bdfid >> 32) & 0xffffffff, (bdfid >> 8) & 0xff, (bdfid >> 3) & 0x1f, bdfid & 0x7


def rsmi_index_for_bus_id(bus_id, lib=None):
    """The rsmi device index whose PCI address is `bus_id` (text), or None."""
    want = parse_bus_id(bus_id)
    lib = lib or _lib()
    if want is None or lib is None:
        return None
    n = C.c_uint32(0)
    if lib.rsmi_num_monitor_devices(C.byref(n)) != 0:
        return None
    for i in range(n.value):
        b = C.c_uint64(0)
        if lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(b)) == 0 and bdf_fields(b.value) == want:
            return i
    return None


class EnergyCounter:
    """joules() of the GPU at `bus_id`; .index is None when the GPU cannot be identified."""

    def __init__(self, bus_id, lib=None):
        self.bus_id = bus_id
        self.lib = lib or _lib()
        self.index = rsmi_index_for_bus_id(bus_id, self.lib) if self.lib is not None else None

    def joules(self):
        if self.index is None:
            return None
        cnt, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
        if self.lib.rsmi_dev_energy_count_get(C.c_uint32(self.index), C.byref(cnt), C.byref(res), C.byref(ts)) != 0:
            return None
        return cnt.value * res.value * 1e-6


def for_hip_device(rtlws, device):
    """EnergyCounter of HIP device `device` (its bus id through the C-ABI), or None."""
    buf = C.create_string_buffer(64)
    try:
        if rtlws.hip_lib().rtlws_device_pci_bus_id(int(device), buf, 64) != 0:
            return None
    except Exception:
        return None
    ec = EnergyCounter(buf.value.decode())
    return ec if ec.index is not None else None
