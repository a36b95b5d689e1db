"""Read kernel resource metadata out of a built HIP shared library.

hipcc embeds one clang offload bundle per translation unit in the library's
.hip_fatbin section; each bundle carries the gfx950 code object (an ELF) whose
NT_AMDGPU_METADATA note lists, per kernel, the register counts, spills and
scratch the hardware will be asked for.  tests/test_abi_cpu.py uses this to
keep the hot instantiations spill-free; tools/ use it for resource tables.

No GPU needed.  Uses llvm-readelf from the ROCm LLVM (same image everywhere).
"""
import os
import re
import struct
import subprocess
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    """Raw bytes of an ELF64 section (little endian)."""
    with open(path, "rb") as f:
        data = f.read()
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        h = sh(i)
        nm = data[stroff + h[0]:data.index(b"\0", stroff + h[0])].decode()
        if nm == name:
            return data[h[4]:h[4] + h[5]]
    raise KeyError(name)


def code_objects(lib_path, arch="gfx950"):
    """Yield the device ELF images for `arch` embedded in lib_path."""
    fat = _section(lib_path, ".hip_fatbin")
    pos = 0
    while True:
        pos = fat.find(_MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", fat, pos + 24)
        q = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fat, q)
            triple = fat[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if arch in triple and size:
                yield fat[pos + off:pos + off + size]
        pos += 24


_KEYS = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count",
         ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size",
         ".max_flat_workgroup_size")


def kernels(lib_path, arch="gfx950", demangle=True):
    """List of dicts (name, vgpr_count, vgpr_spill_count, private_segment_fixed_size, ...)."""
    out = []
    for img in code_objects(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as tf:
            tf.write(img)
            tf.flush()
            txt = subprocess.run([READELF, "--notes", tf.name], capture_output=True, text=True,
                                 check=True).stdout
        # a kernel record starts with "  - .key:" and its own keys sit at a
        # four-space indent; nested lists (.args) are indented deeper
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^  (- |  )(\.[a-z_]+):\s*(.*)$", line)
            if not m:
                continue
            if m.group(1) == "- ":
                cur = {}
                out.append(cur)
            if cur is None or m.group(2) not in _KEYS:
                continue
            v = m.group(3).strip().strip("'\"")
            cur[m.group(2)] = int(v) if re.fullmatch(r"-?\d+", v) else v
    recs = []
    for r in out:
        if ".vgpr_count" in r and ".name" in r:
            recs.append({k.lstrip("."): v for k, v in r.items()})
    if demangle and recs:
        names = "\n".join(r["name"] for r in recs)
        dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
        for r, d in zip(recs, dem):
            r["demangled"] = d
    return recs


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "lib", "librtlws_hip.so")
    for r in kernels(lib):
        print("%-110s vgpr %3d agpr %3d spillV %3d scratch %4d lds %6d" % (
            r.get("demangled", r["name"])[:110], r["vgpr_count"], r.get("agpr_count", 0),
            r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0),
            r.get("group_segment_fixed_size", 0)))
