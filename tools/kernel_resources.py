#!/usr/bin/env python
"""Per-kernel resources of the shipped library, read from the gfx950 code object inside libderp_hip.so (the clang offload
bundle in its .hip_fatbin section -> the ELF's AMDGPU metadata note): VGPRs, SGPRs, scratch and LDS bytes, spills, max
flat workgroup size, and the waves per SIMD those VGPRs allow on gfx950 (512 VGPRs per SIMD lane, granule 8).
usage: python tools/kernel_resources.py [library] > profiles/r04_kernel_resources.txt"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "facebook360_dep_amd", "libderp_hip.so")
data = open(lib, "rb").read()
at = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
(n,) = struct.unpack_from("<Q", data, at + 24)
p, co = at + 32, None
for _ in range(n):
    off, size, tlen = struct.unpack_from("<QQQ", data, p)
    p += 24
    triple = data[p:p + tlen].decode()
    p += tlen
    if "gfx950" in triple:
        co = data[at + off: at + off + size]
assert co, "no gfx950 code object in " + lib
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], text=True)
    filt = subprocess.run(["c++filt"], input="\n".join(re.findall(r"\.name:\s+(\S+)", notes)),
                          capture_output=True, text=True).stdout.split("\n")
rows = []
for block in notes.split("  - .agpr_count:")[1:]:
    get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))  # noqa: E731
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    rows.append((name, get("vgpr_count"), int(re.match(r"\s*(\d+)", block).group(1)), get("sgpr_count"), get("private_segment_fixed_size"),
                 get("group_segment_fixed_size"), get("vgpr_spill_count"), get("sgpr_spill_count"), get("max_flat_workgroup_size")))
names = dict(zip(re.findall(r"\.name:\s+(\S+)", notes), filt))
print("# %s: %d kernels, gfx950 code object %d bytes" % (os.path.basename(lib), len(rows), len(co)))
print("%-34s %5s %5s %5s %8s %7s %6s %6s %6s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "scratchB", "ldsB", "vspill", "sspill", "maxwg", "waves"))
for name, v, a, s, priv, lds, vs, ss, wg in sorted(rows, key=lambda r: -r[1]):
    short = re.sub(r"\(.*", "", names.get(name, name)).replace("derp::", "").replace("void ", "")
    total = ((v + a + 7) // 8) * 8
    waves = min(8, 512 // max(total, 8))
    print("%-34s %5d %5d %5d %8d %7d %6d %6d %6d %5d" % (short[:34], v, a, s, priv, lds, vs, ss, wg, waves))
