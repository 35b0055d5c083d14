#!/usr/bin/env python3
"""VGPR / SGPR / LDS / spill figures of the kernels in libruart_hip.so, read from the code objects' metadata (no GPU needed).
    python tools/kernel_resources.py [name-substring ...]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = open(os.path.join(ROOT, "ruart_amd", "libruart_hip.so"), "rb").read()
pats = sys.argv[1:]
i = 0
rows = []
while True:
    i = d.find(b"\x7fELF", i + 1)
    if i < 0:
        break
    if struct.unpack_from("<H", d, i + 18)[0] != 224:          # EM_AMDGPU
        continue
    shoff = struct.unpack_from("<Q", d, i + 40)[0]
    shentsize, shnum = struct.unpack_from("<HH", d, i + 58)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(d[i:i + shoff + shentsize * shnum])
        f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count"), g("max_flat_workgroup_size")))
print("%-8s %-6s %-8s %-6s %-6s name" % ("vgpr", "sgpr", "lds", "spill", "wg"))
for name, v, s_, l, sp, wg in sorted(rows):
    if not pats or any(p in name for p in pats):
        print("%-8s %-6s %-8s %-6s %-6s %s" % (v, s_, l, sp, wg, name[:150]))
