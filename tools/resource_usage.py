#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) -> a table.
usage: tools/resource_usage.py [out.txt]   (run where hipcc is: the build container cross-compiles)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "salt_amd", "csrc")
rows, cur = [], None
for f in ("salt_align", "salt_pe", "salt_text", "salt_index", "salt_sufsort"):
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        os.path.join(src, f + ".hip"), "-o", "/dev/null"], capture_output=True, text=True, cwd=src)
    for l in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = {"name": m.group(1), "file": f}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, l)
            if m:
                cur[key] = int(m.group(1))
def dem(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
out = ["# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3); occ = waves per SIMD the registers / LDS allow",
       "%-56s %6s %6s %6s %8s %8s %4s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch", "LDS B", "occ")]
seen = set()
for r in rows:
    n = dem(r["name"])
    if n in seen or "rocprim" in n:
        continue
    seen.add(n)
    out.append("%-56s %6s %6s %6s %8s %8s %4s" % (n[:56], r.get("vgpr", "-"), r.get("agpr", "-"), r.get("sgpr", "-"), r.get("scratch", "-"), r.get("lds", "-"), r.get("occ", "-")))
txt = "\n".join(out) + "\n"
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt)
print(txt)
