#!/usr/bin/env python3
"""Compile one .hip source for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(registers, LDS, scratch, occupancy).  usage: tools/kernel_resources.py blight_amd/csrc/blmx_reads.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-fno-gpu-rdc", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+(\w[\w ]*?):\s+(\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if "rocprim" in r["name"] or flt not in r["name"]:
        continue
    print("%-70s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d occ %d" % (r["name"][:70], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", r.get("SGPRs", -1)),
          r.get("LDS Size", 0), r.get("ScratchSize", 0), r.get("Occupancy", -1)))
