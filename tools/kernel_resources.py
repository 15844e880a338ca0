#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of one translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: tools/kernel_resources.py sushi_fft [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sushi_amd import build  # noqa: E402

unit = sys.argv[1]
flags = next(f for n, f, _ in build.UNITS if n == unit)
cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall"] + flags + \
      sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(build.CSRC, unit + ".hip"), "-o", "/tmp/_kr.o"]
build.write_twiddles()
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: +(?:Function )?Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
    elif "error" in line:
        print(line)
print("%-44s %5s %5s %8s %7s %5s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for r in rows:
    print("%-44s %5s %5s %8s %7s %5s" % (r["name"][:44], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
                                          r.get("LDS Size"), r.get("Occupancy")))
