#!/usr/bin/env python3
"""Compiler resource table of every kernel in csrc/ (profiles/rNN_kernel_resources.md): compiles each translation unit with the
flags of csrc/build.sh + -Rpass-analysis=kernel-resource-usage (objects go to a scratch directory, the in-tree build is not
touched) and prints the markdown table.  Usage: python tools/kernel_resources.py > profiles/r03_kernel_resources.md"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc")
units = sorted(f[:-4] for f in os.listdir(SRC) if f.endswith(".hip"))
tmp = tempfile.mkdtemp()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::|^void ", "", o).split("(")[0] for o in out]


def one(u):
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-parameter",
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, u + ".hip"), "-o", os.path.join(tmp, u + ".o")],
                       capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "unit": u}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\]| \[waves/SIMD\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


with ThreadPoolExecutor(6) as ex:
    rows = [r for rs in ex.map(one, units) for r in rs]
names = demangle([r["name"] for r in rows])
print("Compiler resource usage, gfx950, `hipcc -O3 -Rpass-analysis=kernel-resource-usage` (round 3, the flags of csrc/build.sh;\n"
      "`tools/kernel_resources.py`).  Dynamic LDS is set at launch and not in this table.\n")
print("| kernel (file) | VGPRs | scratch B/lane | SGPR spills | VGPR spills | waves/SIMD | static LDS B |")
print("|---|---|---|---|---|---|---|")
seen = set()
for r, nm in sorted(zip(rows, names), key=lambda t: (t[0]["unit"], t[1])):
    key = (r["unit"], nm)
    if key in seen:
        continue
    seen.add(key)
    print(f"| `{nm}` ({r['unit']}.hip) | {r.get('VGPRs', '')} | {r.get('ScratchSize', '')} | {r.get('SGPRs Spill', '')} | {r.get('VGPRs Spill', '')} | "
          f"{r.get('Occupancy', '')} | {r.get('LDS Size', '')} |")
