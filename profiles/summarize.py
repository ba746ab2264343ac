#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (under gpurun_out/<tag>_*) into the committed summaries:
profiles/<tag>_kernel_stats.csv (the profiler's own per-kernel table) and profiles/<tag>_pmc_traffic.md."""
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# the workload tag bench.py prints in config.workload before the colon, e.g. "grch38_like scale 1" or "chr20_like 64000000 bp"
workload = sys.argv[2] if len(sys.argv) > 2 else "grch38_like scale 1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def one(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    if not hits:
        sys.exit("missing " + pattern)
    return hits[0]


shutil.copy(one("%s_trace/**/*kernel_stats.csv" % tag), os.path.join(root, "profiles", "%s_kernel_stats.csv" % tag))


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "tables::").split("(")[0]
        a = acc.setdefault(name, [set(), 0.0])
        a[0].add(r["Dispatch_Id"])
        a[1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel(one("%s_fetch/**/*counter_collection.csv" % tag), "FETCH_SIZE")
write = per_kernel(one("%s_write/**/*counter_collection.csv" % tag), "WRITE_SIZE")
names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1]))
with open(os.path.join(root, "profiles", "%s_pmc_traffic.md" % tag), "w") as f:
    f.write("# %s PMC traffic per kernel, bench.py --steps 1 --warmup 0 (%s, K=50, d=2)\n\n" % (tag, workload))
    f.write("FETCH_SIZE / WRITE_SIZE collected in separate `rocprofv3 --pmc` passes (profiles/collect.sh); both count kB.\n"
            "On gfx950 FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): the `read GB`\n"
            "column doubles it.  Narrow or scattered accesses are uncalibrated, so read small kernels' rows as ratios.\n\n")
    f.write("| kernel | launches | FETCH_SIZE sum kB | read GB (x2) | WRITE_SIZE sum kB | write GB |\n|---|---|---|---|---|---|\n")
    for k in names:
        fs, ws = fetch.get(k, [set(), 0.0]), write.get(k, [set(), 0.0])
        f.write("| %s | %d | %.0f | %.3f | %.0f | %.3f |\n" % (k, max(len(fs[0]), len(ws[0])), fs[1], 2 * fs[1] * 1024 / 1e9, ws[1], ws[1] * 1024 / 1e9))
# the dominant kernel family (every rs_scatter launch of one build), per launch: what bench.py reports as roofline.traffic
import json
sc = [k for k in names if "rs_scatter" in k]
launches = sum(max(len(fetch.get(k, [set()])[0]), len(write.get(k, [set()])[0])) for k in sc)
rd = sum(2 * fetch.get(k, [0, 0.0])[1] * 1024 for k in sc)
wr = sum(write.get(k, [0, 0.0])[1] * 1024 for k in sc)
json.dump({"kernel": "rs_scatter", "workload": "%s K=50 d=2" % workload, "launches": launches, "read_bytes": rd, "write_bytes": wr,
           "bytes_per_launch": (rd + wr) / max(1, launches), "source": "profiles/%s_pmc_traffic.md" % tag},
          open(os.path.join(root, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
print("wrote profiles/%s_kernel_stats.csv and profiles/%s_pmc_traffic.md" % (tag, tag))
