#!/usr/bin/env python3
"""Sums the counters of a rocprofv3 --pmc CSV per kernel:  python profiles/show_pmc.py <counter_collection.csv> [name filter]"""
import csv
import sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void cora::", "").replace("cora::", "")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    a = acc.setdefault(name, {})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["_n"] = a.get("_n", 0) + (1 if r["Counter_Name"] == "SQ_WAVE_CYCLES" else 0)
for name, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:40]:
    print(name[:60], "launches", a.pop("_n"))
    for k, v in sorted(a.items()):
        print("    %-24s %.4g" % (k, v))
