#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv as per-build milliseconds:  python profiles/show_stats.py <csv> <builds> [top]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
builds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print("%-62s calls %5s  per build %9.3f ms  avg %9.3f ms  %5.1f%%" % (r["Name"].replace("cora::", "").replace("void ", "")[:62], r["Calls"],
          float(r["TotalDurationNs"]) / 1e6 / builds, float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("kernel time per build: %.2f ms" % (tot / 1e6 / builds))
