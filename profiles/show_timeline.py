#!/usr/bin/env python3
"""One build's kernels in launch order from a rocprofv3 kernel trace, with the idle time before each:
   python profiles/show_timeline.py <..._kernel_trace.csv> [build index, default last] [min gap us to flag, default 200]
A build starts at every k_pack launch."""
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
starts = [i for i, r in enumerate(rows) if "k_pack" in r["Kernel_Name"]]
lo = starts[which]
hi = starts[starts.index(lo) + 1] if starts.index(lo) + 1 < len(starts) else len(rows)
prev_end = None
busy = gaps = 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void cora::", "").replace("cora::", "")
    dur = (e - s) / 1e3
    busy += dur
    gaps += max(gap, 0.0)
    flag = "   <-- idle %.0f us before" % gap if gap >= min_gap else ""
    if dur >= 500 or flag:
        print("%10.1f us  %-52s%s" % (dur, name[:52], flag))
    prev_end = max(prev_end or e, e)
print("kernels busy %.2f ms, idle between kernels %.2f ms, span %.2f ms" % (busy / 1e3, gaps / 1e3, (prev_end - int(rows[lo]["Start_Timestamp"])) / 1e6))
