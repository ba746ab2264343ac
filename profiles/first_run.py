#!/usr/bin/env python3
"""First build of a process against the steady state: where does the executable's end-to-end time go?
    CORA_TRACE=1 python profiles/first_run.py [scale]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
S = os.path.join(ROOT, "tests", "golden", "splits")
t0 = time.time()
contigs = synth.workload_grch38_like(38, scale)
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
print("generated in %.1f s" % (time.time() - t0), flush=True)
t0 = time.time(); hip.init(0); print("init %.3f s" % (time.time() - t0), flush=True)
job = hip.Job(50, 2, os.path.join(S, "phys_10.txt"), os.path.join(S, "par_8.txt"))
t0 = time.time(); job.set_reference(image, W, geom); print("set_reference %.3f s" % (time.time() - t0), flush=True)
for i in range(3):
    t0 = time.time(); job.run(); print("run %d: %.3f s" % (i, time.time() - t0), flush=True)
