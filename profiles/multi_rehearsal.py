#!/usr/bin/env python3
"""The in-process multi-GPU flow (cora_hip_job_run_multi) rehearsed on ONE GPU at a size where it matters: R ranks sharing
the device build the tables of a grch38_like reference; bytes must equal the single-rank run's, and the time says what the
sharded flow costs on top (partition passes, exchanges as device copies).   python profiles/multi_rehearsal.py [scale] [ranks] [only]
("only": skip the single-rank run -- at scale 1 its pool and the ranks' pools do not fit one GPU together)"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = os.path.join(ROOT, "tests", "golden", "splits")
contigs = synth.workload_grch38_like(38, scale)
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
hip.init(0)
job = hip.Job(50, 2, os.path.join(S, "phys_10.txt"), os.path.join(S, "par_8.txt"))
job.set_reference(image, W, geom)
only = len(sys.argv) > 3 and sys.argv[3] == "only"
print("reference ready: %d image bytes" % len(image), flush=True)
for what, run in (("single", job.run), ("%d ranks in process" % ranks, lambda: job.run_multi(ranks, [0] * ranks))):
    if only and what == "single":
        continue
    for i in range(2):
        t0 = time.time()
        run()
        dt = time.time() - t0
    e, en, x, xn = job.tables()
    st = job.stats()
    print("%-22s %.3f s  kmers %d pairs %d  exact %s inexact %s" % (what, dt, st["kmers"], st["hom_pairs"], hashlib.sha256(e).hexdigest()[:16],
                                                                  hashlib.sha256(x).hexdigest()[:16]), flush=True)
