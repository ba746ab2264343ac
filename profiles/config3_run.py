#!/usr/bin/env python3
"""BASELINE configs[3] at its stated size (K = 64, d = 3, -t 24 on the GRCh38-sized reference): build time, stage times and the most
pool memory in use at once.  Correctness of this configuration is tests/test_gpu_large.py's; this prints the numbers DESIGN.md quotes.
    python profiles/config3_run.py [scale]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
S = os.path.join(ROOT, "tests", "golden", "splits")
contigs = synth.workload_grch38_like(38, scale)
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
hip.init(0)
job = hip.Job(64, 3, os.path.join(S, "phys_10.txt"), os.path.join(S, "par_24.txt"))
job.set_reference(image, W, geom)
for i in range(2):
    t0 = time.time()
    job.run()
    dt = time.time() - t0
    st = job.stats()
    print("run %d: %.3f s  kmers %d pairs %d comparisons %d  stages %s  pool peak %.1f GB" % (
        i, dt, st["kmers"], st["hom_pairs"], st["candidate_pairs"],
        {k: round(v, 1) for k, v in st.items() if k.startswith("ms_")}, st["pool_peak_bytes"] / 1e9), flush=True)
