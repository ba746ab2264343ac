#!/usr/bin/env python3
"""Stage times of the first runs of a fresh job (the second run used to be slower than the third)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cora_amd import hip, synth
hip.init(0)
tmp = tempfile.mkdtemp()
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10); synth.write_round_robin_splits(par, 4, 8)
image, W, geom = synth.fasta_image(synth.workload_chr20_like(20, 64_000_000), 60)
job = hip.Job(50, 2, phys, par); job.set_reference(image, W, geom)
for i in range(5):
    t0 = time.perf_counter(); job.run(); w = (time.perf_counter() - t0) * 1e3
    st = job.stats()
    print("run %d wall %.2f" % (i, w), {k: round(st[k], 2) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")}, flush=True)
