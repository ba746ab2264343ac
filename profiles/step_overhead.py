"""Where does the time between two builds go?  Times job.run() from Python against the library's own ms_total."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cora_amd import hip, synth
hip.init(0)
tmp = tempfile.mkdtemp()
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10); synth.write_round_robin_splits(par, 4, 8)
image, W, geom = synth.fasta_image(synth.workload_chr20_like(20, 64_000_000), 60)
job = hip.Job(50, 2, phys, par); job.set_reference(image, W, geom)
for _ in range(3): job.run()
L = hip.lib()
outer, inner = [], []
for _ in range(10):
    t0 = time.perf_counter(); L.cora_hip_job_run(job._h); t1 = time.perf_counter()
    outer.append((t1 - t0) * 1e3); inner.append(job.stats()["ms_total"])
print("call wall ms  ", ["%.2f" % x for x in outer])
print("library total ", ["%.2f" % x for x in inner])
