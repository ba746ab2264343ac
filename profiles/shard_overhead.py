import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cora_amd import hip, synth, dist as cdist
hip.init(0)
tmp = tempfile.mkdtemp()
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10); synth.write_round_robin_splits(par, 4, 8)
contigs = synth.workload_chr20_like(20, 64_000_000)
image, W, geom = synth.fasta_image(contigs, 60)
job = hip.Job(50, 2, phys, par); job.set_reference(image, W, geom)
sj = cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0)))
for mode, run in (("plain", job.run), ("sharded world 1", sj.run), ("plain", job.run), ("sharded world 1", sj.run)):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    print(mode, "%.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
