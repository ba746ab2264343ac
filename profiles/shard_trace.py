#!/usr/bin/env python3
"""Where a sharded build spends its time at world 1 (no exchange cost: what is left is the stage structure itself).
Wraps ShardedJob's steps with wall-clock timers (each step ends synchronised).   python profiles/shard_trace.py"""
import collections
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cora_amd import hip, synth, dist as cdist  # noqa: E402

hip.init(0)
tmp = tempfile.mkdtemp()
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10)
synth.write_round_robin_splits(par, 4, 8)
image, W, geom = synth.fasta_image(synth.workload_chr20_like(20, 64_000_000), 60)
job = hip.Job(50, 2, phys, par)
job.set_reference(image, W, geom)
sj = cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0)))
acc = collections.OrderedDict()
STAGE = {v: k for k, v in vars(cdist).items() if k.startswith("ST_")}


def timed(name_fn, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        key = name_fn(*a)
        acc[key] = acc.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    return w


sj._stage = timed(lambda st, arg=0: "stage %s" % STAGE.get(st, st), sj._stage)
sj._route = timed(lambda pairs, counts: "route x%d" % len(pairs), sj._route)
sj._gather_to_root = timed(lambda pairs: "gather x%d" % len(pairs), sj._gather_to_root)
_imp, _exp = sj._import, sj._export
for _ in range(2):
    sj.run()
acc.clear()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    sj.run()
total = (time.perf_counter() - t0) * 1e3 / N
for k, v in acc.items():
    print("%-24s %7.3f ms" % (k, v / N))
print("%-24s %7.3f ms (sum of the above %.3f)" % ("whole run", total, sum(acc.values()) / N))
job.run()
print("plain run: ms_total %.3f" % job.stats()["ms_total"])
