"""One rank of the sharded read-side chain for tests/test_gpu_collapse.py: all ranks share GPU 0 and talk through gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch
import torch.distributed as dist

import make_collapse_frag_golden as mcf
from cora_amd import dist as cdist
from cora_amd import hip

rank, world = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
hip.init(0)
comm = cdist.Comm(torch.device("cuda", 0), always_collective=True)
out = os.environ["CORA_TEST_OUT"]
for name in os.environ["CORA_TEST_CASES"].split(","):
    fcase, ref, D, mapper, Lvar = mcf.CASES[name]
    sets, mode, n, size = mcf.mfg.CASES[fcase]
    L = mcf.mfg.mcg.READ_SETS[sets[0]][1]
    imgs = []
    for s in sets:
        imgs.append(open(os.path.join(mcf.mfg.INP, s + "_1.fq"), "rb").read())
        if "PAIRED" in mode:
            imgs.append(open(os.path.join(mcf.mfg.INP, s + "_2.fq"), "rb").read())
    refimg = open(os.path.join(mcf.HERE, "inputs", ref + ".fa"), "rb").read() if ref else None
    got = cdist.read_chain_sharded(comm, mcf.mfg.frag_text(n, size).encode(), L, mode, imgs, refimg, D, mapper, Lvar)
    assert sorted(got) == [g for g in range(n) if g % world == rank]
    for g, files in got.items():
        for fn, b in zip(("reads", "perfect", "lrn"), files):
            open(os.path.join(out, "%s.%s.split%d" % (name, fn, g)), "wb").write(b)
    comm.barrier()
dist.destroy_process_group()
