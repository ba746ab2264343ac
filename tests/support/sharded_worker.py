"""One rank of the sharded flow for tests/test_gpu_sharded.py: all ranks share GPU 0 and talk through gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist

import helpers
from cora_amd import dist as cdist
from cora_amd import hip

rank, world = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("CORA_TEST_BACKEND", "gloo"), rank=rank, world_size=world)
hip.init(0)
comm = cdist.Comm(torch.device("cuda", 0), always_collective=True)
for name in os.environ["CORA_TEST_CASES"].split(","):
    c, want = helpers.golden_case(name)
    job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]), bool(c["L"]))
    job.set_reference_file(helpers.input_path(c["input"]))
    sj = cdist.ShardedJob(job, comm)
    sj.run()
    if rank == 0:
        open(os.path.join(os.environ["CORA_TEST_OUT"], name + ".kept_own"), "w").write(str(int(sj.kept_own)))
    if rank > 0:                                   # a later rank keeps the bytes of its own source interval
        open(os.path.join(os.environ["CORA_TEST_OUT"], "%s.rank%d.bytes" % (name, rank)), "w").write(str(len(job.tables()[2])))
    if rank == 0:
        e, en, i, inn = job.tables()
        out = os.environ["CORA_TEST_OUT"]
        open(os.path.join(out, name + ".Exact"), "wb").write(e)
        open(os.path.join(out, name + ".Inexact"), "wb").write(i)
    job.close()
    comm.barrier()
dist.destroy_process_group()
