"""Sharded (multi-GPU) flow on the GPU box.  The box has ONE GPU, so: (a) world = 1 through the sharded
stage functions must equal the plain run byte for byte; (b) two ranks sharing the GPU exchange through gloo
(host staging), which runs every kernel and every exchange of the N > 1 path except the RCCL transport."""
import os
import subprocess
import sys

import pytest

import helpers
from cora_amd import hip

pytestmark = pytest.mark.gpu
CASES = ["syn_k50_d2", "lowcx_k50_d2", "fam_k48_d3", "multi_k50_d2_L", "q_k50_d2_p10", "pal_k50_d2_p4t5", "ecoli_like_k50_d2"]


@pytest.mark.parametrize("name", CASES)
def test_sharded_world1_equals_reference(name):
    import torch
    from cora_amd import dist as cdist
    hip.init(0)
    c, want = helpers.golden_case(name)
    job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]), bool(c["L"]))
    job.set_reference_file(helpers.input_path(c["input"]))
    cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0))).run()
    e, en, i, inn = job.tables()
    assert e == helpers.read(os.path.join(want, "Exact")) and i == helpers.read(os.path.join(want, "Inexact"))
    assert (en, inn) == (c["exact_nodes"], c["inexact_nodes"])
    job.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranks_over_gloo_equal_reference(world, tmp_path):
    script = os.path.join(helpers.ROOT, "tests", "support", "sharded_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world), CORA_TEST_CASES=",".join(CASES),
               CORA_TEST_OUT=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    for name in CASES:
        c, want = helpers.golden_case(name)
        assert helpers.read(os.path.join(str(tmp_path), name + ".Exact")) == helpers.read(os.path.join(want, "Exact")), name
        assert helpers.read(os.path.join(str(tmp_path), name + ".Inexact")) == helpers.read(os.path.join(want, "Inexact")), name
    # the <Inexact> stage really was split: every later rank wrote part of the big case's table
    for r in range(1, world):
        assert int(open(os.path.join(str(tmp_path), "ecoli_like_k50_d2.rank%d.bytes" % r)).read()) > 0


def test_rccl_collectives_single_rank():
    """The exchange calls exactly as the N > 1 bench makes them (nccl backend = RCCL, device tensors, uneven
    all_to_all_single of int64 / int32 / uint8, int32 sum, int64 min), with the only world size one GPU allows."""
    script = os.path.join(helpers.ROOT, "tests", "support", "sharded_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", CORA_TEST_CASES="q_k50_d2_p10,fam_k48_d3",
               CORA_TEST_OUT=helpers.scratch(), CORA_TEST_BACKEND="nccl")
    p = subprocess.run([sys.executable, script, "0", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:]
    for name in ("q_k50_d2_p10", "fam_k48_d3"):
        c, want = helpers.golden_case(name)
        assert helpers.read(os.path.join(env["CORA_TEST_OUT"], name + ".Exact")) == helpers.read(os.path.join(want, "Exact")), name
        assert helpers.read(os.path.join(env["CORA_TEST_OUT"], name + ".Inexact")) == helpers.read(os.path.join(want, "Inexact")), name
