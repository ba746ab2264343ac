"""world_size-2 (and 3) gloo rehearsal on CPU of the exchange layer the multi-GPU build uses."""
import os
import subprocess
import sys

import pytest

import helpers


@pytest.mark.parametrize("world", [2, 3])
def test_comm_over_gloo(world):
    script = os.path.join(helpers.ROOT, "tests", "support", "comm_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world))
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=170)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


def test_sharded_job_needs_the_library():
    """No CPU path: the sharded driver goes through the same C-ABI, which refuses to start without a GPU."""
    from cora_amd import hip
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.CoraHipError):
        hip.init(0)


def test_lrn_offsets_follow_the_group_order():
    """cora.cpp:1762-1776: every group's long read names are numbered on from the groups before it."""
    from cora_amd import dist as cdist
    assert cdist.lrn_offsets(0, [3, 0, 2, 5]) == [0, 3, 3, 5]
    assert cdist.lrn_offsets(7, [0, 0]) == [7, 7]
    assert cdist.lrn_offsets(0, []) == []
