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


@pytest.mark.parametrize("world,keep_own", [(2, None), (3, None), (2, "0"), (3, "0"), (5, "1")])
def test_sharded_ranks_over_gloo_equal_reference(world, keep_own, tmp_path):
    """keep_own: how the K-mer and seed records reach their owner (shard.hip) -- None = the library's choice (2 to 4 ranks:
    every rank scans everything and keeps its own), "0" = the all-to-all forced, "1" = keeping forced at a world that
    would exchange."""
    script = os.path.join(helpers.ROOT, "tests", "support", "sharded_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + (10 if keep_own else 0)), CORA_TEST_CASES=",".join(CASES),
               CORA_TEST_OUT=str(tmp_path))
    env.pop("CORA_SHARD_FILTER", None)
    if keep_own is not None:
        env["CORA_SHARD_FILTER"] = keep_own
    if world == 3:
        env["CORA_SKETCH"] = "1"       # the seed records carry sketches (what a large reference switches on): one more slot to move
        env["CORA_B_CHAIN"] = "3"      # ... and their second word climbs through a chain of arrays during the sort, as in a rank's arena
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
    # and the records came the way the test asked for
    want_kept = (world <= 4) if keep_own is None else keep_own == "1"
    assert open(os.path.join(str(tmp_path), "ecoli_like_k50_d2.kept_own")).read().strip() == str(int(want_kept))


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


# ------------------------------------------------------------------ the same flow inside ONE process (cora_hip_job_run_multi)
@pytest.mark.parametrize("world,sketch,keep_own", [(1, None, None), (2, None, None), (2, None, "0"), (3, "1", None), (3, "1", "0"), (4, None, None),
                                                   (8, None, None), (8, "1", None), (8, "1", "1"), (8, None, "1")])
def test_in_process_ranks_equal_reference(world, sketch, keep_own, monkeypatch):
    """One host thread per rank, records pulled rank to rank by device copies: the flow the executable runs on a node with
    several GPUs, rehearsed with every rank on the one GPU.  Same bytes as the reference, same whole-job counts as the
    single-GPU run, on two calls in a row (the ranks' replicas are kept between calls).  keep_own as in the test above."""
    hip.init(0)
    monkeypatch.delenv("CORA_SHARD_FILTER", raising=False)
    if keep_own is not None:
        monkeypatch.setenv("CORA_SHARD_FILTER", keep_own)
    if sketch:
        monkeypatch.setenv("CORA_SKETCH", sketch)      # sketches travel with the seed records, as on a large reference
        monkeypatch.setenv("CORA_B_CHAIN", "3" if world == 3 else "5")     # ... through the chained sort a rank with an arena runs (kept and exchanged records)
    for name in CASES:
        c, want = helpers.golden_case(name)
        job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]), bool(c["L"]))
        job.set_reference_file(helpers.input_path(c["input"]))
        job.run()
        ref = job.stats()
        for _ in range(2):
            job.run_multi(world, [0] * world)
            e, en, i, inn = job.tables()
            assert e == helpers.read(os.path.join(want, "Exact")) and i == helpers.read(os.path.join(want, "Inexact")), (name, world)
            assert (en, inn) == (c["exact_nodes"], c["inexact_nodes"])
            st = job.stats()
            for k in ("kmers", "representatives", "exact_classes", "hom_pairs", "inexact_items", "inexact_nodes", "exact_nodes", "candidate_pairs"):
                assert st[k] == ref[k], (name, world, k)
        job.close()


def test_in_process_ranks_refuse_bad_devices():
    hip.init(0)
    c, _ = helpers.golden_case("q_k50_d2")
    job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]))
    job.set_reference_file(helpers.input_path(c["input"]))
    with pytest.raises(hip.CoraHipError):
        job.run_multi(2, [0, 99])
    with pytest.raises(hip.CoraHipError):
        job.run_multi(0)
    job.run_multi(2, [0, 0])                          # and the job is still usable
    job.close()


def test_a_failed_rank_setup_leaves_no_half_built_team(monkeypatch):
    """Round-2 advice: when the set-up of some rank's clone fails, the job must not keep the team; the next call builds it afresh."""
    hip.init(0)
    c, want = helpers.golden_case("q_k50_d2")
    job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]))
    job.set_reference_file(helpers.input_path(c["input"]))
    monkeypatch.setenv("CORA_TEST_FAIL_CLONE", "2")
    with pytest.raises(hip.CoraHipError):
        job.run_multi(3, [0, 0, 0])
    monkeypatch.delenv("CORA_TEST_FAIL_CLONE")
    job.run_multi(3, [0, 0, 0])                       # same device list: must not find (and run on) the clones of the failed call
    e, en, i, inn = job.tables()
    assert e == helpers.read(os.path.join(want, "Exact")) and i == helpers.read(os.path.join(want, "Inexact"))
    job.close()


def test_executable_falls_back_to_one_gpu_when_the_ranks_fail(tmp_path):
    """The multi-GPU flow of the executable has never run on more than one real GPU: when it fails, homTable_setup says why and
    builds on one GPU instead of failing `cora coraIndex` (round-2 advice)."""
    c, want = helpers.golden_case("syn_k50_d2_p10t8")
    exe = os.path.join(hip.BIN_DIR, "homTable_setup")
    ex, inx = str(tmp_path / "E"), str(tmp_path / "I")
    r = subprocess.run([exe, helpers.input_path(c["input"]), str(c["K"]), helpers.split_path("phys", c["p"]), ex, inx,
                        helpers.split_path("par", c["t"]), "BOTH", str(c["d"]), "FULL"], env=dict(os.environ, CORA_RANKS="3", CORA_TEST_FAIL_CLONE="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "falling back to one GPU" in r.stdout and "CORA_TEST_FAIL_CLONE" in r.stdout
    assert helpers.read(ex) == helpers.read(os.path.join(want, "Exact")) and helpers.read(inx) == helpers.read(os.path.join(want, "Inexact"))


@pytest.mark.parametrize("ranks", [2, 5])
def test_executable_builds_with_several_ranks(ranks, tmp_path):
    """CORA_RANKS forces the executable's multi-GPU flow (it would pick it by itself for a reference above 256 Mbp on a
    node with several GPUs): same four files."""
    for name in ("syn_k50_d2_p10t8", "multi_k50_d2_L"):
        c, want = helpers.golden_case(name)
        exe = os.path.join(hip.BIN_DIR, "homTable_setup_L" if c["L"] else "homTable_setup")
        ex, inx = str(tmp_path / (name + ".E")), str(tmp_path / (name + ".I"))
        r = subprocess.run([exe, helpers.input_path(c["input"]), str(c["K"]), helpers.split_path("phys", c["p"]), ex, inx,
                            helpers.split_path("par", c["t"]), "BOTH", str(c["d"]), "FULL"], env=dict(os.environ, CORA_RANKS=str(ranks)),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-2000:]
        assert "Building on %d ranks" % ranks in r.stdout
        assert helpers.read(ex) == helpers.read(os.path.join(want, "Exact")) and helpers.read(inx) == helpers.read(os.path.join(want, "Inexact"))
        assert helpers.read(ex + ".itemCount") == helpers.read(os.path.join(want, "Exact.itemCount"))
        assert helpers.read(inx + ".itemCount") == helpers.read(os.path.join(want, "Inexact.itemCount"))
