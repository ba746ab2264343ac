"""bench.py's contract on the GPU box: `--gpus N` really produces N ranks (rehearsed over gloo with the ranks sharing the
one GPU), and the N=1 line carries the table check, the end-to-end executable figure and the hashes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks_and_they_build_one_table_pair():
    one = _bench("--gpus", "1", "--scale", "0.004", "--steps", "1", "--warmup", "1", "--no-extras")
    two = _bench("--gpus", "2", "--backend", "gloo", "--scale", "0.004", "--steps", "1", "--warmup", "1", "--no-extras")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["world_size"] == 2
    assert two["scaling"] == "strong" and two["config"]["sharded_flow_failure"] is None
    # the same fixed reference whatever N: the ranks' records add up to the single-GPU run's, and so do the results
    for k in ("kmers", "hom_pairs", "inexact_nodes", "exact_nodes"):
        assert one["config"][k] == two["config"][k], k
    assert one["roofline"]["frac"] > 0 and one["cpu_baseline"] is None
    # every rank reports where its last step's time went and what it sent to the other rank (a bad scaling curve must be readable from the line)
    assert [r["rank"] for r in two["ranks"]] == [0, 1] and "ranks" not in one
    for r in two["ranks"]:
        assert r["flow_ms"]["seed_sort_verify"] > 0 and r["library_stage_ms"]["ms_total"] > 0 and r["kept_own"] == two["config"]["kept_own"]
        assert r["sent_bytes"]["items"] >= 0 and r["kmers_owned"] > 0
    assert sum(r["kmers_owned"] for r in two["ranks"]) == one["config"]["kmers"]


def test_a_failed_sharded_flow_is_an_error_unless_replicas_are_allowed():
    """Round-2 review: the first real multi-GPU run must not quietly turn into a replica measurement."""
    env = dict(os.environ, CORA_BENCH_FAIL_SHARDED="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--scale", "0.004", "--steps", "1", "--warmup", "1", "--no-extras"]
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 3 and "sharded flow failed" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run(args + ["--allow-replicas"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and "CORA_BENCH_FAIL_SHARDED" in out["config"]["sharded_flow_failure"]


def test_default_line_checks_its_tables_and_times_the_executable():
    out = _bench("--scale", "0.004", "--steps", "2", "--warmup", "1", "--cpu-sample", "400000", "--collapse-reads", "0")
    assert out["config"]["workload"].startswith("grch38_like scale 0.004")
    assert "LPT_2.dat line 10" in out["config"]["workload"]
    t = out["tables_of_last_step"]
    assert t["table_check"] == "consistent", t
    assert len(t["exact_sha256"]) == 64 and len(t["inexact_sha256"]) == 64
    e = out["e2e"]
    assert e["exit"] == 0 and e["same_bytes_as_timed_step"] is True and e["k_mers_per_s"] > 0
    assert set(e["stage_ms"]) >= {"device_init", "read+upload+scan", "write_files", "total"}
    cb = out["cpu_baseline"]
    assert cb and cb["value"] > 0 and cb["kind"] in ("reference", "port")
    assert out["secondary"]["workload"].startswith("chr20_like 64000000")
    for k in ("roofline", "stage_ms"):
        assert k in out
