#!/usr/bin/env python3
"""Benchmark of the homology-table hot path (BASELINE.json: reference k-mers/s into the d=2 table, GRCh38).

    python bench.py --gpus N --steps K --warmup W

A step is one complete table build (2-bit pack -> canonical k-mers -> radix sort -> classes -> seeds -> pair
verification -> block merge -> both tables as bytes in host memory) over a synthetic reference whose FASTA image is
already resident in HBM when the timed region starts.  The workload is BASELINE configs[2] as SURVEY.md section 8d restates
it (`--workload grch38_like`): 3.09 Gbp in 25 contigs with the GRCh38 primary-assembly lengths, 5 % N in runs of >= 10 kbp,
45 % of the bases from repeat families (cora_amd/synth.py: workload_grch38_like says how the copy numbers are capped and
why), K=50, d=2, cora's default split files (`-p 10 -t 8`: lines 10 and 8 of LPT_2.dat / LPT_4.dat).

N > 1 is STRONG scaling on that same fixed reference: every rank holds the packed reference and owns 1/N of the key
space.  On 2 to 4 GPUs (one xGMI link per GPU pair) every rank scans all positions and keeps the records it owns; above
that it scans 1/N of the positions and the records move by all-to-all over RCCL (cora_amd/dist.py, shard.hip).  Rank 0 ends
up with both tables.  Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the script starts its N ranks itself,
before anything touches a GPU; under `python -m torch.distributed.run` it is one of the ranks.

Outside the timed region, on rank 0 at N=1: the last step's tables are written out, checked record by record against the
bases by oracle/table_check and hashed; the drop-in executable is timed end to end on the same FASTA file (process start
-> four files closed) beside the real reference on a bounded sample (cpu_baseline); the 64 Mbp chr20-sized configuration
(configs[1], round 1's headline) is kept as `secondary`.  One JSON line on rank 0."""
import argparse
import datetime
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
SPLITS = os.path.join(ROOT, "tests", "golden", "splits")       # lines of the reference's LPT_2.dat / LPT_4.dat as cora.cpp:305,311 rewrites them


def make_workload(name, scale, bases, seed_shift=0):
    from cora_amd import synth
    if name == "grch38_like":
        return synth.workload_grch38_like(38 + seed_shift, scale), "grch38_like scale %g" % scale
    return synth.workload_chr20_like(20 + seed_shift, bases), "chr20_like %d bp" % bases


def write_fasta_image(path, image, geom, names, width):
    with open(path, "wb") as f:
        f.write(memoryview(image))
    with open(path + ".fai", "w") as f:
        for name, (off, n) in zip(names, geom):
            f.write("%s\t%d\t%d\t%d\t%d\n" % (name, n, off, width, width + 1))


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def cpu_baseline(workload, K, d, sample_bases, tmp):
    """The real reference (oracle/_ref, kind 'reference') or the C oracle ('port') on a bounded sample of the same
    workload generator, timed on this box's host cores from process start to exit (both tables closed)."""
    import numpy as np
    from cora_amd import synth
    fa = os.path.join(tmp, "sample.fa")
    if workload == "grch38_like":
        contigs = synth.workload_grch38_like(38, sample_bases / float(sum(synth.GRCH38_LENGTHS)))
        what = "grch38_like generator scaled to %d bp" % sum(len(s) for _, s in contigs)
    else:
        contigs = synth.workload_chr20_like(20, sample_bases)
        what = "chr20_like generator at %d bp" % sample_bases
    synth.write_fasta(fa, contigs, 60)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "homTable_setup")
    oracle_bin = os.path.join(ROOT, "oracle", "hts_oracle")
    if os.path.exists(ref_bin):
        kind, exe, threads = "reference", ref_bin, 8
    else:
        if not os.path.exists(oracle_bin):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "hts_oracle"], stdout=subprocess.DEVNULL)
        kind, exe, threads = "port", oracle_bin, 1
    # -p 1: the reference's fastest setting (every physical split costs it 2 s of hash-table zeroing)
    phys, par = os.path.join(SPLITS, "phys_1.txt"), os.path.join(SPLITS, "par_%d.txt" % threads)
    env = dict(os.environ, OMP_DYNAMIC="false", OMP_NUM_THREADS=str(threads))
    env.pop("OMP_THREAD_LIMIT", None)
    out = os.path.join(tmp, "cpu")
    os.makedirs(out, exist_ok=True)
    cmd = [exe, fa, str(K), phys, os.path.join(out, "E"), os.path.join(out, "I"), par, "BOTH", str(d), "FULL"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    if r.returncode != 0:
        return None
    nonN = int(sum((np.isin(s, np.frombuffer(b"ACGTacgt", dtype=np.uint8))).sum() for _, s in contigs))
    return {"value": nonN / dt, "unit": "k-mers/s", "cores": threads, "kind": kind,
            "sample": "%s (K=%d d=%d, -p 1 -t %d), process wall %.1f s" % (what, K, d, threads, dt)}


def pmc_traffic(workload_tag, K, d):
    """HBM bytes per rs_scatter launch from the newest committed PMC summary of this very workload
    (profiles/collect.sh + summarize.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for gfx950)."""
    import glob
    want = "%s K=%d d=%d" % (workload_tag, K, d)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("workload") == want:
            return t["bytes_per_launch"], t.get("source")
    return None, None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh children (nothing here has touched a GPU),
    relay their output and leave with the worst return code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    sys.exit(rc)


def run_steps(run, job, steps, sync):
    t0 = time.perf_counter()
    sc_ms, sc_n, sc_bytes, st = 0.0, 0, 0, None
    for _ in range(steps):
        ts = time.perf_counter()
        run()
        st = job.stats()
        if os.environ.get("CORA_TRACE") or os.environ.get("RANK", "0") == "0":       # (a long run must not look hung)
            print("[bench] step wall %.3f ms  %s" % ((time.perf_counter() - ts) * 1e3, " ".join("%s=%.2f" % (k[3:], st[k]) for k in (
                "ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish"))), file=sys.stderr, flush=True)
        sc_ms += st["ms_sort_scatter"]; sc_n += st["n_sort_scatter"]; sc_bytes += st["sort_scatter_bytes"]
    sync()
    return time.perf_counter() - t0, st, (sc_ms, sc_n, sc_bytes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["grch38_like", "chr20_like"], default="grch38_like")
    ap.add_argument("--scale", type=float, default=1.0, help="grch38_like: contig lengths x scale (1.0 = 3.09 Gbp)")
    ap.add_argument("--bases", type=int, default=64_000_000, help="chr20_like: reference length")
    ap.add_argument("--cpu-sample", type=int, default=16_000_000, help="bases of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip table_check, e2e, cpu_baseline and the secondary configuration")
    ap.add_argument("--collapse-reads", type=int, default=4_000_000, help="reads of the read-side figures of the secondary block (0 = skip)")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded", help="what N>1 ranks do (see the module text)")
    ap.add_argument("--allow-replicas", action="store_true", help="N>1: if the sharded flow fails, measure N independent replicas instead of leaving with an error")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo = host-staged rehearsal, several ranks may share a GPU)")
    ap.add_argument("--tmp", default=os.environ.get("CORA_TMP", tempfile.gettempdir()))
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus)                                  # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print("[bench] --gpus %d but WORLD_SIZE=%d: the launcher decides; measuring %d rank(s)" % (a.gpus, world, world), file=sys.stderr, flush=True)
    import torch
    import torch.distributed as dist
    local_rank %= max(1, torch.cuda.device_count())      # rehearsals put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    # PyTorch creates its HIP context lazily; done inside the first sync() below it made the first TIMED step slower.
    torch.zeros(1, device="cuda").add_(1)
    torch.cuda.synchronize()
    side = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a finite timeout: a rank that fails inside the sharded flow leaves the others blocked in a collective; they then
        # raise instead of waiting for ever (and the launcher tears the job down)
        dist.init_process_group(a.backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
        side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
        assert dist.get_world_size() == world

    from cora_amd import hip, synth
    K, d = 50, 2
    hip.init(local_rank)
    tmp = tempfile.mkdtemp(prefix="cora_bench_", dir=a.tmp)
    phys, par = os.path.join(SPLITS, "phys_10.txt"), os.path.join(SPLITS, "par_8.txt")     # cora's defaults: -p 10 -t 8
    sharded = world > 1 and a.mode == "sharded"
    t_gen = time.time()
    # sharded: ONE fixed reference, the same on every rank (strong scaling); replicas: an independent one per rank
    contigs, tag = make_workload(a.workload, a.scale, a.bases, 0 if (sharded or world == 1) else rank)
    names = [n for n, _ in contigs]
    image, W, geom = synth.fasta_image(contigs, 60)
    total_bases = int(sum(len(s) for _, s in contigs))
    del contigs
    t_gen = time.time() - t_gen
    if rank == 0:
        print("[bench] %s ready after %.1f s (%d ranks)" % (tag, t_gen, world), file=sys.stderr, flush=True)
    e2e = None
    if world == 1 and not a.no_extras:
        fa_path = os.path.join(tmp, "ref.fa")
        write_fasta_image(fa_path, image, geom, names, W)
        e2e = e2e_executable(fa_path, K, d, phys, par, tmp)
    job = hip.Job(K, d, phys, par)
    job.set_reference(image, W, geom)               # H2D copy: the image is HBM-resident from here on
    if sharded:
        from cora_amd import dist as cdist
        sj = cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", local_rank)))
        run = sj.run
    else:
        run = job.run

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Warm-up.  The sharded flow over RCCL can only be rehearsed with one rank per GPU, i.e. not on a one-GPU box.  If a rank fails
    # in it, all ranks agree (over gloo) and LEAVE with status 3 and the cause on stderr: a replica figure would answer no question
    # about the sharded build.  --allow-replicas: measure independent replicas instead, and the line says so.
    fallback = None
    try:
        if sharded and os.environ.get("CORA_BENCH_FAIL_SHARDED"):      # rehearsal of the fall-back itself
            raise RuntimeError("CORA_BENCH_FAIL_SHARDED is set")
        for _ in range(a.warmup):
            tw = time.perf_counter()
            run()
            if rank == 0:
                print("[bench] warm-up step %.3f s" % (time.perf_counter() - tw), file=sys.stderr, flush=True)
        sync()
    except Exception as e:                            # noqa: BLE001
        if not sharded:
            raise
        fallback = "%s: %s" % (type(e).__name__, str(e)[:300])
    if sharded:
        flag = torch.tensor([1 if fallback else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=side)
        if int(flag.item()):
            fallback = fallback or "another rank failed in the sharded flow"
            if not a.allow_replicas:
                print("[bench] rank %d: sharded flow failed (%s); no figure (--allow-replicas would measure replicas instead)" % (rank, fallback),
                      file=sys.stderr, flush=True)
                job.close()
                dist.barrier(group=side)
                dist.destroy_process_group()
                sys.exit(3)
            print("[bench] rank %d: sharded flow failed (%s); measuring replicas" % (rank, fallback), file=sys.stderr, flush=True)
            sharded = False
            job.close()
            job = hip.Job(K, d, phys, par)
            job.set_reference(image, W, geom)
            run = job.run
            for _ in range(a.warmup):
                run()
            sync()
    dt, st, (sc_ms, sc_n, sc_bytes) = run_steps(run, job, a.steps, sync)
    if world > 1:
        dev = "cuda" if a.backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        kt = torch.tensor([float(st["kmers"])], device=dev, dtype=torch.float64)   # sharded: the records this rank owned
        dist.all_reduce(kt, op=dist.ReduceOp.SUM)
        kmers_all = float(kt.item())
    else:
        kmers_all = float(st["kmers"])
    # what every rank spent where in its last step, and what it sent to other ranks: a bad scaling curve can be read from the line
    rank_reports = None
    if world > 1:
        mine = {"rank": rank, "device": local_rank, "kmers_owned": int(st["kmers"]), "representatives_owned": int(st["representatives"]),
                "library_stage_ms": {k: round(st[k], 3) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")},
                "rs_scatter_ms": round(st["ms_sort_scatter"], 3)}
        if sharded and sj.report:
            mine.update({"kept_own": sj.report["kept_own"], "flow_ms": sj.report["ms"], "sent_bytes": sj.report["sent_bytes"]})
        rank_reports = [None] * world if rank == 0 else None
        dist.gather_object(mine, rank_reports, dst=0, group=side)

    if rank == 0:
        achieved = (sc_bytes / 1e9) / (sc_ms / 1e3) if sc_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(tag, K, d) if world == 1 else (None, None)
        if sharded:
            par_txt = ("sharded (strong scaling): key space of the one reference split over %d GPUs; %s; items, bitmaps and table parts over %s"
                       % (world, "every rank scans all positions and keeps the records it owns (2 to 4 GPUs: one xGMI link per pair)" if sj.kept_own
                          else "1/%d of the positions scanned per rank, records to their owner by all-to-all" % world, a.backend))
        elif world > 1:
            par_txt = "replicas: 1 independent reference per GPU"
        else:
            par_txt = "1 GPU"
        out = {
            "metric": "ref k-mers/sec into homology table (d=2)", "value": kmers_all * a.steps / dt, "unit": "k-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (sharded or world == 1) else "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s: %d bp in %d contigs%s, K=%d, d=%d, split files = LPT_2.dat line 10 / LPT_4.dat line 8 (cora -p 10 -t 8), "
                                   "60-column FASTA image resident in HBM; output = both tables as bytes in %shost memory"
                                   % (tag, total_bases, len(names), " (GRCh38 primary-assembly lengths, 5 % N in >= 10 kbp runs, 45 % of bases from "
                                      "capped repeat families: synth.workload_grch38_like)" if a.workload == "grch38_like" else "", K, d,
                                      "rank 0's " if world > 1 else ""),
                       "kmers": int(kmers_all), "kmers_rank0": int(st["kmers"]), "representatives_rank0": int(st["representatives"]),
                       "hom_pairs": int(st["hom_pairs"]),
                       "candidate_pairs": int(st["candidate_pairs"]), "inexact_nodes": int(st["inexact_nodes"]),
                       "exact_nodes": int(st["exact_nodes"]), "hash_mixed_runs": int(st["hash_mixed_runs"]),
                       "full_key_fallback": int(st["full_key_fallback"]), "dirty_nodes": int(st["dirty_nodes"]), "big_nodes": int(st["big_nodes"]),
                       "parallelism": par_txt, "world_size": world, "sharded_flow_failure": fallback, "generate_s": round(t_gen, 1),
                       "kept_own": bool(sharded and sj.kept_own),
                       "arena": "stage arrays cut from one allocation of at most %s GiB per GPU (library default 190, CORA_ARENA_CAP_GB; the executable of `e2e` "
                                "runs the same layout)" % os.environ.get("CORA_ARENA_CAP_GB", "190")},
            "stage_ms": {k: round(st[k], 3) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")},
            "roofline": {"bound": "hbm", "kernel": "rs_scatter (LSD radix scatter, all sorts of the path)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "launches_per_step": sc_n // max(1, a.steps), "avg_launch_ms": sc_ms / max(1, sc_n),
                         "algorithmic_bytes_per_launch": sc_bytes / max(1, sc_n), "traffic": traffic, "traffic_source": traffic_src},
        }
        if rank_reports is not None:
            out["ranks"] = rank_reports
        extras = world == 1 and not a.no_extras
        if extras:
            out.update(extras_after_timing(a, job, image, geom, names, W, K, d, phys, par, tmp, tag, st, e2e))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def e2e_executable(fa, K, d, phys, par, tmp, kmers_hint=None):
    """End to end, like for like with the CPU baseline: the drop-in executable on the FASTA file of the workload, process
    start -> exit with the four files written (SURVEY.md section 8d's definition).  Run BEFORE this process allocates its own
    job: a process that starts while another one's HBM is still being released (277 GB of it once) waits for the driver
    to hand the memory over, which says nothing about the program."""
    from cora_amd import hip
    exe = os.path.join(hip.BIN_DIR, "homTable_setup")
    if not os.path.exists(exe):
        return None
    # The driver runs this on a freshly leased box, where the FIRST process that runs this executable is slower than every later one
    # (releases of device memory early in it took 16 ms each: 163 ms for the ten temporaries of the reference scan against 11 ms;
    # profiles/e2e_variants.py) -- a property of the fresh lease, not of the program.  So the executable is run
    # once on a 250 kbp golden input first, outside the timed run.  (The scan also no longer releases anything before the tables are written.)
    tiny = os.path.join(ROOT, "tests", "golden", "inputs", "syn.fa")
    if os.path.exists(tiny):
        subprocess.run([exe, tiny, str(K), phys, os.path.join(tmp, "tiny_E"), os.path.join(tmp, "tiny_I"), par, "BOTH", str(d), "FULL"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    e2, i2 = os.path.join(tmp, "e2e_Exact"), os.path.join(tmp, "e2e_Inexact")
    t0 = time.time()
    r = subprocess.run([exe, fa, str(K), phys, e2, i2, par, "BOTH", str(d), "FULL"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    wall = time.time() - t0
    e2e = {"what": "cora_amd/bin/homTable_setup on the FASTA file of this workload (page cache warm; one run of the executable on a 250 kbp input beforehand), "
                   "process start -> exit with the four files written",
           "wall_s": round(wall, 3), "exit": r.returncode}
    for line in r.stdout.splitlines():
        if line.startswith("e2e_ms:"):
            toks = line[len("e2e_ms:"):].split("  ")
            e2e["stage_ms"] = {t.rsplit(" ", 1)[0].strip(): float(t.rsplit(" ", 1)[1]) for t in toks if t.strip()}
        if line.startswith("k-mers "):
            e2e["kmers"] = int(line.split()[1])
    if r.returncode == 0:
        e2e["k_mers_per_s"] = e2e.get("kmers", 0) / wall
        e2e["exact_sha256"], e2e["inexact_sha256"] = sha256_file(e2), sha256_file(i2)
    else:
        e2e["output_tail"] = r.stdout[-300:]
    for p in (e2, i2, e2 + ".itemCount", i2 + ".itemCount"):
        if os.path.exists(p):
            os.remove(p)
    return e2e


def extras_after_timing(a, job, image, geom, names, W, K, d, phys, par, tmp, tag, st, e2e):
    """Everything that is reported beside the headline and measured OUTSIDE its timed region (rank 0, N=1)."""
    from cora_amd import hip
    out = {}
    # ---- the last step's tables: bytes checked against the bases, and their hashes
    fa = os.path.join(tmp, "ref.fa")
    if not os.path.exists(fa):
        write_fasta_image(fa, image, geom, names, W)
    ex, en, inx, inn = job.tables()
    ep, ip = os.path.join(tmp, "Exact"), os.path.join(tmp, "Inexact")
    for p, b, n in ((ep, ex, en), (ip, inx, inn)):
        with open(p, "wb") as f:
            f.write(b)
        with open(p + ".itemCount", "w") as f:
            f.write("%d\n" % n)
    table_bytes = (len(ex), len(inx))
    del ex, inx
    job.close()                                     # its HBM is not needed any more
    check = {"exact_sha256": sha256_file(ep), "inexact_sha256": sha256_file(ip), "exact_bytes": table_bytes[0], "inexact_bytes": table_bytes[1]}
    tc = os.path.join(ROOT, "oracle", "table_check")
    if os.path.exists(tc):
        t0 = time.time()
        r = subprocess.run([tc, fa, str(K), str(d), ep, ip], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        check.update({"table_check": "consistent" if r.returncode == 0 else "FAILED (exit %d)" % r.returncode,
                      "table_check_says": r.stdout.strip().splitlines()[-1][:400] if r.stdout.strip() else "", "table_check_s": round(time.time() - t0, 1)})
    else:
        check["table_check"] = "not built"
    out["tables_of_last_step"] = check
    if e2e is not None:
        if e2e.get("exit") == 0:
            e2e["same_bytes_as_timed_step"] = (e2e.pop("exact_sha256") == check["exact_sha256"] and e2e.pop("inexact_sha256") == check["inexact_sha256"])
        out["e2e"] = e2e
    for p in (fa, fa + ".fai", ep, ip, os.path.join(tmp, "e2e_Exact"), os.path.join(tmp, "e2e_Inexact")):
        if os.path.exists(p):
            os.remove(p)
    out["cpu_baseline"] = cpu_baseline(a.workload, K, d, a.cpu_sample, tmp) if a.cpu_sample > 0 else None
    if a.workload == "grch38_like":
        out["secondary"] = secondary_chr20(a, K, d, phys, par, tmp)
    return out


def secondary_chr20(a, K, d, phys, par, tmp):
    """configs[1], round 1's headline: the 64 Mbp chr20-sized contig, same K, d and split files; plus the rows either side of
    the path on that reference."""
    import torch
    from cora_amd import hip, synth
    contigs = synth.workload_chr20_like(20, 64_000_000)
    image, W, geom = synth.fasta_image(contigs, 60)
    del contigs
    job = hip.Job(K, d, phys, par)
    job.set_reference(image, W, geom)
    for _ in range(3):
        job.run()
    torch.cuda.synchronize()
    dt, st, (sc_ms, sc_n, sc_bytes) = run_steps(job.run, job, 10, torch.cuda.synchronize)
    out = {"workload": "chr20_like 64000000 bp, one contig, K=%d d=%d, same split files" % (K, d), "steps": 10, "ms_per_step": dt / 10 * 1e3,
           "k_mers_per_s": st["kmers"] * 10 / dt, "rs_scatter_GB_per_s": (sc_bytes / 1e9) / (sc_ms / 1e3) if sc_ms else None,
           "hom_pairs": int(st["hom_pairs"]), "inexact_nodes": int(st["inexact_nodes"]), "exact_nodes": int(st["exact_nodes"]),
           "stage_ms": {k: round(st[k], 3) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")}}
    ex, en, inx, inn = job.tables()
    out["exact_sha256"], out["inexact_sha256"] = hashlib.sha256(ex).hexdigest(), hashlib.sha256(inx).hexdigest()
    del ex, inx
    # the consumer-side row: the tables just built -> loaded image in HBM
    from cora_amd import tables
    best, wall = None, None
    for _ in range(3):
        tw = time.perf_counter()
        lt = tables.from_job(job)
        w = (time.perf_counter() - tw) * 1e3
        info = lt.info()
        lt.close()
        if best is None or info["ms_total"] < best["ms_total"]:
            best, wall = info, w
    out["table_loader"] = {"what": "both tables of this step -> CSR lists + reverse lookups + dense position hubs in HBM (mappingInference.cpp:554-723, 1084-1153)",
                           "ms_device": round(best["ms_total"], 3), "ms_wall_with_upload_and_allocation": round(wall, 3),
                           "stage_ms": {k: round(best[k], 3) for k in ("ms_parse", "ms_lookups", "ms_hubs")},
                           "image_bytes": int(best["device_bytes"]), "hub_positions": int(best["hub_positions"])}
    job.close()
    # the reference-side file row: contig index of the same image, `cora faiGenerate` beside it
    best = None
    img_bytes = bytes(image)
    for _ in range(3):
        fa = hip.Fasta(img_bytes)
        info = fa.info()
        fai = fa.fai_text()
        fa.close()
        if best is None or info["ms_scan"] < best["ms_scan"]:
            best = info
    fa_path = os.path.join(tmp, "bench.fa")
    with open(fa_path, "wb") as f:
        f.write(img_bytes)
    fai_oracle = os.path.join(ROOT, "oracle", "fai_oracle")
    cpu = None
    if os.path.exists(fai_oracle):
        tw = time.perf_counter()
        r = subprocess.run([fai_oracle, fa_path], stdout=subprocess.PIPE)
        cpu = {"ms": round((time.perf_counter() - tw) * 1e3, 1), "kind": "port", "cores": 1, "same_bytes": r.stdout == fai}
    out["fasta_scan"] = {"what": "contig index (.fai bytes of cora faiGenerate, cora.cpp:1006-1029) of the 64 Mbp image, image resident in HBM",
                         "ms_device": round(best["ms_scan"], 3), "image_bytes": int(best["image_bytes"]), "lines": int(best["lines"]),
                         "GB_per_s_algorithmic": round(2 * best["image_bytes"] / 1e6 / max(best["ms_scan"], 1e-6), 1),
                         "conforming": int(best["conforming"]), "cpu": cpu}
    # the read-side row: reads drawn from that reference collapsed against it (collapse.cpp, whole-input mode)
    if a.collapse_reads > 0:
        import numpy as np
        from cora_amd import collapse as ccol
        img = np.frombuffer(img_bytes, dtype=np.uint8)
        src = img[np.isin(img, np.frombuffer(b"ACGT", dtype=np.uint8))]      # header and N-run dropped (the header's letters, if any, are a few bases of noise)
        fq = synth.fastq_image(77, src, a.collapse_reads, 100)
        del src, img
        cj = ccol.Collapse(5, 100, "SINGLE", "FULL", "GENERIC")
        cj.set_reference(img_bytes)
        cj.add_reads(fq)
        best = None
        for _ in range(3):
            cj.run()
            cs = cj.stats()
            if best is None or cs["ms_device"] < best["ms_device"]:
                best = cs
        cj.close()
        recs = best["ref_windows"] + best["read_mers"]
        out["collapse"] = {"what": "read-mer collapsing of %d synthetic 100-base reads against the 64 Mbp reference (collapse.cpp whole-input mode): three output "
                                   "files as bytes; FASTQ and FASTA images resident in HBM when timed" % a.collapse_reads,
                           "ms_device": round(best["ms_device"], 3), "records": int(recs), "G_records_per_s": round(recs / best["ms_device"] / 1e6, 3),
                           "M_reads_per_s": round(best["reads"] / best["ms_device"] / 1e3, 1),
                           "stage_ms": {k: round(best[k], 3) for k in ("ms_parse", "ms_keys", "ms_group", "ms_order", "ms_text")},
                           "distinct": int(best["distinct"]), "printed": int(best["printed"]), "output_bytes": int(best["reads_bytes"] + best["perfect_bytes"] + best["lrn_bytes"]),
                           "sort_scatter_TB_per_s": round(best["sort_scatter_bytes"] / max(best["ms_sort_scatter"], 1e-9) / 1e9, 2)}
        # the row before it: the same reads partitioned into 8 fragment files by their 4-character signals (fastqSplit.cpp)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from cora_amd import fastqsplit
        import make_fqsplit_golden as mfg
        fsj = fastqsplit.FastqSplit(mfg.frag_text(8, 4).encode(), 100, "SINGLE")
        fsj.add_reads(fq)
        best = None
        for _ in range(3):
            fsj.run()
            fst = fsj.stats()
            if best is None or fst["ms_device"] < best["ms_device"]:
                best = fst
        fsj.close()
        out["fastq_split"] = {"what": "the same %d reads partitioned into 8 fragment files by 4-character signals (fastqSplit.cpp, mode SINGLE): files as bytes "
                                      "in host memory; FASTQ image resident in HBM when timed" % a.collapse_reads,
                              "ms_device": round(best["ms_device"], 3), "M_reads_per_s": round(best["reads"] / best["ms_device"] / 1e3, 1),
                              "stage_ms": {k: round(best[k], 3) for k in ("ms_parse", "ms_classify", "ms_order", "ms_write")},
                              "image_bytes": int(best["image_bytes"]), "output_bytes": int(best["output_bytes"])}
    return out


if __name__ == "__main__":
    main()
