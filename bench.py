#!/usr/bin/env python3
"""Benchmark of the homology-table hot path (BASELINE.json: reference k-mers/s into the d=2 table).

    python bench.py --gpus N --steps K --warmup W

A step is one complete table build (2-bit pack -> canonical k-mers -> radix sort -> classes ->
seeds -> pair verification -> block merge -> both tables as bytes) over a synthetic reference whose
FASTA image is already resident in HBM when the timed region starts.  N=1 runs configs[1]: the
chr20-sized (64 Mbp) synthetic contig, K=50, d=2.  For N>1 the ranks build ONE table pair over a reference
of N such contigs (weak scaling: 64 Mbp per GPU): every rank holds the packed reference, scans 1/N of the
positions and owns 1/N of the key space; records move by all-to-all over RCCL (cora_amd/dist.py) and rank 0
ends up with both tables.  --mode replicas instead builds one independent 64 Mbp table per rank (no
collective in the data path).  One JSON line on rank 0."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(K, d, sample_bases, tmp):
    """The real reference (oracle/_ref, kind 'reference') or the C oracle ('port') on a bounded
    sample of the same workload, timed on this box's host cores."""
    from cora_amd import synth
    fa = os.path.join(tmp, "sample.fa")
    contigs = synth.workload_chr20_like(20, sample_bases)
    synth.write_fasta(fa, contigs, 60)
    phys, par = os.path.join(tmp, "phys1.txt"), os.path.join(tmp, "par8.txt")
    synth.write_round_robin_splits(phys, 2, 1)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "homTable_setup")
    oracle_bin = os.path.join(ROOT, "oracle", "hts_oracle")
    if os.path.exists(ref_bin):
        kind, exe, threads = "reference", ref_bin, 8
    else:
        if not os.path.exists(oracle_bin):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "hts_oracle"], stdout=subprocess.DEVNULL)
        kind, exe, threads = "port", oracle_bin, 1
    synth.write_round_robin_splits(par, 4, threads)
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
    nonN = int(sum((s != ord("N")).sum() for _, s in contigs))
    return {"value": nonN / dt, "unit": "k-mers/s", "cores": threads, "kind": kind,
            "sample": "chr20_like generator at %d bp (K=%d d=%d, -p 1 -t %d), %.1f s wall" % (sample_bases, K, d, threads, dt)}


def pmc_traffic(bases, K, d):
    """HBM bytes per rs_scatter launch from the newest committed PMC summary of this very workload
    (profiles/collect.sh + summarize.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for gfx950)."""
    import glob
    want = "chr20_like %d bp K=%d d=%d" % (bases, K, d)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("workload") == want:
            return t["bytes_per_launch"], t.get("source")
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=64_000_000, help="reference length (default: chr20-sized)")
    ap.add_argument("--cpu-sample", type=int, default=16_000_000, help="bases of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--collapse-reads", type=int, default=4_000_000, help="reads of the read-side figure printed beside the headline (0 = skip)")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded", help="what N>1 ranks do (see the module text)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo = host-staged rehearsal)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    local_rank %= max(1, torch.cuda.device_count())      # rehearsals put several ranks on one GPU
    torch.cuda.set_device(local_rank)
    # PyTorch creates its HIP context lazily; done inside the first sync() below it made the first TIMED step 5 ms slower
    # (seen with --warmup 1: 19.8 ms, then 14.0 ms).  Create it here, before anything is measured.
    torch.zeros(1, device="cuda").add_(1)
    torch.cuda.synchronize()
    side = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
        side = dist.new_group(backend="gloo")        # host-side agreement between the ranks (see the warm-up below)

    from cora_amd import hip, synth
    K, d = 50, 2
    hip.init(local_rank)
    tmp = tempfile.mkdtemp(prefix="cora_bench_")
    phys, par = os.path.join(tmp, "phys10.txt"), os.path.join(tmp, "par8.txt")
    synth.write_round_robin_splits(phys, 2, 10)      # cora's defaults: -p 10 -t 8
    synth.write_round_robin_splits(par, 4, 8)
    sharded = world > 1 and a.mode == "sharded"
    if sharded:                                     # one reference of `world` contigs, the same on every rank
        contigs = [("chr20_like_%d" % c, synth.workload_chr20_like(20 + c, a.bases)[0][1]) for c in range(world)]
    else:
        contigs = synth.workload_chr20_like(20 + rank, a.bases)
    image, W, geom = synth.fasta_image(contigs, 60)
    del contigs
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

    # Warm-up.  The sharded flow over RCCL can only be rehearsed with one rank per GPU, i.e. not on a one-GPU box: if a rank
    # fails in it, all ranks agree (over gloo) to measure independent replicas instead, and the line says so.
    fallback = None
    try:
        if sharded and os.environ.get("CORA_BENCH_FAIL_SHARDED"):      # rehearsal of the fall-back itself
            raise RuntimeError("CORA_BENCH_FAIL_SHARDED is set")
        for _ in range(a.warmup):
            run()
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
            print("[bench] rank %d: sharded flow failed (%s); measuring replicas" % (rank, fallback), file=sys.stderr, flush=True)
            sharded = False
            job.close()
            image, W, geom = synth.fasta_image(synth.workload_chr20_like(20 + rank, a.bases), 60)
            job = hip.Job(K, d, phys, par)
            job.set_reference(image, W, geom)
            run = job.run
            for _ in range(a.warmup):
                run()
            sync()
    t0 = time.perf_counter()
    sc_ms, sc_n, sc_bytes = 0.0, 0, 0
    for _ in range(a.steps):
        ts = time.perf_counter()
        run()
        st = job.stats()
        if os.environ.get("CORA_TRACE"):
            print("[bench] step wall %.3f ms  %s" % ((time.perf_counter() - ts) * 1e3, " ".join("%s=%.2f" % (k[3:], st[k]) for k in (
                "ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish"))), file=sys.stderr)
        sc_ms += st["ms_sort_scatter"]; sc_n += st["n_sort_scatter"]; sc_bytes += st["sort_scatter_bytes"]
    sync()
    dt = time.perf_counter() - t0
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

    if rank == 0:
        achieved = (sc_bytes / 1e9) / (sc_ms / 1e3) if sc_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(a.bases, K, d) if world == 1 else (None, None)
        out = {
            "metric": "ref k-mers/sec into homology table (d=2)", "value": kmers_all * a.steps / dt, "unit": "k-mers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("one reference of %d chr20_like synthetic contigs (%d bp each, one per GPU), K=%d, d=%d, 60-column FASTA "
                                    "image resident in every GPU's HBM; output = both tables of the whole reference as bytes in rank 0's "
                                    "host memory" % (world, a.bases, K, d)) if sharded else
                                   ("chr20_like synthetic contig, %d bp per GPU, K=%d, d=%d, 60-column FASTA image resident in HBM; "
                                    "output = both tables as bytes in host memory" % (a.bases, K, d)),
                       "kmers_rank0": int(st["kmers"]), "hom_pairs": int(st["hom_pairs"]), "inexact_nodes": int(st["inexact_nodes"]),
                       "exact_nodes": int(st["exact_nodes"]), "hash_mixed_runs": int(st["hash_mixed_runs"]),
                       "full_key_fallback": int(st["full_key_fallback"]), "dirty_nodes": int(st["dirty_nodes"]), "big_nodes": int(st["big_nodes"]), "parallelism": ("1 GPU" if world == 1 else "sharded: positions and key space split over %d GPUs, all-to-all over %s" % (world, a.backend)
                                       if sharded else "replicas: 1 independent reference per GPU"), "sharded_flow_failure": fallback},
            "stage_ms": {k: round(st[k], 3) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")},
            "roofline": {"bound": "hbm", "kernel": "rs_scatter (LSD radix scatter, all sorts of the path)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "launches_per_step": sc_n // max(1, a.steps), "avg_launch_ms": sc_ms / max(1, sc_n),
                         "algorithmic_bytes_per_launch": sc_bytes / max(1, sc_n), "traffic": traffic, "traffic_source": traffic_src},
        }
        cb = cpu_baseline(K, d, a.cpu_sample, tmp) if (a.cpu_sample > 0 and world == 1) else None
        out["cpu_baseline"] = cb
        if world == 1:       # outside the timed region: the consumer-side row, the tables just built -> loaded image in HBM
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
            out["fasta_scan"] = {"what": "contig index (.fai bytes of cora faiGenerate, cora.cpp:1006-1029) of the bench image, image resident in HBM",
                                 "ms_device": round(best["ms_scan"], 3), "image_bytes": int(best["image_bytes"]), "lines": int(best["lines"]),
                                 "GB_per_s_algorithmic": round(2 * best["image_bytes"] / 1e6 / max(best["ms_scan"], 1e-6), 1),
                                 "conforming": int(best["conforming"]), "cpu": cpu}
            # the read-side row: reads drawn from the bench reference collapsed against it (collapse.cpp, whole-input mode)
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
                out["collapse"] = {"what": "read-mer collapsing of %d synthetic 100-base reads against the bench reference (collapse.cpp whole-input mode): three output "
                                           "files as bytes; FASTQ and FASTA images resident in HBM when timed" % a.collapse_reads,
                                   "ms_device": round(best["ms_device"], 3), "records": int(recs), "G_records_per_s": round(recs / best["ms_device"] / 1e6, 3),
                                   "M_reads_per_s": round(best["reads"] / best["ms_device"] / 1e3, 1),
                                   "stage_ms": {k: round(best[k], 3) for k in ("ms_parse", "ms_keys", "ms_group", "ms_order", "ms_text")},
                                   "distinct": int(best["distinct"]), "printed": int(best["printed"]), "output_bytes": int(best["reads_bytes"] + best["perfect_bytes"] + best["lrn_bytes"]),
                                   "sort_scatter_TB_per_s": round(best["sort_scatter_bytes"] / max(best["ms_sort_scatter"], 1e-9) / 1e9, 2),
                                   "cpu": "profiles/collapse_bench.py times oracle/collapse_oracle on a quarter-size sample (same bytes required)"}
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
                                      "image_bytes": int(best["image_bytes"]), "output_bytes": int(best["output_bytes"]),
                                      "cpu": "profiles/fqsplit_bench.py times oracle/fastqsplit_oracle on the same input (same bytes required)"}
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
