#!/usr/bin/env python3
"""collapse figures at a given size: reads of a seeded synthetic reference collapsed on the GPU, stage
times from HIP events, and the CPU restatement (oracle/collapse_oracle) beside it on a bounded sample whose three output
files must be the same bytes.
    python profiles/collapse_bench.py [reads] [ref_bases] [read_len] [cpu_sample_reads]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import collapse as ccol, hip, synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
sample = int(sys.argv[4]) if len(sys.argv) > 4 else reads // 4
hip.init(0)
rs = np.random.RandomState(11)
ref = synth.random_seq(rs, bases)
fasta = bytes(synth.fasta_image([("ref", ref)], 60)[0])
D = 5


def run(fq, with_ref=True, split="FULL", rounds=3):
    """Best of `rounds` runs of ONE collapse object (the first run also pays for the device allocations)."""
    c = ccol.Collapse(D, L, "SINGLE", split, "GENERIC")
    if with_ref:
        c.set_reference(fasta)
    c.add_reads(fq)
    best = None
    for i in range(rounds):
        t0 = time.perf_counter()
        c.run()
        wall = (time.perf_counter() - t0) * 1e3
        st = c.stats()
        st["wall_run"] = wall
        if i == 0:
            first = st["ms_device"]
        if best is None or st["ms_device"] < best["ms_device"]:
            best = st
    best["first_run_ms_device"] = first
    outs = c.outputs()
    c.close()
    return best, outs


t0 = time.perf_counter()
fq = synth.fastq_image(12, ref, reads, L)
print("generated %d reads x %d (%.0f MB FASTQ) over %d reference bases in %.1f s" % (reads, L, len(fq) / 1e6, bases, time.perf_counter() - t0), flush=True)
one = os.environ.get("CORA_CB_ONE")          # profiling: a single FULL run, nothing else
for split in (("FULL",) if one else ("FULL", "HALF")):
    st, outs = run(fq, True, split, rounds=1 if one else 3)
    recs = st["ref_windows"] + st["read_mers"]
    print("%s: %d reference windows + %d read-mers -> %d distinct, %d printed; outputs %d / %d / %d B" % (
        split, st["ref_windows"], st["read_mers"], st["distinct"], st["printed"], len(outs[0]), len(outs[1]), len(outs[2])), flush=True)
    print("   device %.2f ms (first run %.2f; parse %.2f, keys %.2f, group %.2f, order %.2f, text %.2f), run() wall %.1f ms: %.2f G records/s, %.1f M reads/s; sort scatter %.2f ms in %d launches = %.2f TB/s" % (
        st["ms_device"], st["first_run_ms_device"], st["ms_parse"], st["ms_keys"], st["ms_group"], st["ms_order"], st["ms_text"], st["wall_run"], recs / st["ms_device"] / 1e6,
        st["reads"] / st["ms_device"] / 1e3, st["ms_sort_scatter"], st["n_sort_scatter"],
        st["sort_scatter_bytes"] / max(st["ms_sort_scatter"], 1e-9) / 1e9), flush=True)
if one:
    sys.exit(0)
st, outs = run(fq, False, "FULL")
print("NOREF FULL: %d read-mers -> %d distinct; device %.2f ms = %.1f M reads/s" % (st["read_mers"], st["distinct"], st["ms_device"], st["reads"] / st["ms_device"] / 1e3), flush=True)

# CPU side on a bounded sample of the same shape (a quarter of the reference and of the reads by default; the restatement
# keeps its keys in a 4 GiB pool): the restatement of the reference's program, same bytes required
sample = min(sample, reads)
sbases = max(1000, int(bases * sample / reads))
ref = ref[:sbases]
fasta = bytes(synth.fasta_image([("ref", ref)], 60)[0])
fq = synth.fastq_image(13, ref, sample, L)
tmp = tempfile.mkdtemp(prefix="cora_cb_")
fqp, fap, lst = os.path.join(tmp, "s.fq"), os.path.join(tmp, "ref.fa"), os.path.join(tmp, "list")
open(fqp, "wb").write(fq)
open(fap, "wb").write(fasta)
open(fap + ".fai", "wb").write(subprocess.run([os.path.join(ROOT, "oracle", "fai_oracle"), fap], stdout=subprocess.PIPE, check=True).stdout)
open(lst, "w").write("1\n%s\t%d\n" % (fqp, sample))
o = [os.path.join(tmp, x) for x in ("reads", "perfect", "lrn")]
cmd = [os.path.join(ROOT, "oracle", "collapse_oracle"), lst, str(D), o[0], fap, "60", str(L), o[1], "0", o[2], "SINGLE", "FULL", "GENERIC", "NONE", "0",
       os.path.join(tmp, "kill"), "0"]
t0 = time.perf_counter()
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
cpu_s = time.perf_counter() - t0
st, outs = run(fq, True, "FULL", rounds=2)
same = r.returncode == 0 and all(open(p, "rb").read() == b for p, b in zip(o, outs))
print("CPU restatement, 1 core, %d reads + %d reference windows: %.1f s = %.2f M records/s; GPU on the same sample %.2f ms = %.2f G records/s; same bytes: %s" % (
    sample, st["ref_windows"], cpu_s, (st["ref_windows"] + sample) / cpu_s / 1e6, st["ms_device"], (st["ref_windows"] + sample) / st["ms_device"] / 1e6, same), flush=True)
for p in o + [fqp, fap, fap + ".fai", lst]:
    if os.path.exists(p):
        os.remove(p)
sys.exit(0 if same else 1)
