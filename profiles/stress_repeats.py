#!/usr/bin/env python3
"""Skewed inputs (BASELINE.json configs[4] in spirit): repeat families up to thousands of copies and Mbp-long tandem
low-complexity stretches.  Builds on the GPU, prints the stage times and counts, and lets oracle/table_check verify every
record against the bases.    python profiles/stress_repeats.py <bases> <max_copies>"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
max_copies = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tmp = tempfile.mkdtemp(prefix="cora_stress_", dir=os.environ.get("CORA_TMP", "/tmp"))
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10)
synth.write_round_robin_splits(par, 4, 8)
contigs = synth.workload_repeat_rich(5, bases, max_copies)
fa = os.path.join(tmp, "ref.fa")
synth.write_fasta(fa, contigs, 60)
del contigs
hip.init(0)
job = hip.Job(50, 2, phys, par)
job.set_reference_file(fa)
for rep in range(2):
    t0 = time.time()
    job.run()
    st = job.stats()
    print("run %d: %.1f ms wall; stages %s" % (rep, (time.time() - t0) * 1e3, {k: round(st[k], 2) for k in ("ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total")}), flush=True)
print({k: st[k] for k in ("kmers", "representatives", "exact_classes", "exact_nodes", "candidate_pairs", "hom_pairs", "inexact_items", "inexact_nodes", "hash_mixed_runs", "full_key_fallback", "dirty_nodes", "big_nodes")}, flush=True)
ex, en, inx, inn = job.tables()
print("<Exact> %.1f MB, <Inexact> %.1f MB" % (len(ex) / 1e6, len(inx) / 1e6), flush=True)
ep, ip = os.path.join(tmp, "Exact"), os.path.join(tmp, "Inexact")
open(ep, "wb").write(ex)
open(ip, "wb").write(inx)
open(ep + ".itemCount", "w").write("%d\n" % en)
open(ip + ".itemCount", "w").write("%d\n" % inn)
del ex, inx
job.close()
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "oracle", "table_check"), fa, "50", "2", ep, ip], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout.strip()[-1500:])
print("table_check exit %d after %.1f s" % (r.returncode, time.time() - t0), flush=True)
for f in (fa, fa + ".fai", ep, ip):
    os.remove(f)
sys.exit(r.returncode)
