#!/usr/bin/env python3
"""One-off validation at sizes no oracle reaches: build both tables of an N-base chr20_like reference on the GPU, write
them out, and let oracle/table_check verify every record against the bases (it is pinned on the real reference's own
tables by tests/test_table_check.py).      python profiles/check_large.py 3100000000"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
tmp = tempfile.mkdtemp(prefix="cora_large_", dir=os.environ.get("CORA_TMP", "/tmp"))
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10)
synth.write_round_robin_splits(par, 4, 8)
t0 = time.time()
contigs = synth.workload_chr20_like(20, bases)
fa = os.path.join(tmp, "ref.fa")
synth.write_fasta(fa, contigs, 60)
del contigs
print("reference written in %.1f s" % (time.time() - t0), flush=True)
hip.init(0)
job = hip.Job(50, 2, phys, par)
job.set_reference_file(fa)
t0 = time.time()
job.run()
print("built in %.3f s" % (time.time() - t0), flush=True)
ex, en, inx, inn = job.tables()
st = job.stats()
print({k: st[k] for k in ("kmers", "representatives", "hom_pairs", "inexact_nodes", "exact_nodes", "hash_mixed_runs", "full_key_fallback", "dirty_nodes", "big_nodes")}, flush=True)
ep, ip = os.path.join(tmp, "Exact"), os.path.join(tmp, "Inexact")
open(ep, "wb").write(ex)
open(ip, "wb").write(inx)
open(ep + ".itemCount", "w").write("%d\n" % en)
open(ip + ".itemCount", "w").write("%d\n" % inn)
del ex, inx
job.close()
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "oracle", "table_check"), fa, "50", "2", ep, ip], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout.strip())
print("table_check exit %d after %.1f s" % (r.returncode, time.time() - t0), flush=True)
for f in (fa, fa + ".fai", ep, ip):
    os.remove(f)
sys.exit(r.returncode)
