#!/usr/bin/env python3
"""Table loaders at size: build both tables of an N-base chr20_like reference on the GPU, load
them into HBM with the HIP loader (5 times), check the loaded image against oracle/loader_oracle array by array, and time
the CPU side: the restatement and -- where oracle/_ref/ref_loader was built -- the real reference's loaders.
    python profiles/loader_bench.py 64000000 [--no-ref]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cora_amd import hip, synth, tables  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
tmp = tempfile.mkdtemp(prefix="cora_loader_", dir=os.environ.get("CORA_TMP", "/tmp"))
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 10)
synth.write_round_robin_splits(par, 4, 8)
contigs = synth.workload_chr20_like(20, bases)
lens = [len(s) for _, s in contigs]
fa = os.path.join(tmp, "ref.fa")
synth.write_fasta(fa, contigs, 60)
del contigs
hip.init(0)
job = hip.Job(50, 2, phys, par)
job.set_reference_file(fa)
job.run()
ex, en, inx, inn = job.tables()
print("tables: <Exact> %d records %.1f MB, <Inexact> %d nodes %.1f MB, build %.1f ms" % (en, len(ex) / 1e6, inn, len(inx) / 1e6, job.stats()["ms_total"]), flush=True)
best = None
for rep in range(5):
    t0 = time.time()
    t = tables.from_job(job)
    wall = (time.time() - t0) * 1e3
    info = t.info()
    print("load %d: parse %.3f  reverse lookups %.3f  hubs %.3f  total %.3f ms on the device; %.1f ms wall with upload and allocation; %.0f MB image" % (
        rep, info["ms_parse"], info["ms_lookups"], info["ms_hubs"], info["ms_total"], wall, info["device_bytes"] / 1e6), flush=True)
    if best is None or info["ms_total"] < best["ms_total"]:
        best = info
    if rep < 4:
        t.close()
ep, ip = os.path.join(tmp, "Exact"), os.path.join(tmp, "Inexact")
open(ep, "wb").write(ex)
open(ip, "wb").write(inx)
open(ep + ".itemCount", "w").write("%d\n" % en)
open(ip + ".itemCount", "w").write("%d\n" % inn)
import helpers as H  # noqa: E402
out = os.path.join(tmp, "oracle.dump")
t0 = time.time()
r = H.run_loader_oracle(ep, ip, 2, lens, out)
t_port = time.time() - t0
assert r.returncode == 0, r.stdout
H.assert_same_arrays(t.arrays(), H.read_dump(out), "loader at %d bases" % bases)
print("loaded image equals oracle/loader_oracle's, array by array", flush=True)
print("CPU restatement (1 core, dense hubs, incl. writing its %.0f MB dump): %.2f s" % (os.path.getsize(out) / 1e6, t_port), flush=True)
os.remove(out)
ref = os.path.join(ROOT, "oracle", "_ref", "ref_loader")
if os.path.exists(ref) and "--no-ref" not in sys.argv:
    t0 = time.time()
    r = subprocess.run([ref, ep, ip, "50", "2", out] + [str(x) for x in lens], stdout=subprocess.DEVNULL)
    print("real reference loaders (1 core, hash-map hubs, incl. dump): %.2f s (exit %d)" % (time.time() - t0, r.returncode), flush=True)
print("GPU/CPU(port): %.0fx at %.3f ms" % (t_port * 1e3 / best["ms_total"], best["ms_total"]))
t.close()
job.close()
for f in (fa, fa + ".fai", ep, ip, out):
    if os.path.exists(f):
        os.remove(f)
