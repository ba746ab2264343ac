#!/usr/bin/env python3
"""The drop-in executable on the bench workload's FASTA file under several settings of its reference upload, each run a fresh process
(8 s apart: a process that starts while the previous one's 190 GiB are still being scrubbed waits for the memory).
    python profiles/e2e_variants.py [scale] [VAR=value,VAR=value ...]..."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import synth  # noqa: E402
import bench  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[2:]] or [{}]
S = os.path.join(ROOT, "tests", "golden", "splits")
tmp = tempfile.mkdtemp(prefix="cora_e2e_")
contigs = synth.workload_grch38_like(38, scale)
names = [n for n, _ in contigs]
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
fa = os.path.join(tmp, "ref.fa")
bench.write_fasta_image(fa, image, geom, names, W)
del image
if os.environ.get("E2E_TINY_FIRST"):          # does a run of the executable on a 250 kbp reference make the first big run fast?
    g = os.path.join(ROOT, "tests", "golden", "inputs")
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cora_amd", "bin", "homTable_setup"), os.path.join(g, "syn.fa"), "50", os.path.join(S, "phys_10.txt"),
                        os.path.join(tmp, "tE"), os.path.join(tmp, "tI"), os.path.join(S, "par_8.txt"), "BOTH", "2", "FULL"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print("tiny run first: exit %d, wall %.3f s" % (r.returncode, time.time() - t0), flush=True)
for i, var in enumerate(variants):
    e, x = os.path.join(tmp, "E"), os.path.join(tmp, "I")
    for p in (e, x):
        if os.path.exists(p):
            os.remove(p)
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cora_amd", "bin", "homTable_setup"), fa, "50", os.path.join(S, "phys_10.txt"), e, x,
                        os.path.join(S, "par_8.txt"), "BOTH", "2", "FULL"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, CORA_TRACE="1", **var))
    print("run %d %s: exit %d, wall %.3f s" % (i, var, r.returncode, time.time() - t0))
    for line in r.stdout.splitlines():
        if any(k in line for k in ("fasta:", "scan done", "scan starts", "scan returned", "e2e_ms", "ms: pack", "ERROR", "arena", "destroyed", "shut down", "[cora] run", "warm:", "items: waited", "inexact host finish")):
            print("   ", line)
    sys.stdout.flush()
    time.sleep(8)
