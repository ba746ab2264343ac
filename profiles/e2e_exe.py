#!/usr/bin/env python3
"""Writes the bench workload as a FASTA file and runs the drop-in executable on it, twice, printing what it prints.
    python profiles/e2e_exe.py [scale]"""
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
S = os.path.join(ROOT, "tests", "golden", "splits")
tmp = tempfile.mkdtemp(prefix="cora_e2e_")
contigs = synth.workload_grch38_like(38, scale)
names = [n for n, _ in contigs]
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
fa = os.path.join(tmp, "ref.fa")
bench.write_fasta_image(fa, image, geom, names, W)
del image
for i in range(2):
    e, x = os.path.join(tmp, "E%d" % i), os.path.join(tmp, "I%d" % i)
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cora_amd", "bin", "homTable_setup"), fa, "50", os.path.join(S, "phys_10.txt"), e, x,
                        os.path.join(S, "par_8.txt"), "BOTH", "2", "FULL"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, CORA_TRACE=os.environ.get("CORA_TRACE", "1")))
    print("run %d: exit %d, wall %.3f s" % (i, r.returncode, time.time() - t0))
    print(r.stdout[-3000:], flush=True)
