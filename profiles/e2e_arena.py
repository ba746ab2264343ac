#!/usr/bin/env python3
"""The drop-in executable's wall time (process start -> exit) on the bench workload for several arena sizes:
    python profiles/e2e_arena.py"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import synth  # noqa: E402
import bench  # noqa: E402

S = os.path.join(ROOT, "tests", "golden", "splits")
tmp = tempfile.mkdtemp(prefix="cora_e2e_")
contigs = synth.workload_grch38_like(38, 1.0)
names = [n for n, _ in contigs]
image, W, geom = synth.fasta_image(contigs, 60)
del contigs
fa = os.path.join(tmp, "ref.fa")
bench.write_fasta_image(fa, image, geom, names, W)
del image
for env in [{"CORA_TRACE": "1"}, {"CORA_TRACE": "1", "CORA_ARENA_GB": "161"}, {"CORA_TRACE": "1", "CORA_ARENA_GB": "190"}, {"CORA_TRACE": "1", "CORA_ARENA_GB": "210"}, {"CORA_TRACE": "1", "CORA_ARENA_GB": "236"}, {"CORA_TRACE": "1"}]:
    time.sleep(20)          # (a process that starts while the previous one's memory is still being given back runs several times slower)
    e, x = os.path.join(tmp, "E"), os.path.join(tmp, "I")
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cora_amd", "bin", "homTable_setup"), fa, "50", os.path.join(S, "phys_10.txt"), e, x,
                        os.path.join(S, "par_8.txt"), "BOTH", "2", "FULL"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, **env))
    wall = time.time() - t0
    line = [l for l in r.stdout.splitlines() if l.startswith("e2e_ms:")]
    print("%-28s exit %d, wall %.3f s;  %s" % (env, r.returncode, wall, line[0] if line else ""), flush=True)
    print("\n".join(l for l in r.stdout.splitlines() if "destroyed" in l or "device pool" in l), flush=True)
