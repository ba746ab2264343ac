#!/usr/bin/env python3
"""collapse beyond the point where the reference's hash map grows, AT THE REFERENCE'S OWN TABLE SIZE (rehash(200000000): first
growth with element 182 356 234, second with 370 310 930): a random reference of `ref_bases` bases (every window distinct), R-base
read-mers, reads drawn from it and mutated.  GPU run (stats, stage times) and, unless cpu=0, oracle/collapse_oracle on the same
files: the three outputs must be the same bytes.  The oracle keeps one node per table element: 4e8 elements take ~35 GB and
several minutes.     python profiles/collapse_grown.py [ref_bases=400000000] [reads=2000000] [R=36] [cpu=1]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import collapse as ccol, hip, synth  # noqa: E402

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 36
cpu = int(sys.argv[4]) if len(sys.argv) > 4 else 1
hip.init(0)
rs = np.random.RandomState(5)
t0 = time.perf_counter()
ref = synth.random_seq(rs, bases)
tmp = tempfile.mkdtemp(prefix="cora_grown_")
fa = os.path.join(tmp, "ref.fa")
synth.write_fasta(fa, [("ref", ref)], 60)
fq = synth.fastq_image(12, ref, reads, R)
fqp = os.path.join(tmp, "r.fq")
open(fqp, "wb").write(fq)
print("reference %d bases, %d reads x %d: generated in %.1f s" % (bases, reads, R, time.perf_counter() - t0), flush=True)
c = ccol.Collapse(4, R, "SINGLE", "FULL", "GENERIC")
c.set_reference(open(fa, "rb").read())
c.add_reads(fq)
for i in range(2):
    t0 = time.perf_counter()
    c.run()
    wall = time.perf_counter() - t0
    st = c.stats()
    print("GPU run %d: %.3f s wall, device %.1f ms (order %.1f ms); %d windows + %d read-mers -> %d distinct, %d printed, table grew %d times" % (
        i, wall, st["ms_device"], st["ms_order"], st["ref_windows"], st["read_mers"], st["distinct"], st["printed"], st["table_growth_steps"]), flush=True)
outs = c.outputs()
c.close()
print("GPU   sha256:", [hashlib.sha256(o).hexdigest()[:16] for o in outs], [len(o) for o in outs], flush=True)
if cpu:
    lst = os.path.join(tmp, "l.txt")
    open(lst, "w").write("1\n%s\t%d\n" % (fqp, reads))
    o = [os.path.join(tmp, x) for x in ("r", "p", "l")]
    t0 = time.perf_counter()
    r = subprocess.Popen([os.path.join(ROOT, "oracle", "collapse_oracle"), lst, "4", o[0], fa, "60", str(R), o[1], "0", o[2], "SINGLE", "FULL", "GENERIC",
                          "NONE", "0", os.path.join(tmp, "k"), "0"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    while r.poll() is None:                     # (a silent run of several minutes would be taken for a hang)
        time.sleep(30)
        print("   oracle running, %.0f s" % (time.perf_counter() - t0), flush=True)
    print("oracle: status %d in %.0f s: %s" % (r.returncode, time.perf_counter() - t0, r.stderr.read().decode()[-200:].strip()), flush=True)
    if r.returncode == 0:
        want = [open(x, "rb").read() for x in o]
        print("oracle sha256:", [hashlib.sha256(x).hexdigest()[:16] for x in want], [len(x) for x in want])
        print("SAME BYTES" if want == outs else "DIFFERENT")
