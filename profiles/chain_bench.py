#!/usr/bin/env python3
"""The read side as a chain (SURVEY 8f.2 + 8f.4): fastqSplit -> collapse in its fragmented mode, once per group, on ONE
collapse object; fragments handed over in HBM versus through host memory, and the whole-input collapse of the same reads.
    python profiles/chain_bench.py [reads] [ref_bases] [groups]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cora_amd import collapse as ccol, fastqsplit, hip, synth
import make_fqsplit_golden as mfg

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L, D = 100, 5
hip.init(0)
ref = synth.random_seq(np.random.RandomState(11), bases)
fasta = bytes(synth.fasta_image([("ref", ref)], 60)[0])
fq = synth.fastq_image(12, ref, reads, L)
sig = mfg.frag_text(groups, 4)
tokens = sig.split("\n")[1:1 + groups]
fs = fastqsplit.FastqSplit(sig.encode(), L, "SINGLE")
fs.add_reads(fq)
c = ccol.Collapse(D, L, "SINGLE", "FULL", "GENERIC")
c.set_reference(fasta)
for in_hbm in (True, False, True, False):
    fs.set_device_only(in_hbm)
    t0 = time.perf_counter()
    fs.run()
    dev_ms = fs.stats()["ms_device"]
    t1 = time.perf_counter()
    lrn, out_bytes, printed = 0, 0, 0
    for g in range(groups):
        c.set_fragment(tokens[g], 4)
        c.set_lrn_offset(lrn)
        if in_hbm:
            c.add_reads_device(*fs.group_device(g))
        else:
            c.add_reads(fs.group(g))
        c.run()
        st = c.stats()
        dev_ms += st["ms_device"]
        lrn += st["lrn_names"]
        printed += st["printed"]
        out_bytes += st["reads_bytes"] + st["perfect_bytes"] + st["lrn_bytes"]
    t2 = time.perf_counter()
    print("chain, fragments %s: fastqSplit %.1f ms wall + %d x collapse %.1f ms wall = %.1f ms; device %.1f ms; %d groups printed, %d B of output" % (
        "in HBM       " if in_hbm else "through host ", (t1 - t0) * 1e3, groups, (t2 - t1) * 1e3, (t2 - t0) * 1e3, dev_ms, printed, out_bytes), flush=True)
w = ccol.Collapse(D, L, "SINGLE", "FULL", "GENERIC")
w.set_reference(fasta)
w.add_reads(fq)
for _ in range(2):
    t0 = time.perf_counter()
    w.run()
    t1 = time.perf_counter()
    st = w.stats()
    print("whole-input collapse of the same reads: %.1f ms wall, device %.1f ms; %d groups printed, %d B of output" % (
        (t1 - t0) * 1e3, st["ms_device"], st["printed"], st["reads_bytes"] + st["perfect_bytes"] + st["lrn_bytes"]), flush=True)
