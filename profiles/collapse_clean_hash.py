#!/usr/bin/env python3
"""The hashing loop for reference windows WITHOUT the per-base N test (collapse.hip: fnv_of_clean), against the general loop:
same three files, on repeated runs, for read-mers of several lengths?  (Round 1 dropped such a variant because it "hashed
some windows differently from run to run"; this is the check the round-1 review asked for.)
    python profiles/collapse_clean_hash.py [reads] [ref_bases]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 32_000_000
mode = os.environ.get("CORA_COLLAPSE_CLEAN_HASH", "0")
from cora_amd import collapse as ccol, hip, synth  # noqa: E402

hip.init(0)
rs = np.random.RandomState(11)
ref = synth.random_seq(rs, bases)
ref[1000:1200] = ord("N")
fasta = bytes(synth.fasta_image([("ref", ref)], 60)[0])
clean = ref[ref != ord("N")]
for L, split in ((100, "FULL"), (100, "HALF"), (150, "THREEWAY"), (250, "FULL"), (36, "FULL")):
    fq = synth.fastq_image(12, clean, reads, L)
    c = ccol.Collapse(5, L, "SINGLE", split, "GENERIC")
    c.set_reference(fasta)
    c.add_reads(fq)
    seen = set()
    for rep in range(4):
        c.run()
        st = c.stats()
        seen.add((st["distinct"], st["printed"]) + tuple(hashlib.sha256(o).hexdigest()[:16] for o in c.outputs()))
    c.close()
    print("clean_hash=%s L=%d %s: %d distinct result(s) over 4 runs: %s" % (mode, L, split, len(seen), sorted(seen)), flush=True)
