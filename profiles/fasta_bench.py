#!/usr/bin/env python3
"""FASTA scan / loader figures at a given size: scan time of a conforming image, of the same bases in a
ragged layout, the gather into a dense image, and the C port of the awk program beside them.
    python profiles/fasta_bench.py [bases]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import hip, synth

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
hip.init(0)
rs = np.random.RandomState(1)
seq = rs.randint(0, 4, bases).astype(np.uint8)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[seq]
image = bytes(synth.fasta_image([("big", seq)], 60)[0])
widths = rs.choice([59, 60, 61, 70, 120], size=bases // 50)
cuts = np.cumsum(widths)
cuts = cuts[cuts < bases]
ragged = np.insert(seq, cuts, 10)
ragged = b">big\n" + ragged.tobytes() + b"\n"
del seq
tmp = tempfile.mkdtemp(prefix="cora_fb_")
phys, par = os.path.join(tmp, "p"), os.path.join(tmp, "t")
synth.write_round_robin_splits(phys, 2, 1)
synth.write_round_robin_splits(par, 4, 1)
for label, img in (("conforming", image), ("ragged", ragged)):
    best = None
    for _ in range(3):
        f = hip.Fasta(img)
        info = f.info()
        fai = f.fai_text()
        f.close()
        if best is None or info["ms_scan"] < best["ms_scan"]:
            best = info
    print("%-10s %d bytes, %d lines: scan %.3f ms = %.0f GB/s over 2 reads of the image; conforming=%d" % (
        label, best["image_bytes"], best["lines"], best["ms_scan"], 2 * best["image_bytes"] / 1e6 / best["ms_scan"], best["conforming"]), flush=True)
    job = hip.Job(50, 2, phys, par)
    t0 = time.perf_counter()
    job.set_reference_fasta(img)
    print("%-10s set_reference_fasta wall %.1f ms (upload + scan + %s)" % (label, (time.perf_counter() - t0) * 1e3,
          "device copy" if best["conforming"] else "gather"), flush=True)
    job.close()
    p = os.path.join(tmp, label + ".fa")
    open(p, "wb").write(img)
    t0 = time.perf_counter()
    out = subprocess.run([os.path.join(ROOT, "oracle", "fai_oracle"), p], stdout=subprocess.PIPE).stdout
    print("%-10s C port of the awk program: %.0f ms, same bytes: %s" % (label, (time.perf_counter() - t0) * 1e3, out == fai), flush=True)
    os.remove(p)
