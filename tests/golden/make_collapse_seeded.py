#!/usr/bin/env python3
"""Seeded collapse cases beyond the hand-made read sets: read lengths on both sides of every word boundary of the packed
path (read-mers of 12 .. 298 bases), many N, heavy duplication (long read names), paired / split / mapper modes.  Inputs
are regenerated from seeds (cora_amd.synth.fastq_image over a golden reference with its N and lower case removed); the
sha256 of the three files the REAL reference (oracle/_ref/collapse) writes are kept in collapse_seeded_manifest.json.
Run in the build container only:  python tests/golden/make_collapse_seeded.py"""
import hashlib, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from cora_amd import synth  # noqa: E402

MANIFEST = os.path.join(HERE, "collapse_seeded_manifest.json")

CASES = {   # name: (reference, L, reads, seed, paired, split, mapper, D, dup_frac, n_frac)
    "L150_full": ("syn", 150, 4000, 101, False, "FULL", "GENERIC", 3, 0.25, 0.02),
    "L250_full_p_bwa": ("syn", 250, 1500, 102, True, "FULL", "BWA", 3, 0.25, 0.02),
    "L298_half_bt2": ("syn", 298, 2000, 103, False, "HALF", "BOWTIE_2", 3, 0.25, 0.05),
    "L298_full": ("fam", 298, 2000, 104, False, "FULL", "GENERIC", 3, 0.3, 0.01),
    "L36_three": ("syn", 36, 6000, 105, False, "THREEWAY", "GENERIC", 4, 0.25, 0.1),
    "L64_full": ("multi", 64, 5000, 106, False, "FULL", "BWA_MEM", 3, 0.25, 0.03),
    "L128_full_p": ("syn", 128, 2500, 107, True, "FULL", "GENERIC", 3, 0.25, 0.03),
    "L129_full": ("syn", 129, 4000, 108, False, "FULL", "GENERIC", 3, 0.25, 0.03),
    "L96_three": ("syn", 96, 4000, 109, False, "THREEWAY", "BWA", 4, 0.25, 0.2),
    "L100_manyN": ("syn", 100, 5000, 110, False, "FULL", "GENERIC", 3, 0.2, 0.6),
    "L100_heavy_dups": ("syn", 100, 30000, 111, False, "FULL", "GENERIC", 5, 0.7, 0.01),
    "L100_heavy_dups_bwa_p": ("syn", 100, 20000, 112, True, "HALF", "BWA", 5, 0.7, 0.01),
    "L33_full_noref": (None, 33, 8000, 113, False, "FULL", "GENERIC", 3, 0.3, 0.05),
    "L100_heavy_dups_noref": (None, 100, 30000, 115, False, "FULL", "GENERIC", 5, 0.7, 0.01),
    "L100_heavy_dups_noref_bwa": (None, 100, 30000, 116, False, "FULL", "BWA", 5, 0.7, 0.01),
    "L257_half_noref_p": (None, 257, 1500, 114, True, "HALF", "BOWTIE_2", 3, 0.3, 0.05),
}


def clean_reference(name):
    """The golden reference's bases, upper case, anything but A C G T dropped: what the reads are drawn from."""
    seq = []
    for line in open(os.path.join(HERE, "inputs", (name or "syn") + ".fa"), "rb"):
        if not line.startswith(b">"):
            seq.append(line.strip().upper())
    a = np.frombuffer(b"".join(seq), dtype=np.uint8)
    return a[np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8))].copy()


def materialize(name, tmp):
    """Writes the case's FASTQ files and file list under tmp; -> the 16 arguments of collapse (without the program)."""
    ref, L, n, seed, paired, split, mapper, D, dup, nfr = CASES[name]
    src = clean_reference(ref)
    files = []
    for m in range(2 if paired else 1):
        p = os.path.join(tmp, "%s_%d.fq" % (name, m + 1))
        with open(p, "wb") as f:
            f.write(synth.fastq_image(seed * 10 + m, src, n, L, dup_frac=dup, n_frac=nfr))
        files.append(p)
    lst = os.path.join(tmp, name + ".list")
    with open(lst, "w") as f:
        f.write("1\n%s\t%d\n" % ("\t".join(files), n))
    outs = [os.path.join(tmp, name + "." + x) for x in ("reads", "perfect", "lrn")]
    for o in outs:
        if os.path.exists(o):
            os.remove(o)
    refarg = os.path.join(HERE, "inputs", ref + ".fa") if ref else "NOREF"
    args = [lst, str(D), outs[0], refarg, "60", str(L), outs[1], "0", outs[2], "PAIRED" if paired else "SINGLE", split, mapper, "NONE", "0",
            os.path.join(tmp, name + ".kill"), "0"]
    return args, outs


def run(exe, name, tmp):
    args, outs = materialize(name, tmp)
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return [open(o, "rb").read() if os.path.exists(o) else None for o in outs], r


def main():
    tmp = tempfile.mkdtemp(prefix="collapse_seeded_")
    manifest = {}
    for name in CASES:
        outs, r = run(os.path.join(ROOT, "oracle", "_ref", "collapse"), name, tmp)
        assert all(o is not None for o in outs), (name, r.returncode, r.stdout[-300:], r.stderr[-300:])
        manifest[name] = {"sha256": [hashlib.sha256(o).hexdigest() for o in outs], "bytes": [len(o) for o in outs]}
        print(name, manifest[name]["bytes"], r.returncode, flush=True)
    json.dump(manifest, open(MANIFEST, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
