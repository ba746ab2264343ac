#!/usr/bin/env python3
"""Generates tests/golden/: inputs, split files and the tables the REAL
reference binary (oracle/_ref/homTable_setup[_L], built from
/root/reference/homTable_setup.cpp by oracle/build_ref.sh) writes for them.

Run in the build container only (needs /root/reference for the LPT tables and
the _ref binaries).  The committed outputs are data: inputs and expected
bytes.  No reference source is stored.

    python tests/golden/make_golden.py [case ...]
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from cora_amd import synth  # noqa: E402

REF = os.environ.get("CORA_REFERENCE", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref", "homTable_setup")
BIN_L = BIN + "_L"

S = synth.seq_from_str


# --------------------------------------------------------------------------
# split files exactly as cora.cpp:295-314 writes them from LPT_2/LPT_4
# --------------------------------------------------------------------------
def lpt_split(which: int, n: int) -> str:
    with open(os.path.join(REF, "LPT_%d.dat" % which)) as f:
        line = f.read().split("\n")[n - 1]
    return "%d\n%s\n" % (n, line.replace("X", "\n"))


def split_file(kind: str, n: int) -> str:
    d = synth.ensure_dir(os.path.join(HERE, "splits"))
    p = os.path.join(d, "%s_%d.txt" % (kind, n))
    if not os.path.exists(p):
        with open(p, "w") as f:
            f.write(lpt_split(2 if kind == "phys" else 4, n))
    return p


# --------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------
def inp_syn():
    rs = np.random.RandomState(1234)
    a = synth.random_seq(rs, 200_000)
    b = synth.random_seq(rs, 50_000)
    unit = synth.random_seq(rs, 300)
    for c in range(40):
        inst = synth.mutate(rs, unit, 0.015)
        if c % 3 == 2:
            inst = synth.revcomp(inst)
        tgt, at = (a, 2000 + c * 4500) if c < 30 else (b, 1000 + (c - 30) * 4000)
        tgt[at:at + 300] = inst
    a[150_000:150_200] = a[10_000:10_200]            # one exact 200 bp duplicate
    a[120_000:120_050] = ord("N")                     # a 50-base N run
    a[60_000:60_100] |= 0x20                          # 100 lower-case bases
    b[49_000:49_003] = S("RYK")                       # IUPAC codes become N
    return [("ctgA", a), ("ctgB", b)]


def pal_seed1_kmer(rs, K=50, n=3):
    """50-mer whose seed-1 positions are reverse-complement symmetric (Q1)."""
    x = synth.random_seq(rs, K)
    for p in range(0, 8 * n, n):
        x[K - 1 - p] = synth._COMP[x[p]]
    return x


def comp_at(x, *pos):
    y = x.copy()
    for p in pos:
        y[p] = synth._COMP[y[p]]
    return y


def inp_pal():
    rs = np.random.RandomState(77)
    g = synth.random_seq(rs, 9000)
    X = pal_seed1_kmer(rs)
    g[1000:1050] = X
    g[2000:2050] = comp_at(X, 1, 25)      # Z
    g[3000:3050] = comp_at(X, 2)          # W
    # a second palindromic-seed family whose head is its own near reverse complement:
    # fully RC-symmetric except one mirrored pair in seed 2 -> self link with 2 mismatches
    Y = pal_seed1_kmer(rs)
    for p in range(25):
        Y[49 - p] = synth._COMP[Y[p]]
    Y2 = Y.copy()
    Y2[48] = ord("A") if Y2[48] != ord("A") else ord("C")   # break the mirrored pair 1/48 (seed 2)
    g[4000:4050] = Y2
    g[5000:5050] = comp_at(Y2, 4)
    g[6000:6050] = synth.revcomp(comp_at(Y2, 7, 30))
    # an exact reverse-complement palindrome 50-mer (Q2) sharing the family
    g[7000:7050] = Y
    return [("pal", g)]


def inp_q():
    rs = np.random.RandomState(5)
    g = synth.random_seq(rs, 3000)
    while True:
        U = synth.random_seq(rs, 50)
        if bytes(synth.revcomp(U)) < bytes(U):
            break
    g[0:50] = U
    g[1500:1550] = synth.revcomp(U)
    h = synth.random_seq(rs, 25)
    g[2200:2250] = np.concatenate([h, synth.revcomp(h)])
    return [("q", g)]


def inp_q4(tail):
    """Q4: the contig's last N-free stretch is `tail` long and repeats bases
    21.. of the contig.  With tail == K the reference never finds that window
    (no exact class); with tail == K+1 both windows exist (one class, block 2).
    A second copy with one substitution feeds the inexact table the same way."""
    rs = np.random.RandomState(44)
    head = synth.random_seq(rs, 400)
    near = head[220:220 + tail].copy()
    near[10] = synth._COMP[near[10]]
    head[300:300 + tail] = near
    g = np.concatenate([head, S("N"), head[20:20 + tail]])
    return [("q4", g)]


def inp_lowcx():
    """Low-complexity stress: tandem, hairpin and poly-A stretches with a few
    mutations -> tied targets, forward+reverse homologies, palindromic seeds."""
    rs = np.random.RandomState(99)
    parts = []
    for rep in range(6):
        parts.append(synth.random_seq(rs, 700))
        parts.append(synth.mutate(rs, np.tile(S("AT"), 120), 0.03))
        parts.append(synth.random_seq(rs, 300))
        h = synth.random_seq(rs, 90)
        parts.append(np.concatenate([h, synth.mutate(rs, synth.revcomp(h), 0.02)]))
        parts.append(synth.random_seq(rs, 200))
        parts.append(synth.mutate(rs, np.tile(S("A"), 200), 0.04))
        parts.append(synth.mutate(rs, np.tile(S("ACGT"), 60), 0.03))
        parts.append(synth.mutate(rs, np.tile(S("GAATTC"), 40), 0.03))
    g = np.concatenate(parts)
    return [("lowcx", g)]


def inp_multi():
    """130 short contigs (forces the _L layout), shared repeats, Q4 at contig ends."""
    rs = np.random.RandomState(130)
    unit = synth.random_seq(rs, 180)
    out = []
    for c in range(130):
        n = int(rs.randint(120, 900))
        s = synth.random_seq(rs, n)
        if c % 7 == 0 and n > 400:
            inst = synth.mutate(rs, unit, 0.02)
            if c % 14 == 0:
                inst = synth.revcomp(inst)
            s[100:280] = inst
        if c % 11 == 3:
            s[n - 51] = ord("N")          # last stretch exactly K=50 long (Q4)
        if c % 11 == 5:
            s[n - 52] = ord("n")          # last stretch K+1 long
        out.append(("c%03d" % c, s))
    return out


def inp_fam():
    """Two contigs, one 60-copy family (lists > 16 items: std::sort's unstable
    regime) at 3 % divergence plus a 25-copy family at 0.5 %."""
    rs = np.random.RandomState(4242)
    a = synth.random_seq(rs, 120_000)
    b = synth.random_seq(rs, 40_000)
    u1 = synth.random_seq(rs, 260)
    for c in range(60):
        inst = synth.mutate(rs, u1, 0.03)
        if c % 4 == 1:
            inst = synth.revcomp(inst)
        tgt, at = (a, 500 + c * 2300) if c < 48 else (b, 700 + (c - 48) * 3000)
        tgt[at:at + 260] = inst
    u2 = synth.random_seq(rs, 150)
    for c in range(25):
        inst = synth.mutate(rs, u2, 0.005)
        at = 1500 + c * 4400
        a[at:at + 150] = inst
    return [("famA", a), ("famB", b)]


INPUTS = {
    "syn": inp_syn, "pal": inp_pal, "q": inp_q, "q4a": lambda: inp_q4(50), "q4b": lambda: inp_q4(51),
    "lowcx": inp_lowcx, "multi": inp_multi, "fam": inp_fam,
    "ecoli_like": lambda: synth.workload_ecoli_like(2024),
    "chr20_like": lambda: synth.workload_chr20_like(20, 64_000_000),     # round 1's bench workload (configs[1])
    "grch38_like_s002": lambda: synth.workload_grch38_like(38, 0.002),   # bench.py's workload (configs[2]) scaled to 6.2 Mbp: 25 contigs
}
# inputs too big to commit are rebuilt from the recipe by the tests (sha256 pinned)
RECIPE_ONLY = {"ecoli_like", "chr20_like", "grch38_like_s002"}
# cases whose tables are too big to commit: only their sha256 and sizes are pinned
HASH_ONLY = {"chr20_like_k50_d2", "chr20_like_k50_d3", "grch38_like_s002_k50_d2", "grch38_like_s002_k64_d3"}

#        case name        input   K   d  p   t  L
CASES = [
    ("syn_k50_d2",        "syn",  50, 2, 3,  2, 0),
    ("syn_k50_d2_p10t8",  "syn",  50, 2, 10, 8, 0),
    ("syn_k50_d1",        "syn",  50, 1, 1,  1, 0),
    ("syn_k64_d3",        "syn",  64, 3, 2,  4, 0),
    ("syn_k33_d2",        "syn",  33, 2, 1,  3, 0),
    ("syn_k40_d3",        "syn",  40, 3, 2,  2, 0),
    ("syn_k47_d2_L",      "syn",  47, 2, 1,  2, 1),
    ("pal_k50_d2",        "pal",  50, 2, 1,  1, 0),
    ("pal_k50_d2_p4t5",   "pal",  50, 2, 4,  5, 0),
    ("q_k50_d2",          "q",    50, 2, 1,  1, 0),
    ("q_k50_d2_p10",      "q",    50, 2, 10, 1, 0),
    ("q4a_k50_d2",        "q4a",  50, 2, 1,  1, 0),
    ("q4b_k50_d2",        "q4b",  50, 2, 1,  1, 0),
    ("lowcx_k50_d2",      "lowcx", 50, 2, 2, 3, 0),
    ("lowcx_k36_d1",      "lowcx", 36, 1, 1, 2, 0),
    ("lowcx_k64_d3",      "lowcx", 64, 3, 3, 7, 0),
    ("multi_k50_d2_L",    "multi", 50, 2, 2, 3, 1),
    ("fam_k50_d2",        "fam",  50, 2, 1,  4, 0),
    ("fam_k48_d3",        "fam",  48, 3, 2,  2, 0),
    ("ecoli_like_k50_d2", "ecoli_like", 50, 2, 1, 8, 0),
    ("chr20_like_k50_d2", "chr20_like", 50, 2, 1, 8, 0),
    # d = 3 on a contig longer than 50 Mbp: the window-by-window compaction (HTS:1467-1581), with cora's default -p 10 -t 8
    ("chr20_like_k50_d3", "chr20_like", 50, 3, 10, 8, 0),
    # configs[3] as SURVEY.md section 0 reads it: -H 3 -K 64 and -t 24 (LPT_4.dat line 24)
    ("syn_k64_d3_p10t24", "syn",  64, 3, 10, 24, 0),
    # the bench generator at 1/500 of its size (25 contigs, N runs, capped repeat families, soft-masked copies), cora's defaults
    ("grch38_like_s002_k50_d2", "grch38_like_s002", 50, 2, 10, 8, 0),
    ("grch38_like_s002_k64_d3", "grch38_like_s002", 64, 3, 10, 24, 0),
]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    want = set(sys.argv[1:])
    manifest_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(manifest_path)) if os.path.exists(manifest_path) else {"inputs": {}, "cases": {}}
    indir = synth.ensure_dir(os.path.join(HERE, "inputs"))
    tmp = tempfile.mkdtemp(prefix="golden_")
    made = {}
    for name, inp, K, d, p, t, L in CASES:
        if want and name not in want:
            continue
        if inp not in made:
            contigs = INPUTS[inp]()
            where = tmp if inp in RECIPE_ONLY else indir
            fa = os.path.join(where, inp + ".fa")
            synth.write_fasta(fa, contigs, 60)
            made[inp] = fa
            manifest["inputs"][inp] = {"sha256": sha256(fa), "recipe_only": inp in RECIPE_ONLY,
                                       "bases": int(sum(len(s) for _, s in contigs))}
        fa = made[inp]
        phys, par = split_file("phys", p), split_file("par", t)
        out = os.path.join(tmp, name)
        os.makedirs(out)
        ex, inx = os.path.join(out, "Exact"), os.path.join(out, "Inexact")
        env = dict(os.environ, OMP_DYNAMIC="false")
        env.pop("OMP_THREAD_LIMIT", None)
        cmd = [BIN_L if L else BIN, fa, str(K), phys, ex, inx, par, "BOTH", str(d), "FULL"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout[-3000:])
            raise SystemExit("reference failed on %s (exit %d)" % (name, r.returncode))
        if name not in HASH_ONLY:
            dst = synth.ensure_dir(os.path.join(HERE, "cases", name))
            for fn in ("Exact", "Exact.itemCount", "Inexact", "Inexact.itemCount"):
                shutil.copy(os.path.join(out, fn), os.path.join(dst, fn))
        manifest["cases"][name] = dict(manifest["cases"].get(name, {}), **{
            "input": inp, "K": K, "d": d, "p": p, "t": t, "L": L,
            "exact_nodes": int(open(ex + ".itemCount").read()),
            "inexact_nodes": int(open(inx + ".itemCount").read()),
            "exact_sha256": sha256(ex), "inexact_sha256": sha256(inx),
            "exact_bytes": os.path.getsize(ex), "inexact_bytes": os.path.getsize(inx), "hash_only": name in HASH_ONLY,
        })
        print("%-22s exact %6d nodes %8d B   inexact %7d nodes %9d B" % (
            name, manifest["cases"][name]["exact_nodes"], os.path.getsize(ex),
            manifest["cases"][name]["inexact_nodes"], os.path.getsize(inx)), flush=True)
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
