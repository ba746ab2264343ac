#!/usr/bin/env python3
"""Writes tests/golden/collapse_inputs/*.fq (seeded read sets over the golden references) and, for every mode
combination in CASES, the three files the REAL reference (oracle/_ref/collapse[_L], built from
/root/reference/collapse.cpp by oracle/build_ref.sh) writes: full bytes for the cases marked keep, sha256 for all
(tests/golden/collapse_manifest.json).  Run in the build container only; the outputs are data and are committed."""
import hashlib, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INP = os.path.join(HERE, "collapse_inputs")
OUT = os.path.join(HERE, "collapse_cases")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def load_ref(path):
    seqs, cur = [], []
    for line in open(path):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip().upper())
    seqs.append("".join(cur))
    return seqs


def read_set(ref_name, L, n, seed, paired):
    """n reads (pairs) of length L: exact copies of reference stretches in both strands, mutated ones, ones with N, heavy
    duplicates, reads that are not in the reference at all, palindromes."""
    rs = np.random.RandomState(seed)
    ref = [s for s in load_ref(os.path.join(HERE, "inputs", ref_name + ".fa")) if len(s) >= L + 5 and set(s) <= set("ACGTN")]
    popular = []
    for _ in range(6):
        s = ref[rs.randint(len(ref))]
        p = rs.randint(0, len(s) - L + 1)
        popular.append(s[p:p + L])
    popular.append("".join("ACGT"[i] for i in rs.randint(0, 4, L)))          # a popular read that is not in the reference

    def one():
        k = rs.randint(0, 100)
        if k < 45 or k >= 92:
            s = ref[rs.randint(len(ref))]
            p = rs.randint(0, len(s) - L + 1)
            r = s[p:p + L]
            if "N" in r and k < 45:
                r = r.replace("N", "A")
            if 30 <= k < 40:                                                 # one or two substitutions
                r = list(r)
                for _ in range(rs.randint(1, 3)):
                    r[rs.randint(L)] = "ACGT"[rs.randint(4)]
                r = "".join(r)
            if 40 <= k < 45:
                r = list(r); r[rs.randint(L)] = "N"; r = "".join(r)
            if rs.randint(2):
                r = rc(r)
            return r
        if k < 75:
            r = popular[rs.randint(len(popular))]
            return rc(r) if rs.randint(2) else r
        if k < 85:
            return "".join("ACGT"[i] for i in rs.randint(0, 4, L))
        h = "".join("ACGT"[i] for i in rs.randint(0, 4, L // 2))
        mid = "" if L % 2 == 0 else "N"
        return h + mid + rc(h)                                               # its own reverse complement (even L)
    files = []
    for m in range(2 if paired else 1):
        lines = []
        for i in range(n):
            r = one()
            lines.append("@r%d/%d\n%s\n+\n%s\n" % (i, m + 1, r, "I" * L))
        files.append("".join(lines))
    return files


READ_SETS = {          # name: (reference, L, n, seed, paired)
    "syn100p": ("syn", 100, 700, 1, True),
    "syn100s": ("syn", 100, 900, 2, False),
    "syn101p": ("syn", 101, 400, 3, True),
    "multi60s": ("multi", 60, 1500, 4, False),
    "multi60p": ("multi", 60, 500, 5, True),
    "fam75s": ("fam", 75, 800, 6, False),
    "syn100s_b": ("syn", 100, 300, 7, False),
    "syn100s_big": ("syn", 100, 6000, 8, False),     # popular reads with hundreds of copies: long read names in most fragment groups
}

CASES = {   # name: (read sets (samples), reference or None, D, input mode, split, mapper, L variant, lrn offset, keep bytes)
    "p_full_generic": (["syn100p"], "syn", 3, "PAIRED", "FULL", "GENERIC", False, 0, True),
    "p_full_bwa": (["syn100p"], "syn", 3, "PAIRED", "FULL", "BWA", False, 0, False),
    "p_full_bwamem": (["syn100p"], "syn", 3, "PAIRED", "FULL", "BWA_MEM", False, 5, False),
    "p_full_bowtie2": (["syn100p"], "syn", 3, "PAIRED", "FULL", "BOWTIE_2", False, 0, True),
    "p_half_generic": (["syn100p"], "syn", 3, "PAIRED", "HALF", "GENERIC", False, 0, True),
    "p_three_generic": (["syn100p"], "syn", 3, "PAIRED", "THREEWAY", "GENERIC", False, 0, False),
    "p_half_odd": (["syn101p"], "syn", 4, "PAIRED", "HALF", "BWA", False, 98, False),
    "p_three_odd": (["syn101p"], "syn", 4, "PAIRED", "THREEWAY", "BOWTIE_2", False, 0, False),
    "p_full_noref": (["syn100p"], None, 3, "PAIRED", "FULL", "GENERIC", False, 0, False),
    "s_full_generic": (["syn100s"], "syn", 2, "SINGLE", "FULL", "GENERIC", False, 0, True),
    "s_full_bwa_d2": (["syn100s"], "syn", 2, "SINGLE", "FULL", "BWA", False, 0, True),
    "s_full_bwamem_d2": (["syn100s"], "syn", 2, "SINGLE", "FULL", "BWA_MEM", False, 0, False),
    "s_half_generic": (["syn100s"], "syn", 3, "SINGLE", "HALF", "GENERIC", False, 0, False),
    "s_three_bwa": (["syn100s"], "syn", 3, "SINGLE", "THREEWAY", "BWA", False, 0, False),
    "s_two_samples": (["syn100s", "syn100s_b"], "syn", 3, "SINGLE", "FULL", "GENERIC", False, 9, True),
    "s_noref_half": (["syn100s"], None, 3, "SINGLE", "HALF", "BOWTIE_2", False, 0, False),
    "multi_s_full": (["multi60s"], "multi", 3, "SINGLE", "FULL", "GENERIC", False, 0, True),
    "multi_s_full_L": (["multi60s"], "multi", 3, "SINGLE", "FULL", "GENERIC", True, 0, False),
    "multi_p_half_L": (["multi60p"], "multi", 3, "PAIRED", "HALF", "BWA_MEM", True, 0, False),
    "multi_s_three": (["multi60s"], "multi", 5, "SINGLE", "THREEWAY", "GENERIC", False, 0, False),
    "p_full_generic_d4": (["syn100p"], "syn", 4, "PAIRED", "FULL", "GENERIC", False, 0, True),
    "s_full_bowtie2_d4": (["syn100s"], "syn", 4, "SINGLE", "FULL", "BOWTIE_2", False, 97, False),
    "s_half_generic_d5": (["syn100s"], "syn", 5, "SINGLE", "HALF", "GENERIC", False, 0, False),
    "fam_s_full": (["fam75s"], "fam", 3, "SINGLE", "FULL", "BWA", False, 0, False),
    "fam_s_full_d6": (["fam75s"], "fam", 6, "SINGLE", "FULL", "GENERIC", False, 0, False),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def run_case(exe, case, tmp, extra=()):
    """Runs a collapse-compatible executable on a case; -> (reads, perfect, lrn) bytes."""
    sets, ref, D, mode, split, mapper, Lvar, lrn0, keep = case
    L = READ_SETS[sets[0]][1]
    paired = mode == "PAIRED"
    lst = os.path.join(tmp, "list.txt")
    with open(lst, "w") as f:
        f.write("%d\n" % len(sets))
        for s in sets:
            n = READ_SETS[s][2]
            if paired:
                f.write("%s\t%s\t%d\n" % (os.path.join(INP, s + "_1.fq"), os.path.join(INP, s + "_2.fq"), n))
            else:
                f.write("%s\t%d\n" % (os.path.join(INP, s + "_1.fq"), n))
    outs = [os.path.join(tmp, x) for x in ("reads", "perfect", "lrn")]
    for o in outs:
        if os.path.exists(o):
            os.remove(o)
    refarg = os.path.join(HERE, "inputs", ref + ".fa") if ref else "NOREF"
    cmd = [exe, lst, str(D), outs[0], refarg, "60", str(L), outs[1], "0", outs[2], mode, split, mapper, "NONE", "0",
           os.path.join(tmp, "kill"), str(lrn0)] + list(extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return [open(o, "rb").read() if os.path.exists(o) else None for o in outs], r


def main():
    os.makedirs(INP, exist_ok=True)
    os.makedirs(OUT, exist_ok=True)
    for name, (ref, L, n, seed, paired) in READ_SETS.items():
        for m, text in enumerate(read_set(ref, L, n, seed, paired)):
            open(os.path.join(INP, "%s_%d.fq" % (name, m + 1)), "w").write(text)
    manifest = {}
    tmp = tempfile.mkdtemp(prefix="collapse_golden_")
    for name, case in CASES.items():
        exe = os.path.join(ROOT, "oracle", "_ref", "collapse_L" if case[6] else "collapse")
        outs, r = run_case(exe, case, tmp)
        assert all(o is not None for o in outs), (name, r.returncode, r.stdout[-500:])
        manifest[name] = {"sha256": [sha(o) for o in outs], "bytes": [len(o) for o in outs], "returncode": r.returncode}
        if case[8]:
            d = os.path.join(OUT, name)
            os.makedirs(d, exist_ok=True)
            for fn, o in zip(("reads", "perfect", "lrn"), outs):
                open(os.path.join(d, fn), "wb").write(o)
        print(name, manifest[name]["bytes"], r.returncode, flush=True)
    json.dump(manifest, open(os.path.join(HERE, "collapse_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
