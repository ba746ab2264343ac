#!/usr/bin/env python3
"""fastqSplit cases: the fragment files the REAL reference (oracle/_ref/fastqSplit, built from
/root/reference/fastqSplit.cpp by oracle/build_ref.sh) writes for every mapping mode, both signal sizes and several group
counts, over the read sets of tests/golden/collapse_inputs.  The fragmentation-signal files are written the way
cora.cpp:335-390 writes them from a line of LPT_4.dat (tests/golden/splits/par_<n>.txt holds those lines); the 2-character
form, which cora never issues but the program accepts, is made the same way from LPT_2.dat lines (phys_<n>.txt).
Kept: sha256 + size of every .frag<i> file (fqsplit_manifest.json).  Run in the build container only."""
import hashlib, json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INP = os.path.join(HERE, "collapse_inputs")
MANIFEST = os.path.join(HERE, "fqsplit_manifest.json")
sys.path.insert(0, HERE)
import make_collapse_golden as mcg  # noqa: E402

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def frag_text(n, size):
    """cora.cpp:335-390: per group the signals, a blank, the reverse complement of the whole line, a blank; blanks become
    underscores; the last group gets the N signals ("N_" in front)."""
    src = os.path.join(HERE, "splits", ("par_%d.txt" if size == 4 else "phys_%d.txt") % n)
    lines = open(src).read().split("\n")[1:]
    lines = [ln for ln in lines if ln.strip()]
    assert len(lines) == n, (src, len(lines))
    out = ["%d %d" % (n, size)]
    for i, ln in enumerate(lines):
        rc = "".join(COMP.get(c, c) for c in reversed(ln))
        s = ln + " " + rc + " "
        if i == n - 1:
            s = "N " + s
        out.append(s.replace(" ", "_"))
    return "\n".join(out) + "\n"


CASES = {   # name: (read sets (samples), mapping mode, groups, signal size)
    "single_4_g3": (["syn100s"], "SINGLE", 3, 4),
    "single_4_g8_two_samples": (["syn100s", "syn100s_b"], "SINGLE", 8, 4),
    "split_single_4_g5": (["syn100s"], "SPLIT_SINGLE", 5, 4),
    "threeway_single_4_g7": (["multi60s"], "THREEWAY_SINGLE", 7, 4),
    "paired_4_g4": (["syn100p"], "PAIRED", 4, 4),
    "split_paired_4_g2_odd": (["syn101p"], "SPLIT_PAIRED", 2, 4),
    "threeway_paired_4_g8": (["multi60p"], "THREEWAY_PAIRED", 8, 4),
    "single_4_g1": (["fam75s"], "SINGLE", 1, 4),
    "single_4_g4_big": (["syn100s_big"], "SINGLE", 4, 4),
    "single_2_g2": (["syn100s"], "SINGLE", 2, 2),
    "split_single_2_g10": (["fam75s"], "SPLIT_SINGLE", 10, 2),
    "threeway_single_2_g2": (["syn100s"], "THREEWAY_SINGLE", 2, 2),
    "paired_2_g10": (["syn100p"], "PAIRED", 10, 2),
    "split_paired_2_g1": (["multi60p"], "SPLIT_PAIRED", 1, 2),
    "threeway_paired_2_g10": (["syn101p"], "THREEWAY_PAIRED", 10, 2),      # holds the reference's copy slip (fastqSplit.cpp:579-580)
}


def materialize(name, tmp):
    """-> (the five arguments of fastqSplit, output prefix, group count)"""
    sets, mode, n, size = CASES[name]
    L = mcg.READ_SETS[sets[0]][1]
    frag = os.path.join(tmp, name + ".signals")
    open(frag, "w").write(frag_text(n, size))
    lst = os.path.join(tmp, name + ".list")
    with open(lst, "w") as f:
        f.write("%d\n" % len(sets))
        for s in sets:
            cnt = mcg.READ_SETS[s][2]
            if "PAIRED" in mode:
                f.write("%s\t%s\t%d\n" % (os.path.join(INP, s + "_1.fq"), os.path.join(INP, s + "_2.fq"), cnt))
            else:
                f.write("%s\t%d\n" % (os.path.join(INP, s + "_1.fq"), cnt))
    prefix = os.path.join(tmp, name + ".out")
    for i in range(n):
        if os.path.exists("%s.frag%d" % (prefix, i)):
            os.remove("%s.frag%d" % (prefix, i))
    return [frag, str(L), lst, prefix, mode], prefix, n


def run(exe, name, tmp):
    args, prefix, n = materialize(name, tmp)
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    outs = []
    for i in range(n):
        p = "%s.frag%d" % (prefix, i)
        outs.append(open(p, "rb").read() if os.path.exists(p) else None)
    return outs, r


def digest(outs):
    return {"sha256": [hashlib.sha256(o).hexdigest() for o in outs], "bytes": [len(o) for o in outs]}


def main():
    tmp = tempfile.mkdtemp(prefix="fqsplit_golden_")
    manifest = {}
    for name in CASES:
        outs, r = run(os.path.join(ROOT, "oracle", "_ref", "fastqSplit"), name, tmp)
        assert r.returncode == 0 and all(o is not None for o in outs), (name, r.returncode, r.stdout[-300:], r.stderr[-300:])
        manifest[name] = digest(outs)
        print(name, manifest[name]["bytes"], flush=True)
    json.dump(manifest, open(MANIFEST, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
