#!/usr/bin/env python3
"""collapse in its fragmented mode (argv[13] = one group's signals, argv[1] = that group's fragment file; the per-group
calls of cora.cpp:1733-1760, long-read-name numbers carried from group to group).  Fragment files come from a
fastqSplit-compatible program over the cases of make_fqsplit_golden.py; expected = sha256 + size of the three files the REAL
reference (oracle/_ref/fastqSplit -> oracle/_ref/collapse[_L]) writes per group (collapse_frag_manifest.json).
Run in the build container only."""
import hashlib, json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MANIFEST = os.path.join(HERE, "collapse_frag_manifest.json")
sys.path.insert(0, HERE)
import make_fqsplit_golden as mfg  # noqa: E402

MODES = {"SINGLE": ("SINGLE", "FULL"), "SPLIT_SINGLE": ("SINGLE", "HALF"), "THREEWAY_SINGLE": ("SINGLE", "THREEWAY"),
         "PAIRED": ("PAIRED", "FULL"), "SPLIT_PAIRED": ("PAIRED", "HALF"), "THREEWAY_PAIRED": ("PAIRED", "THREEWAY")}

CASES = {   # name: (fastqSplit case, reference or None, D, mapper, L variant)
    "f_single_g3": ("single_4_g3", "syn", 3, "GENERIC", False),
    "f_split_single_g5_bwa": ("split_single_4_g5", "syn", 3, "BWA", False),
    "f_paired_g4_bt2": ("paired_4_g4", "syn", 3, "BOWTIE_2", False),
    "f_threeway_paired_g8_L": ("threeway_paired_4_g8", "multi", 4, "GENERIC", True),
    "f_paired_sig2_g10_bwamem": ("paired_2_g10", "syn", 3, "BWA_MEM", False),
    "f_threeway_paired_sig2_g10": ("threeway_paired_2_g10", "syn", 4, "GENERIC", False),
    "f_single_g1": ("single_4_g1", "fam", 3, "GENERIC", False),
    "f_split_paired_g2_noref": ("split_paired_4_g2_odd", None, 3, "GENERIC", False),
    "f_big_noref_g4": ("single_4_g4_big", None, 3, "GENERIC", False),       # long-read-name numbers run on through the groups
    "f_big_g4_bwa": ("single_4_g4_big", "syn", 3, "BWA", False),
    "f_two_samples_g8_d2": ("single_4_g8_two_samples", "syn", 2, "GENERIC", False),
}


def run(split_exe, collapse_exe, name, tmp, extra=()):
    """-> (list over groups of [reads, perfect, lrn] bytes, list of return codes)"""
    fcase, ref, D, mapper, Lvar = CASES[name]
    sets, mode, n, size = mfg.CASES[fcase]
    L = mfg.mcg.READ_SETS[sets[0]][1]
    args, prefix, _ = mfg.materialize(fcase, tmp)
    r = subprocess.run([split_exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (name, r.returncode, r.stdout[-300:], r.stderr[-300:])
    tokens = mfg.frag_text(n, size).split("\n")[1:1 + n]
    inmode, split = MODES[mode]
    refarg = os.path.join(HERE, "inputs", ref + ".fa") if ref else "NOREF"
    outs, codes, lrn = [], [], 0
    for g in range(n):
        o = [os.path.join(tmp, "%s.%s.split%d" % (name, x, g)) for x in ("reads", "perfect", "lrn")]
        for p in o:
            if os.path.exists(p):
                os.remove(p)
        cmd = [collapse_exe, "%s.frag%d" % (prefix, g), str(D), o[0], refarg, "60", str(L), o[1], "0", o[2], inmode, split, mapper,
               tokens[g], str(size), os.path.join(tmp, name + ".kill"), str(lrn)] + list(extra)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        codes.append(r.returncode)
        got = [open(p, "rb").read() if os.path.exists(p) else None for p in o]
        outs.append(got)
        if got[2] is not None:
            lrn += got[2].count(b"\n")                       # cora counts the lines with wc -l (cora.cpp:1762-1774)
    return outs, codes


def digest(outs):
    return [{"sha256": [hashlib.sha256(b).hexdigest() for b in g], "bytes": [len(b) for b in g]} for g in outs]


def main():
    tmp = tempfile.mkdtemp(prefix="collapse_frag_golden_")
    manifest = {}
    for name, case in CASES.items():
        cexe = os.path.join(ROOT, "oracle", "_ref", "collapse_L" if case[4] else "collapse")
        outs, codes = run(os.path.join(ROOT, "oracle", "_ref", "fastqSplit"), cexe, name, tmp)
        assert all(b is not None for g in outs for b in g), (name, codes)
        manifest[name] = digest(outs)
        print(name, [g["bytes"] for g in manifest[name]], codes[:2], flush=True)
    json.dump(manifest, open(MANIFEST, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
