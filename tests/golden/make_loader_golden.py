#!/usr/bin/env python3
"""Pins for the table LOADERS: runs the real reference's loaders, through the dump harness
oracle/_ref/ref_loader* (built by oracle/build_ref.sh from /root/reference/mappingInference.cpp), over

 * every golden table pair of manifest.json                      -> manifest["cases"][name]["loader_sha256"]
 * hand-made and seeded random tables that no table builder would produce (overlapping classes, a class member that
   covers another class's representative, reverse members, duplicated members, every shape of the reverse-lookup
   walk)                                                         -> tests/golden/loader_cases/<name>/ + manifest["loader_cases"]

Only runs where /root/reference was present when oracle/_ref was built; the outputs are committed, so the tests need
neither.  usage: python tests/golden/make_loader_golden.py [--big]   (--big also pins the hash-only 64 Mbp cases: it
re-runs the reference table build, minutes)
"""
import hashlib
import json
import os
import random
import shutil
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def ref_loader(L, d):
    return os.path.join(REFDIR, "ref_loader" + ("_L" if L else "") + ("_e3" if d == 3 else ""))


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


# ---- table writers (formats F1 / F2 of SURVEY.md section 8a) ----

def chr_bytes(c, L):
    return struct.pack("<H", c) if L else struct.pack("<B", c)


def exact_bytes(records, L):
    """records: [(chr, pos, block, [(dir, chr, pos), ...]), ...]"""
    out = bytearray()
    for c, p, block, members in records:
        out += chr_bytes(c, L) + struct.pack("<I", p)
        for dr, mc, mp in members:
            out += bytes([dr]) + chr_bytes(mc, L) + struct.pack("<I", mp)
        out += b"\xff" + struct.pack("<I", block)
    return bytes(out)


def inexact_bytes(nodes, L, d):
    """nodes: [(chr, pos, [(dir, chr, pos, [off...], blockLen), ...]), ...]"""
    out = bytearray()
    for c, p, items in nodes:
        out += chr_bytes(c, L) + struct.pack("<i", p)
        for dr, tc, tp, offs, bl in items:
            assert len(offs) == d
            out += bytes([dr]) + chr_bytes(tc, L) + struct.pack("<i", tp) + bytes(offs) + struct.pack("<h", bl)
        out += b"\xff"
    return bytes(out)


def write_tables(dirn, records, nodes, L, d):
    os.makedirs(dirn, exist_ok=True)
    open(os.path.join(dirn, "Exact"), "wb").write(exact_bytes(records, L))
    open(os.path.join(dirn, "Exact.itemCount"), "w").write("%d\n" % len(records))
    open(os.path.join(dirn, "Inexact"), "wb").write(inexact_bytes(nodes, L, d))
    open(os.path.join(dirn, "Inexact.itemCount"), "w").write("%d\n" % len(nodes))


# ---- tables no builder writes ----

def item(dr, c, p, d, bl, first_off=None):
    offs = sorted(random.Random(p * 7 + bl).sample(range(1, 60), d))
    if first_off is not None:
        offs[0] = first_off
    return (dr, c, p, offs, bl)


def case_weird_a():
    """One contig of 400.  Exact hazards one by one; the inexact nodes sit where the copies read and write."""
    L, d, lens = 0, 2, [400]
    records = [
        (1, 10, 5, [(0, 1, 100)]),                 # plain forward member
        (1, 20, 5, [(0, 1, 102)]),                 # overlaps the previous member range: the later class wins 102..104
        (1, 30, 3, [(0, 1, 50)]),                  # writes 50..52 ...
        (1, 50, 3, [(0, 1, 80)]),                  # ... which this class then copies FROM (already overwritten)
        (1, 60, 4, [(1, 1, 93)]),                  # reverse member: 93, 92, 91, 90
        (1, 120, 2, [(0, 1, 130)]),                # 130 holds a node of its own: overwritten by the copy
        (1, 140, 1, [(1, 1, 140), (0, 1, 150), (1, 1, 150)]),   # Q2 shape: palindrome listed twice
        (1, 200, 6, [(0, 1, 203)]),                # member range overlaps the class's own representative range
        (1, 220, 4, [(1, 1, 222)]),                # reverse member running back over the representative range
        (1, 300, 3, [(0, 1, 310), (0, 1, 311), (1, 1, 320)]),   # members overlapping each other
        (1, 12, 2, [(0, 1, 350)]),                 # unsorted record, representative range inside an earlier one
    ]
    node_pos = [10, 11, 14, 21, 22, 31, 32, 51, 61, 63, 100, 121, 130, 131, 140, 150, 201, 204, 205, 221, 223, 300, 301, 302, 312]
    nodes = [(1, p, [item(0, 1, 360 + (p % 7), d, 1 + (p % 5))]) for p in node_pos]
    return L, d, lens, records, nodes


def case_weird_b():
    """Reverse lookups (mappingInference.cpp:638-650): every shape of the walk, over three contigs, 2-byte ids."""
    L, d, lens = 1, 1, [300, 300, 120]
    spec = {   # contig -> [(pos, block length of the FIRST item, more block lengths...)]
        1: [(5, 1), (6, 3), (7, 1), (8, 1), (9, 1), (20, 30, 2), (21, 1), (25, 2, 9), (26, 1), (40, 1), (49, 1), (50, 1), (51, 5), (52, 1),
            (55, 2), (56, 1), (57, 1), (100, 60), (110, 3), (111, 1), (112, 1), (113, 1), (130, 40), (150, 1), (159, 1), (160, 1), (169, 1), (170, 1)],
        2: [(1, 50), (2, 1), (3, 1), (50, 1), (51, 1), (298, 3), (299, 2), (300, 1)],
        3: [(60, 61), (61, 1), (90, 2, 31), (120, 1)],
    }
    nodes = []
    for c in sorted(spec):
        for ent in spec[c]:
            p, bls = ent[0], ent[1:]
            nodes.append((c, p, [item(k & 1, 1 + (p + k) % 3, 10 + k, d, bl) for k, bl in enumerate(bls)]))
    records = [(2, 10, 3, [(0, 3, 30), (1, 1, 200)]), (3, 100, 2, [(0, 1, 5)]), (1, 100, 12, [(0, 2, 1)])]
    return L, d, lens, records, nodes


def case_fuzz(seed, L, d, lens, n_rec, n_nodes):
    rng = random.Random(seed)
    nc = len(lens)
    records = []
    for _ in range(n_rec):
        c = rng.randint(1, nc)
        block = rng.choice([1, 1, 1, 2, 3, 5, 8, 20])
        p = rng.randint(1, lens[c - 1] - block + 1)
        members = []
        for _ in range(rng.choice([1, 1, 2, 3, 6])):
            mc = rng.randint(1, nc)
            if rng.random() < 0.5:
                members.append((0, mc, rng.randint(1, lens[mc - 1] - block + 1)))
            else:
                members.append((1, mc, rng.randint(block, lens[mc - 1])))
        records.append((c, p, block, members))
    pos = set()
    while len(pos) < n_nodes:
        c = rng.randint(1, nc)
        if rng.random() < 0.7 and pos:   # clustered, so that the reverse-lookup walk has something to do
            bc, bp = rng.choice(sorted(pos))
            cand = (bc, min(lens[bc - 1], bp + rng.randint(1, 12)))
        else:
            cand = (c, rng.randint(1, lens[c - 1]))
        pos.add(cand)
    nodes = []
    for c, p in sorted(pos):
        items = []
        for k in range(rng.choice([1, 1, 2, 4])):
            tc = rng.randint(1, nc)
            bl = rng.choice([1, 1, 2, 3, 7, 15, 40, 63])
            offs = sorted(rng.sample(range(1, 64), rng.randint(1, d)))
            offs += [0] * (d - len(offs))
            items.append((rng.randint(0, 1), tc, rng.randint(1, lens[tc - 1]), offs, bl))
        nodes.append((c, p, items))
    return L, d, lens, records, nodes


LOADER_CASES = {
    "weird_a": case_weird_a,
    "weird_b_L": case_weird_b,
    "fuzz1_d2": lambda: case_fuzz(1, 0, 2, [500], 60, 120),
    "fuzz2_d3": lambda: case_fuzz(2, 0, 3, [300, 200, 100], 80, 150),
    "fuzz3_d1_L": lambda: case_fuzz(3, 1, 1, [256, 64, 64, 700], 120, 300),
    "fuzz4_d2": lambda: case_fuzz(4, 0, 2, [5000, 3000], 600, 2500),
    "empty_exact": lambda: (0, 2, [200], [], [(1, 7, [item(0, 1, 90, 2, 3)]), (1, 9, [item(1, 1, 50, 2, 1)])]),
    "empty_inexact": lambda: (0, 2, [200], [(1, 10, 4, [(0, 1, 100), (1, 1, 180)])], []),
    "empty_both": lambda: (0, 2, [100], [], []),
}


def contig_lengths(inp):
    import make_golden
    return [len(s) for _, s in make_golden.INPUTS[inp]()]


def run_ref(L, d, K, ex, inx, lens, out, exact_only=False, lookup=None):
    env = dict(os.environ, CORA_REF_EXACT_ONLY="1" if exact_only else "0")
    if lookup:
        env["CORA_REF_LOOKUP_ALL"] = lookup
    r = subprocess.run([ref_loader(L, d), ex, inx, str(K), str(d), out] + [str(x) for x in lens], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit("ref_loader failed (%d): %s" % (r.returncode, r.stdout[-2000:]))


def main():
    big = "--big" in sys.argv
    mpath = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(mpath))
    tmp = tempfile.mkdtemp(prefix="loader_golden_")
    lens_of = {}
    for name, c in sorted(manifest["cases"].items()):
        if c.get("hash_only") and not big:
            continue
        inp = c["input"]
        if inp not in lens_of:
            lens_of[inp] = contig_lengths(inp)
        if c.get("hash_only"):
            import make_golden
            from cora_amd import synth
            fa = os.path.join(tmp, inp + ".fa")
            if not os.path.exists(fa):
                synth.write_fasta(fa, make_golden.INPUTS[inp](), 60)
            dirn = os.path.join(tmp, name)
            os.makedirs(dirn)
            env = dict(os.environ, OMP_DYNAMIC="false")
            cmd = [make_golden.BIN_L if c["L"] else make_golden.BIN, fa, str(c["K"]), make_golden.split_file("phys", c["p"]),
                   os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), make_golden.split_file("par", c["t"]), "BOTH", str(c["d"]), "FULL"]
            subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, check=True)
            assert sha256(os.path.join(dirn, "Inexact")) == c["inexact_sha256"]
        else:
            dirn = os.path.join(HERE, "cases", name)
        out = os.path.join(tmp, name + ".dump")
        lk = os.path.join(tmp, name + ".lookup")
        run_ref(c["L"], c["d"], c["K"], os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), lens_of[inp], out, lookup=lk)
        c["loader_sha256"] = sha256(out)
        c["lookup_sha256"] = sha256(lk)
        os.remove(lk)
        c["contig_lengths"] = lens_of[inp]
        print("%-22s loaded image %10d B  %s" % (name, os.path.getsize(out), c["loader_sha256"][:16]), flush=True)
        os.remove(out)
    manifest["loader_cases"] = {}
    for name, fn in sorted(LOADER_CASES.items()):
        L, d, lens, records, nodes = fn()
        dirn = os.path.join(HERE, "loader_cases", name)
        if os.path.isdir(dirn):
            shutil.rmtree(dirn)
        write_tables(dirn, records, nodes, L, d)
        out = os.path.join(dirn, "reference.dump")
        lk = os.path.join(tmp, name + ".lookup")
        run_ref(L, d, 50, os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), lens, out, lookup=lk)
        eo = os.path.join(tmp, name + ".exact_only.dump")
        run_ref(L, d, 50, os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), lens, eo, exact_only=True)
        manifest["loader_cases"][name] = {"L": L, "d": d, "contig_lengths": lens, "records": len(records), "nodes": len(nodes),
                                          "loader_sha256": sha256(out), "loader_exact_only_sha256": sha256(eo), "lookup_sha256": sha256(lk)}
        print("%-22s %4d records %5d nodes  loaded image %8d B" % (name, len(records), len(nodes), os.path.getsize(out)), flush=True)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
