"""Shared test plumbing: golden manifest, input materialisation, oracle runner."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_BIN = os.path.join(ORACLE_DIR, "hts_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "homTable_setup")

MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
_cache_dir = None


def scratch():
    global _cache_dir
    if _cache_dir is None:
        _cache_dir = tempfile.mkdtemp(prefix="cora_tests_")
    return _cache_dir


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def input_path(name):
    """FASTA of a golden input; recipe-only inputs are rebuilt and sha256-checked."""
    meta = MANIFEST["inputs"][name]
    if not meta["recipe_only"]:
        return os.path.join(GOLDEN, "inputs", name + ".fa")
    p = os.path.join(scratch(), name + ".fa")
    if not os.path.exists(p):
        sys.path.insert(0, GOLDEN)
        import make_golden
        from cora_amd import synth
        synth.write_fasta(p, make_golden.INPUTS[name](), 60)
        assert sha256(p) == meta["sha256"], "recipe for %s no longer reproduces the pinned input" % name
    return p


def split_path(kind, n):
    return os.path.join(GOLDEN, "splits", "%s_%d.txt" % (kind, n))


def build_oracle():
    if not os.path.exists(ORACLE_BIN) or os.path.getmtime(ORACLE_BIN) < os.path.getmtime(os.path.join(ORACLE_DIR, "hts_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "hts_oracle", "liboracle_sort.so"], stdout=subprocess.DEVNULL)
    return ORACLE_BIN


def run_tool(binary, fasta, K, d, phys, par, outdir, L=False, extra=()):
    """Runs a homTable_setup-compatible executable; returns (exact_path, inexact_path, proc)."""
    os.makedirs(outdir, exist_ok=True)
    ex, inx = os.path.join(outdir, "Exact"), os.path.join(outdir, "Inexact")
    for p in (ex, inx, ex + ".itemCount", inx + ".itemCount"):
        if os.path.exists(p):
            os.remove(p)
    cmd = [binary, fasta, str(K), phys, ex, inx, par, "BOTH", str(d), "FULL"] + list(extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return ex, inx, r


def run_oracle(fasta, K, d, phys, par, outdir, L=False):
    ex, inx, r = run_tool(build_oracle(), fasta, K, d, phys, par, outdir, extra=(["L"] if L else []))
    assert r.returncode == 0, r.stdout[-2000:]
    return ex, inx


def read(path):
    with open(path, "rb") as f:
        return f.read()


def golden_case(name):
    c = MANIFEST["cases"][name]
    d = os.path.join(GOLDEN, "cases", name)
    return c, d


def assert_same_tables(got_ex, got_inx, want_dir, label=""):
    for got, fn in ((got_ex, "Exact"), (got_inx, "Inexact")):
        w = read(os.path.join(want_dir, fn))
        g = read(got)
        if g != w:
            n = min(len(g), len(w))
            first = next((i for i in range(n) if g[i] != w[i]), n)
            raise AssertionError("%s %s differs: got %d B, want %d B, first difference at byte %d" % (label, fn, len(g), len(w), first))
        wc = read(os.path.join(want_dir, fn + ".itemCount"))
        gc = read(got + ".itemCount")
        assert gc == wc, "%s %s.itemCount: got %r want %r" % (label, fn, gc, wc)


# ---- table loaders ----
LOADER_ORACLE = os.path.join(ORACLE_DIR, "loader_oracle")
LOADER_CASES = os.path.join(GOLDEN, "loader_cases")


def build_loader_oracle():
    src = os.path.join(ORACLE_DIR, "loader_oracle.c")
    if not os.path.exists(LOADER_ORACLE) or os.path.getmtime(LOADER_ORACLE) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "loader_oracle"], stdout=subprocess.DEVNULL)
    return LOADER_ORACLE


def run_loader_oracle(exact, inexact, d, lens, out, L=False, exact_only=False, lookup_all=None):
    cmd = [build_loader_oracle()] + (["-L"] if L else []) + (["--exact-only"] if exact_only else [])
    if lookup_all:
        cmd += ["--lookup-all", lookup_all]
    r = subprocess.run(cmd + [exact, inexact, str(d), out] + [str(x) for x in lens], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r


def read_dump(path):
    """-> the 18 arrays of oracle/loader_dump.h as numpy arrays (dtypes from cora_amd.tables.ARRAYS)."""
    import numpy as np
    from cora_amd.tables import ARRAYS
    raw = read(path)
    out, at = [], 0
    for name, dt in ARRAYS:
        nbytes = int.from_bytes(raw[at:at + 8], "little")
        at += 8
        out.append(np.frombuffer(raw, dtype=dt, count=nbytes // np.dtype(dt).itemsize, offset=at))
        at += nbytes
    assert at == len(raw), "dump has trailing bytes"
    return out


def dump_sha256(arrays):
    """sha256 of the dump file these arrays would make (u64 byte count + data, in order)."""
    h = hashlib.sha256()
    for a in arrays:
        h.update(int(a.nbytes).to_bytes(8, "little"))
        h.update(memoryview(a).cast("B") if a.nbytes else b"")
    return h.hexdigest()


def assert_same_arrays(got, want, label=""):
    import numpy as np
    from cora_amd.tables import ARRAYS
    for (name, _), g, w in zip(ARRAYS, got, want):
        assert len(g) == len(w), "%s %s: %d elements, want %d" % (label, name, len(g), len(w))
        if not np.array_equal(g, w):
            i = int(np.nonzero(g != w)[0][0])
            raise AssertionError("%s %s differs first at [%d]: got %d want %d (%d of %d differ)" % (
                label, name, i, g[i], w[i], int((g != w).sum()), len(w)))
