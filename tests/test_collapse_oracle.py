"""CPU: oracle/collapse_oracle (restatement of /root/reference/collapse.cpp, whole-input mode) against the files the real
reference wrote for every mode combination (tests/golden/make_collapse_golden.py -> collapse_manifest.json, full bytes
for the kept cases), plus the hash-table constants the output order rests on."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import helpers

sys.path.insert(0, helpers.GOLDEN)
import make_collapse_golden as mcg  # noqa: E402

COLLAPSE_ORACLE = os.path.join(helpers.ORACLE_DIR, "collapse_oracle")
MANIFEST = json.load(open(os.path.join(helpers.GOLDEN, "collapse_manifest.json")))


def build_collapse_oracle():
    src = os.path.join(helpers.ORACLE_DIR, "collapse_oracle.c")
    if not os.path.exists(COLLAPSE_ORACLE) or os.path.getmtime(COLLAPSE_ORACLE) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "collapse_oracle"], stdout=subprocess.DEVNULL)
    return COLLAPSE_ORACLE


def check_outputs(name, outs):
    """outs = [reads, perfect, lrn] bytes of a run of case `name`."""
    want = MANIFEST[name]
    d = os.path.join(helpers.GOLDEN, "collapse_cases", name)
    for i, fn in enumerate(("reads", "perfect", "lrn")):
        assert outs[i] is not None, "%s: no %s file" % (name, fn)
        if os.path.isdir(d):
            w = open(os.path.join(d, fn), "rb").read()
            if outs[i] != w:
                n = min(len(w), len(outs[i]))
                first = next((k for k in range(n) if w[k] != outs[i][k]), n)
                raise AssertionError("%s %s differs: got %d B, want %d B, first difference at byte %d: got %r want %r" % (
                    name, fn, len(outs[i]), len(w), first, outs[i][max(0, first - 30):first + 30], w[max(0, first - 30):first + 30]))
        assert len(outs[i]) == want["bytes"][i], (name, fn, len(outs[i]), want["bytes"][i])
        assert hashlib.sha256(outs[i]).hexdigest() == want["sha256"][i], "%s: %s differs from the reference's" % (name, fn)


def test_manifest_covers_every_case():
    assert sorted(MANIFEST) == sorted(mcg.CASES)
    modes = {(c[3], c[4], c[5]) for c in mcg.CASES.values()}
    assert len(modes) >= 14


@pytest.mark.parametrize("name", sorted(mcg.CASES))
def test_oracle_matches_the_reference(name, tmp_path):
    case = mcg.CASES[name]
    outs, r = mcg.run_case(build_collapse_oracle(), case, str(tmp_path), extra=(["L"] if case[6] else []))
    assert r.returncode == 0, r.stderr[-500:]
    check_outputs(name, outs)


def test_hashtable_constants(tmp_path):
    """Bucket count, hash function and first-rehash point of libstdc++'s tr1 table as the reference configures it
    (collapse.cpp:640-643), measured on the real container of this image."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <tr1/unordered_map>
#include <string>
#include <cstdio>
int main() {
    std::tr1::unordered_map<std::string, int> m;
    float z = m.max_load_factor(); m.max_load_factor(z / 1.13); m.rehash(200000000);
    std::tr1::hash<std::string> h;
    std::tr1::__detail::_Prime_rehash_policy pol(z / 1.13);
    size_t B = pol._M_next_bkt(200000000), lo = 1, hi = 400000000;
    while (lo < hi) { size_t mid = (lo + hi) / 2; std::tr1::__detail::_Prime_rehash_policy q(z / 1.13); q._M_next_resize = 0;
                      if (q._M_need_rehash(B, mid - 1, 1).first) hi = mid; else lo = mid + 1; }
    printf("%zu %zu %zu %zu\n", m.bucket_count(), h(std::string("ACGTN")), B, lo);
}''')
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-O1", "-w", "-o", exe, str(src)])
    got = subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.split()
    fnv = 14695981039346656037
    for c in b"ACGTN":
        fnv = ((fnv ^ c) * 1099511628211) & (2 ** 64 - 1)
    assert [int(x) for x in got] == [206062531, fnv, 206062531, 182356234]
    text = open(os.path.join(helpers.ORACLE_DIR, "collapse_oracle.c")).read()
    assert "g_at[32] = {0, 182356234ULL," in text and "g_buckets[32] = {206062531ULL," in text


def test_oracle_refuses_what_it_does_not_restate(tmp_path):
    case = list(mcg.CASES["s_full_generic"])
    lst = tmp_path / "l.txt"
    bad = tmp_path / "bad.fq"
    bad.write_text("@r\n" + "ACGTX" * 20 + "\n+\n" + "I" * 100 + "\n")
    lst.write_text("1\n%s\t1\n" % bad)
    args = [build_collapse_oracle(), str(lst), "3", str(tmp_path / "r"), "NOREF", "60", "100", str(tmp_path / "p"), "0", str(tmp_path / "l"),
            "SINGLE", "FULL", "GENERIC", "NONE", "0", str(tmp_path / "k"), "0"]
    assert subprocess.run(args, stderr=subprocess.DEVNULL).returncode == 78
    args[13], args[14] = "AA_AC_", "3"                       # fragment signal length other than 2 or 4 (collapse.cpp:557-561)
    assert subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 6
    args[13], args[14] = "AA_AC", "2"                        # not a whole number of signals: the reference asserts
    assert subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 19


import make_collapse_seeded as mcs  # noqa: E402

SEEDED = json.load(open(mcs.MANIFEST))


def check_seeded(name, outs):
    want = SEEDED[name]
    for i, fn in enumerate(("reads", "perfect", "lrn")):
        assert outs[i] is not None, "%s: no %s file" % (name, fn)
        assert len(outs[i]) == want["bytes"][i], (name, fn, len(outs[i]), want["bytes"][i])
        assert hashlib.sha256(outs[i]).hexdigest() == want["sha256"][i], "%s: %s differs from the reference's" % (name, fn)


def test_seeded_manifest_covers_every_case():
    assert sorted(SEEDED) == sorted(mcs.CASES)


@pytest.mark.parametrize("name", sorted(mcs.CASES))
def test_oracle_matches_the_reference_on_seeded_reads(name, tmp_path):
    """Read-mers of 12 .. 298 bases, many N, long names: the restatement against the real reference's sha256."""
    outs, r = mcs.run(build_collapse_oracle(), name, str(tmp_path))
    assert r.returncode == 0, r.stderr[-500:]
    check_seeded(name, outs)


import make_collapse_frag_golden as mcf  # noqa: E402

FRAG = json.load(open(mcf.MANIFEST))


def check_frag(name, outs):
    assert all(b is not None for g in outs for b in g), "%s: an output file is missing" % name
    got = mcf.digest(outs)
    assert [g["bytes"] for g in got] == [g["bytes"] for g in FRAG[name]], (name, [g["bytes"] for g in got], [g["bytes"] for g in FRAG[name]])
    assert got == FRAG[name], "%s: files differ from the reference's" % name


@pytest.mark.parametrize("name", sorted(mcf.CASES))
def test_oracles_match_the_reference_in_fragmented_mode(name, tmp_path):
    """fastqsplit_oracle -> collapse_oracle per group (argv[13] = the group's signals), long-read-name numbers carried along:
    the per-group files of the real fastqSplit -> collapse chain."""
    subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "fastqsplit_oracle"], stdout=subprocess.DEVNULL)
    outs, codes = mcf.run(os.path.join(helpers.ORACLE_DIR, "fastqsplit_oracle"), build_collapse_oracle(), name, str(tmp_path),
                          extra=(["L"] if mcf.CASES[name][4] else []))
    assert set(codes) == {0}, codes
    check_frag(name, outs)
