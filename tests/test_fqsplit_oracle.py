"""CPU: oracle/fastqsplit_oracle (restatement of /root/reference/fastqSplit.cpp) against the fragment files the real
reference wrote for every mapping mode, both signal sizes and several group counts (tests/golden/make_fqsplit_golden.py
-> fqsplit_manifest.json)."""
import json
import os
import subprocess
import sys

import pytest

import helpers

sys.path.insert(0, helpers.GOLDEN)
import make_fqsplit_golden as mfg  # noqa: E402

FQSPLIT_ORACLE = os.path.join(helpers.ORACLE_DIR, "fastqsplit_oracle")
MANIFEST = json.load(open(mfg.MANIFEST))


def build_fqsplit_oracle():
    src = os.path.join(helpers.ORACLE_DIR, "fastqsplit_oracle.c")
    if not os.path.exists(FQSPLIT_ORACLE) or os.path.getmtime(FQSPLIT_ORACLE) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "fastqsplit_oracle"], stdout=subprocess.DEVNULL)
    return FQSPLIT_ORACLE


def check_fragments(name, outs):
    assert all(o is not None for o in outs), "%s: a fragment file is missing" % name
    got = mfg.digest(outs)
    assert got["bytes"] == MANIFEST[name]["bytes"], (name, got["bytes"], MANIFEST[name]["bytes"])
    assert got["sha256"] == MANIFEST[name]["sha256"], "%s: fragment files differ from the reference's" % name


def test_manifest_covers_every_case():
    assert sorted(MANIFEST) == sorted(mfg.CASES)
    assert {c[1] for c in mfg.CASES.values()} == {"SINGLE", "SPLIT_SINGLE", "THREEWAY_SINGLE", "PAIRED", "SPLIT_PAIRED", "THREEWAY_PAIRED"}
    assert {c[3] for c in mfg.CASES.values()} == {2, 4}


def test_signal_file_is_what_cora_writes():
    """cora.cpp:335-390 on LPT_4.dat line 3: every 4-mer appears once per group line pair, the last group starts with N_."""
    text = mfg.frag_text(3, 4).split("\n")
    assert text[0] == "3 4" and text[3].startswith("N_") and not text[1].startswith("N_")
    sigs = [s for ln in text[1:4] for s in ln.replace("N_", "", 1).split("_") if s]
    assert set(map(len, sigs)) == {4} and len(set(sigs)) == 256


@pytest.mark.parametrize("name", sorted(mfg.CASES))
def test_oracle_matches_the_reference(name, tmp_path):
    outs, r = mfg.run(build_fqsplit_oracle(), name, str(tmp_path))
    assert r.returncode == 0, r.stderr[-500:]
    check_fragments(name, outs)


def test_signal_table_of_the_library_host_only():
    """cora_hip_fqsplit_parse_signals (no GPU needed): the table it fills sends every signal where the signals file says,
    N-holding signals to the last group, unlisted ones to group 0 (fastqSplit.cpp:54-129, 16-17)."""
    from cora_amd import fastqsplit
    sym = {"N": 0, "A": 1, "C": 2, "G": 3, "T": 4}
    for n, size in ((3, 4), (8, 4), (1, 4), (2, 2), (10, 2)):
        text = mfg.frag_text(n, size)
        p = fastqsplit.parse_signals(text.encode())
        assert (p.signal_size, p.n_groups) == (size, n)
        want = [0] * (5 ** size)
        for g, ln in enumerate(text.split("\n")[1:1 + n]):
            if ln.startswith("N_"):
                for v in range(5 ** size):
                    digits = [(v // 5 ** k) % 5 for k in range(size)]
                    if 0 in digits:
                        want[v] = g
                ln = ln[2:]
            for s in filter(None, ln.split("_")):
                v = 0
                for ch in s:
                    v = v * 5 + sym[ch]
                want[v] = g
        assert list(p.group_of[:5 ** size]) == want
    with pytest.raises(Exception):
        fastqsplit.parse_signals(b"2 3\nAC_GT_\n")            # signal size 3: the reference exits 7
    with pytest.raises(Exception):
        fastqsplit.parse_signals(b"1 2\nAC_G\n")              # not a whole number of signals: the reference asserts
