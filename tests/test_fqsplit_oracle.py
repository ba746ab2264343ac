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
