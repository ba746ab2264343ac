"""Pins the CPU restatement (oracle/hts_oracle.c) to tables written by the real
reference binary for the same inputs (tests/golden/, made by make_golden.py)."""
import os

import pytest

import helpers

SMALL = [n for n, c in sorted(helpers.MANIFEST["cases"].items()) if not c.get("hash_only")]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_tables(name, tmp_path):
    c, want = helpers.golden_case(name)
    ex, inx = helpers.run_oracle(helpers.input_path(c["input"]), c["K"], c["d"],
                                 helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]),
                                 str(tmp_path), L=bool(c["L"]))
    helpers.assert_same_tables(ex, inx, want, name)
    # cora removes these afterwards (cora.cpp:1590-1595): they must exist
    assert os.path.exists(ex + "_preCompact_toDelete")
    assert os.path.exists(inx + "_preCompact_toDelete_t1")
