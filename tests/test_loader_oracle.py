"""CPU: the loader restatement (oracle/loader_oracle.c) against the REAL reference's loaders.

The pins are sha256 values of what /root/reference/mappingInference.cpp's LoadHomologyTables built, dumped by
oracle/_ref/ref_loader* (tests/golden/make_loader_golden.py); for the hand-made tables the reference's whole dump is
committed beside the tables (tests/golden/loader_cases/<name>/reference.dump)."""
import os

import numpy as np
import pytest

import helpers as H
from cora_amd.tables import ARRAYS

CASES = sorted(n for n, c in H.MANIFEST["cases"].items() if not c.get("hash_only") and "loader_sha256" in c)
LCASES = sorted(H.MANIFEST["loader_cases"])


def oracle_dump(exact, inexact, d, lens, L, name, exact_only=False, lookup_all=None):
    out = os.path.join(H.scratch(), "loader_%s%s.dump" % (name, "_eo" if exact_only else ""))
    r = H.run_loader_oracle(exact, inexact, d, lens, out, L=L, exact_only=exact_only, lookup_all=lookup_all)
    assert r.returncode == 0, r.stdout
    return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_loader_on_golden_tables(name):
    c, dirn = H.golden_case(name)
    lk = os.path.join(H.scratch(), "lookup_%s.bin" % name)
    out = oracle_dump(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], c["L"], name, lookup_all=lk)
    assert H.sha256(out) == c["loader_sha256"]
    assert H.sha256(lk) == c["lookup_sha256"], "RetrieveOffsetAndDirInfo over every position differs from the reference's"


@pytest.mark.parametrize("name", LCASES)
def test_oracle_matches_reference_loader_on_handmade_tables(name):
    c = H.MANIFEST["loader_cases"][name]
    dirn = os.path.join(H.LOADER_CASES, name)
    lk = os.path.join(H.scratch(), "lookup_%s.bin" % name)
    out = oracle_dump(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], c["L"], name, lookup_all=lk)
    assert H.sha256(lk) == c["lookup_sha256"], "RetrieveOffsetAndDirInfo over every position differs from the reference's"
    H.assert_same_arrays(H.read_dump(out), H.read_dump(os.path.join(dirn, "reference.dump")), name)
    assert H.sha256(out) == c["loader_sha256"] == H.sha256(os.path.join(dirn, "reference.dump"))
    eo = oracle_dump(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], c["L"], name, exact_only=True)
    assert H.sha256(eo) == c["loader_exact_only_sha256"]


def test_dump_helpers_round_trip():
    dirn = os.path.join(H.LOADER_CASES, "weird_a")
    arrays = H.read_dump(os.path.join(dirn, "reference.dump"))
    assert [a.dtype for a in arrays] == [np.dtype(dt) for _, dt in ARRAYS]
    assert H.dump_sha256(arrays) == H.sha256(os.path.join(dirn, "reference.dump"))
    a = dict(zip([n for n, _ in ARRAYS], arrays))
    assert len(a["eq_off"]) == len(a["eq_chr"]) + 1 and a["eq_off"][-1] == len(a["eqi_pos"])
    assert len(a["hm_off"]) == len(a["hm_chr"]) + 1 and a["hm_off"][-1] == len(a["hmi_pos"])
    assert len(a["equiv_hub"]) == len(a["hom_hub"]) == 400


def test_weird_tables_hold_what_they_are_meant_to():
    """The hand-made hazards really are in the reference's result: a later class wins an overlap, a copy reads an
    already overwritten source, the reverse-lookup walk produces non-trivial values."""
    a = dict(zip([n for n, _ in ARRAYS], H.read_dump(os.path.join(H.LOADER_CASES, "weird_a", "reference.dump"))))
    eh, hh = a["equiv_hub"], a["hom_hub"]
    assert list(eh[99:106]) == [1, 1, 2, 2, 2, 2, 2]                  # positions 100..106: class 2 wins 102..104
    assert hh[80 - 1] == hh[30 - 1] and hh[81 - 1] == hh[31 - 1]      # 80.. copied from 50.., which held 30..'s values
    assert hh[130 - 1] == hh[120 - 1]                                 # node 130's own entry overwritten
    b = dict(zip([n for n, _ in ARRAYS], H.read_dump(os.path.join(H.LOADER_CASES, "weird_b_L", "reference.dump"))))
    assert b["hm_maxrev"].max() >= 4 and (b["hm_maxrev"] == 0).sum() >= 10


def test_oracle_rejects_positions_outside_the_contigs():
    dirn = os.path.join(H.LOADER_CASES, "weird_a")
    r = H.run_loader_oracle(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), 2, [300], os.path.join(H.scratch(), "x.dump"))
    assert r.returncode == 5
