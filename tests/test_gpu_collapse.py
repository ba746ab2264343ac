"""GPU: the collapse path (cora_hip_collapse_*, through the C-ABI and through the drop-in executables) against the files
the real reference wrote for every mode combination (tests/golden/collapse_manifest.json) and against the oracle on
seeded inputs; forced hash collisions must not change a byte."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from test_collapse_oracle import MANIFEST, build_collapse_oracle, check_outputs, check_seeded, mcg, mcs
from cora_amd import collapse as ccol, hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    hip.init(0)


def _run_case(case):
    sets, ref, D, mode, split, mapper, Lvar, lrn0, keep = case
    L = mcg.READ_SETS[sets[0]][1]
    c = ccol.Collapse(D, L, mode, split, mapper, Lvar, lrn0)
    try:
        if ref:
            c.set_reference(open(os.path.join(helpers.GOLDEN, "inputs", ref + ".fa"), "rb").read())
        for s in sets:
            c.add_reads(open(os.path.join(mcg.INP, s + "_1.fq"), "rb").read())
            if mode == "PAIRED":
                c.add_reads(open(os.path.join(mcg.INP, s + "_2.fq"), "rb").read())
        c.run()
        return c.outputs(), c.stats()
    finally:
        c.close()


def test_table_geometry_is_libstdcxx_own():
    assert ccol.table_geometry() == (206062531, 182356234)


@pytest.mark.parametrize("name", sorted(mcg.CASES))
def test_matches_the_reference(name):
    outs, st = _run_case(mcg.CASES[name])
    check_outputs(name, outs)
    assert st["hash_collisions"] == 0


@pytest.mark.parametrize("bits", ["8", "16"])
@pytest.mark.parametrize("name", ["p_full_generic", "s_half_generic", "multi_s_three", "p_full_noref", "s_full_bwa_d2"])
def test_forced_hash_collisions_change_nothing(name, bits, monkeypatch):
    monkeypatch.setenv("CORA_COLLAPSE_HASH_BITS", bits)
    outs, st = _run_case(mcg.CASES[name])
    assert st["hash_collisions"] > 0
    check_outputs(name, outs)


COLLAPSE_EXE = os.path.join(helpers.ROOT, "cora_amd", "bin", "collapse")


@pytest.mark.parametrize("name", sorted(mcs.CASES))
def test_executable_matches_the_reference_on_seeded_reads(name, tmp_path):
    """The drop-in executable (the reference's sixteen arguments) on seeded reads whose read-mers span every word count of
    the packed path (12 .. 298 bases), with many N and names long enough for the long-read-name file."""
    outs, r = mcs.run(COLLAPSE_EXE, name, str(tmp_path))
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-500:])
    check_seeded(name, outs)
    assert os.path.exists(os.path.join(str(tmp_path), name + ".kill"))       # the kill-signal file of CL:2420-2424


def test_executable_argument_errors(tmp_path):
    r = subprocess.run([COLLAPSE_EXE, "x"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    args, _ = mcs.materialize("L64_full", str(tmp_path))
    bad = list(args); bad[10] = "QUARTER"                                      # split mode (CL:584-588)
    r = subprocess.run([COLLAPSE_EXE] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    bad = list(args); bad[12] = "FRAGMENT"                                     # fragmented mode is refused, not mis-run
    r = subprocess.run([COLLAPSE_EXE] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 19
