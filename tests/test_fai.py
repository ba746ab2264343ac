"""CPU: oracle/fai_oracle (the restatement of `cora faiGenerate`, cora.cpp:1006-1029) against the .fai files the real
command wrote for every crafted FASTA under tests/golden/fai_cases and every golden input (make_fai_golden.py)."""
import glob
import os
import subprocess

import pytest

import helpers

FAI_ORACLE = os.path.join(helpers.ORACLE_DIR, "fai_oracle")


def fai_cases():
    out = []
    for fa in sorted(glob.glob(os.path.join(helpers.GOLDEN, "fai_cases", "*.fa"))):
        out.append((fa, fa + ".fai"))
    for fa in sorted(glob.glob(os.path.join(helpers.GOLDEN, "inputs", "*.fa"))):
        out.append((fa, os.path.join(helpers.GOLDEN, "fai_inputs", os.path.basename(fa) + ".fai")))
    return out


def build_fai_oracle():
    src = os.path.join(helpers.ORACLE_DIR, "fai_oracle.c")
    if not os.path.exists(FAI_ORACLE) or os.path.getmtime(FAI_ORACLE) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "fai_oracle"], stdout=subprocess.DEVNULL)
    return FAI_ORACLE


def test_there_are_cases():
    assert len(fai_cases()) >= 25


@pytest.mark.parametrize("fa,fai", fai_cases(), ids=lambda p: os.path.basename(p))
def test_oracle_matches_the_real_command(fa, fai):
    got = subprocess.run([build_fai_oracle(), fa], stdout=subprocess.PIPE, check=True).stdout
    assert got == open(fai, "rb").read()


def test_committed_input_fai_files_are_what_the_command_writes():
    """The .fai files beside the golden inputs (used by every table test) equal the command's in columns 1-2, the only
    ones homTable_setup reads (HTS:171-197)."""
    for fa in sorted(glob.glob(os.path.join(helpers.GOLDEN, "inputs", "*.fa"))):
        a = [l.split("\t")[:2] for l in open(fa + ".fai").read().splitlines()]
        b = [l.split("\t")[:2] for l in open(os.path.join(helpers.GOLDEN, "fai_inputs", os.path.basename(fa) + ".fai")).read().splitlines()]
        assert a == b, fa
