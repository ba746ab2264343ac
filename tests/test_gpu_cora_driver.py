"""The UNMODIFIED reference driver over the drop-in executables: `cora coraIndex` (cora.cpp:917-1005 parses, :295-314
writes the split files from LPT_2.dat / LPT_4.dat, :488-493 switches to homTable_setup_L above 127 contigs, :552-569
refuses existing outputs, :1581-1601 runs `homTable_setup ... BOTH d FULL` through system() and then removes the
reference's intermediates).  oracle/_ref/cora is the reference's own cora.cpp compiled by oracle/build_ref.sh; the
executables beside it are cora_amd/bin/homTable_setup[_L]; LPT_2.dat / LPT_4.dat are rebuilt from the committed split
files (each is one line of the reference's table with X for the line breaks, cora.cpp:108-166)."""
import os
import shutil
import subprocess

import pytest

import helpers
from cora_amd import hip

pytestmark = pytest.mark.gpu
CORA = os.path.join(helpers.ORACLE_DIR, "_ref", "cora")


def _install(dirn):
    os.makedirs(dirn)
    shutil.copy(CORA, os.path.join(dirn, "cora"))
    for exe in ("homTable_setup", "homTable_setup_L"):
        os.symlink(os.path.join(hip.BIN_DIR, exe), os.path.join(dirn, exe))       # $ORIGIN-relative rpath: the link resolves to the real file
    for which, kind, lines in ((2, "phys", 10), (4, "par", 24)):
        with open(os.path.join(dirn, "LPT_%d.dat" % which), "w") as f:
            for n in range(1, lines + 1):
                p = helpers.split_path(kind, n)
                if os.path.exists(p):
                    body = open(p).read().split("\n", 1)[1]
                    f.write(body.rstrip("\n").replace("\n", "X") + "\n")
                else:
                    f.write("line %d of the table is not among the committed split files\n" % n)
    return os.path.join(dirn, "cora")


def _run(cora, cwd, args):
    return subprocess.run([cora] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)


@pytest.mark.skipif(not os.path.exists(CORA), reason="oracle/_ref/cora is built by oracle/build_ref.sh where /root/reference exists")
@pytest.mark.parametrize("name", ["syn_k50_d2_p10t8", "multi_k50_d2_L", "syn_k64_d3_p10t24"])
def test_cora_coraindex_runs_the_drop_in(name, tmp_path):
    c, want = helpers.golden_case(name)
    cora = _install(str(tmp_path / "install"))
    work = str(tmp_path / "work")
    os.makedirs(work)
    ref = os.path.join(work, "ref.fa")
    shutil.copy(helpers.input_path(c["input"]), ref)
    shutil.copy(helpers.input_path(c["input"]) + ".fai", ref + ".fai")
    args = ["coraIndex", "-K", str(c["K"]), "-H", str(c["d"]), "-p", str(c["p"]), "-t", str(c["t"]), ref, "Exact", "Inexact"]
    r = _run(cora, work, args)
    assert r.returncode == 0, r.stdout[-3000:]
    assert ("homTable_setup_L " if c["L"] else "homTable_setup ") in r.stdout          # the _L switch (cora.cpp:488-493)
    assert "WARNING: last command returned non-zero exit code" not in r.stdout         # neither the build nor cora's rm of the intermediates
    assert "No such file" not in r.stdout
    for fn in ("Exact", "Exact.itemCount", "Inexact", "Inexact.itemCount"):
        assert helpers.read(os.path.join(work, fn)) == helpers.read(os.path.join(want, fn)), fn
    left = sorted(os.listdir(work))
    assert not [f for f in left if "_preCompact_toDelete" in f], left                  # cora removed the placeholders (cora.cpp:1590-1595)
    # the split files cora wrote are the committed ones (cora.cpp:295-314)
    tmpd = os.path.join(work, "__temporary_CORA_files")
    assert helpers.read(os.path.join(tmpd, "__TEMP__Aux_Physical_Splits_File")) == helpers.read(helpers.split_path("phys", c["p"]))
    assert helpers.read(os.path.join(tmpd, "__TEMP__Aux_Parallel_Splits_File")) == helpers.read(helpers.split_path("par", c["t"]))
    # a second run must not overwrite the tables (cora.cpp:552-569)
    before = helpers.read(os.path.join(work, "Inexact"))
    r2 = _run(cora, work, args)
    assert r2.returncode in (8, 9), (r2.returncode, r2.stdout[-1000:])
    assert helpers.read(os.path.join(work, "Inexact")) == before
