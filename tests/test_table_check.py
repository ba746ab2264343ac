"""oracle/table_check (semantic checker of the two tables, test infrastructure) is pinned here: every table the REAL
reference wrote (tests/golden/cases) must pass it, and tables with a single damaged field must fail it.  The GPU tests
then use it where no byte-exact expectation exists (references too large for the oracle)."""
import os
import shutil
import subprocess

import pytest

import helpers

CHECK = os.path.join(helpers.ORACLE_DIR, "table_check")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "table_check"], stdout=subprocess.DEVNULL)


def run_check(fa, K, d, ex, inx, L, symmetric=False):
    return subprocess.run([CHECK, fa, str(K), str(d), ex, inx] + (["L"] if L else []) + (["--symmetric"] if symmetric else []),
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("name", sorted(n for n, c in helpers.MANIFEST["cases"].items() if not c.get("hash_only")))
def test_reference_tables_pass(name):
    c, want = helpers.golden_case(name)
    # where none of the reference's order-dependent quirks acts, its links are symmetric as well
    quirky = name.startswith(("pal", "lowcx", "multi"))
    r = run_check(helpers.input_path(c["input"]), c["K"], c["d"], os.path.join(want, "Exact"), os.path.join(want, "Inexact"), c["L"],
                  symmetric=not quirky)
    assert r.returncode == 0, r.stdout
    assert ("NOT symmetric" in r.stdout) == quirky, r.stdout


def _damaged(tmp_path, name, table, offset_from, xor):
    c, want = helpers.golden_case(name)
    d = str(tmp_path / "case")
    shutil.copytree(want, d)
    path = os.path.join(d, table)
    b = bytearray(open(path, "rb").read())
    b[offset_from(b, c)] ^= xor
    open(path, "wb").write(bytes(b))
    return run_check(helpers.input_path(c["input"]), c["K"], c["d"], os.path.join(d, "Exact"), os.path.join(d, "Inexact"), c["L"])


def test_damaged_tables_fail(tmp_path):
    # F2, first node of a 1-byte-contig case: chr(1) pos(4) | dir(1) chr(1) pos(4) off1 off2 blockLen(2)
    r = _damaged(tmp_path / "a", "syn_k50_d2", "Inexact", lambda b, c: 1 + 4 + 1 + 1 + 4, 0x01)          # first mismatch offset
    assert r.returncode == 1 and "offsets" in r.stdout, r.stdout
    r = _damaged(tmp_path / "b", "syn_k50_d2", "Inexact", lambda b, c: 1 + 4 + 1 + 1, 0x04)              # target position
    assert r.returncode == 1, r.stdout
    r = _damaged(tmp_path / "c", "syn_k50_d2", "Inexact", lambda b, c: 1 + 4, 0x01)                      # direction
    assert r.returncode == 1, r.stdout
    r = _damaged(tmp_path / "d", "syn_k50_d2", "Exact", lambda b, c: 1 + 4 + 1 + 1, 0x02)                # an entry's position
    assert r.returncode == 1 and "Exact" in r.stdout, r.stdout
    # a wrong record count
    c, want = helpers.golden_case("syn_k50_d2")
    d = str(tmp_path / "e")
    shutil.copytree(want, d)
    open(os.path.join(d, "Inexact.itemCount"), "w").write("1\n")
    r = run_check(helpers.input_path(c["input"]), c["K"], c["d"], os.path.join(d, "Exact"), os.path.join(d, "Inexact"), c["L"])
    assert r.returncode == 1 and "itemCount" in r.stdout, r.stdout
