"""The reference's hash map GROWS (collapse.cpp:640-643: tr1::unordered_map, rehash(200000000); first at 182 356 234 distinct
read-mers -- a human reference alone has 2.9e9 windows), and its output order after a growth step is a function of the
whole insertion history.  CPU part: (1) the growth history that cora_hip_collapse_table_growth derives from libstdc++'s
policy equals what the REAL container does, element by element; (2) the order the oracle derives from hashes and
insertion ranks equals the REAL container's iteration order, for small initial tables that grow several times while a
few thousand keys go in (tests/native/tr1_table_order.cpp is that container, set up as the reference sets up its own)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_collapse_oracle import build_collapse_oracle

NATIVE = os.path.join(helpers.ROOT, "tests", "native")


def container_exe():
    exe = os.path.join(NATIVE, "tr1_table_order")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-w", "-o", exe, src])
    return exe


def growth(request, max_elements):
    lib = C.CDLL(os.path.join(helpers.ROOT, "cora_amd", "lib", "libcora_hip.so"))
    lib.cora_hip_collapse_table_growth.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    at, b = (C.c_uint64 * 64)(), (C.c_uint64 * 64)()
    n = lib.cora_hip_collapse_table_growth(request, max_elements, at, b, 64)
    return [(int(at[i]), int(b[i])) for i in range(n)]


def growth_env(request, max_elements):
    return ",".join("%d:%d" % e for e in growth(request, max_elements))


def real_container(request, keys):
    out = subprocess.run([container_exe(), str(request)], input=("\n".join(keys) + "\n").encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    steps = [tuple(int(x) for x in ln.split()[1:]) for ln in out if ln.startswith("G ")]
    order = [ln[2:] for ln in out if ln.startswith("O ")]
    return steps, order


def test_reference_sized_history():
    g = growth(200000000, 1 << 32)
    assert g[:6] == [(0, 206062531), (182356234, 418451333), (370310930, 849749479), (751990753, 1725587117), (1527068289, 3504151727),
                     (3101019521, 8589934583)]
    assert g[-1][0] > (1 << 32) or len(g) == 6


@pytest.mark.parametrize("request_", [1, 7, 100, 1000, 5000])
def test_growth_history_is_the_real_containers(request_):
    rs = np.random.RandomState(request_)
    keys = ["".join("ACGT"[c] for c in rs.randint(0, 4, 24)) for _ in range(30000)]
    steps, _ = real_container(request_, keys)
    distinct = len(set(keys))
    mine = growth(request_, distinct)
    assert len(steps) >= 3
    assert steps == [e for e in mine if e[0] <= distinct]
    assert mine[-1][0] > distinct or mine[-1] == steps[-1]


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


@pytest.mark.parametrize("request_,n_reads,R", [(1, 3000, 36), (50, 6000, 50), (1000, 9000, 100), (5000, 60000, 64)])
def test_oracle_order_after_growth_is_the_real_containers(request_, n_reads, R, tmp_path):
    """Reads without a reference, every read one read-mer: the oracle prints the distinct read-mers in the table's
    iteration order; the real container, fed the same canonical keys in the same order, must list them alike."""
    rs = np.random.RandomState(1000 + request_)
    pool = ["".join("ACGTN"[c] for c in rs.choice(5, R, p=[.245, .245, .245, .245, .02])) for _ in range(n_reads // 2)]
    reads = [pool[i] for i in rs.randint(0, len(pool), n_reads)]                        # duplicates: found, not inserted
    reads = [r if rs.rand() < .5 else _revcomp(r) for r in reads]
    fq = tmp_path / "r.fq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * R) for i, r in enumerate(reads)))
    lst = tmp_path / "l.txt"
    lst.write_text("1\n%s\t1\n" % fq)
    keys = []
    seen = set()
    for r in reads:
        k = min(r, _revcomp(r))
        if k not in seen:
            seen.add(k)
            keys.append(k)
    env = dict(os.environ, CORA_COLLAPSE_GROWTH=growth_env(request_, len(keys)))
    args = [build_collapse_oracle(), str(lst), "4", str(tmp_path / "reads"), "NOREF", "60", str(R), str(tmp_path / "p"), "0", str(tmp_path / "l"),
            "SINGLE", "FULL", "GENERIC", "NONE", "0", str(tmp_path / "k"), "0"]
    r = subprocess.run(args, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-300:]
    got = [ln for ln in open(tmp_path / "reads").read().split("\n") if ln and ln[0] in "ACGTN" and len(ln) == R]
    steps, order = real_container(request_, keys)
    assert len(steps) >= 3, "the table must have grown"
    assert got == order
