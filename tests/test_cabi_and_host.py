"""CPU-side checks: the C-ABI library loads and exports every symbol include/cora_hip.h declares,
the executables keep the reference's argument errors, host set-up logic, and the oracle's
std::sort restatement against the real libstdc++."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import helpers
from cora_amd import hip, synth

ROOT = helpers.ROOT


def _built():
    if not os.path.exists(hip.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cora_amd", "csrc"), "-j", "4", "all"], stdout=subprocess.DEVNULL)


def test_library_exports_every_declared_symbol():
    _built()
    hdr = open(os.path.join(ROOT, "include", "cora_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(cora_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libcora_hip.so does not export %s" % name
    assert sorted(hip.EXPORTS) == declared


def test_struct_layouts_match_header():
    assert ctypes.sizeof(hip.Contig) == 16
    assert ctypes.sizeof(hip.Params) == 6 * 4 + 16 + 256
    assert ctypes.sizeof(hip.Stats) == 11 * 8 + 7 * 4 + 4 + 8 + 16 + 16 + 8      # + seed_sketch, verify_slots, pool_peak_bytes


def test_executable_argument_errors_need_no_gpu(tmp_path):
    _built()
    exe = os.path.join(hip.BIN_DIR, "homTable_setup")
    assert subprocess.run([exe], stdout=subprocess.DEVNULL).returncode == 19                        # HTS:2392-2405
    args = ["ref.fa", "70", "p", "E", "I", "t", "BOTH", "2", "FULL"]
    assert subprocess.run([exe] + args, stdout=subprocess.DEVNULL).returncode == 65                   # HTS:2418-2424
    args[1] = "50"
    assert subprocess.run([exe] + args, stdout=subprocess.DEVNULL).returncode == 66                   # HTS:2433-2437
    args[2] = helpers.split_path("phys", 1)
    assert subprocess.run([exe] + args, stdout=subprocess.DEVNULL).returncode == 67                   # HTS:2452-2456
    args[5] = helpers.split_path("par", 1)
    args[3], args[4] = str(tmp_path / "E"), str(tmp_path / "I")
    assert subprocess.run([exe] + args, stdout=subprocess.DEVNULL).returncode == 50                   # HTS:148-153
    fa = str(tmp_path / "x.fa")
    open(fa, "w").write(">a\nACGT\n")
    args[0] = fa
    assert subprocess.run([exe] + args, stdout=subprocess.DEVNULL).returncode == 52                   # no .fai, HTS:171-179


def test_collapse_argument_errors_need_no_gpu():
    """Exit codes of the reference's argument checks that come before any device work (collapse.cpp:512-628)."""
    _built()
    exe = os.path.join(hip.BIN_DIR, "collapse")
    assert subprocess.run([exe], stdout=subprocess.DEVNULL).returncode == 10                         # CL:512-533
    args = ["list", "5", "out", "NOREF", "60", "100", "perfect", "x", "lrn", "SINGLE", "FULL", "GENERIC", "NONE", "0", "kill", "0"]
    bad = list(args); bad[12] = "AAAA_TTTT_"; bad[13] = "3"
    assert subprocess.run([exe] + bad, stdout=subprocess.DEVNULL).returncode == 19                   # signal length not 2 or 4: CL:557-561
    bad = list(args); bad[10] = "QUARTER"
    assert subprocess.run([exe] + bad, stdout=subprocess.DEVNULL).returncode == 9                    # CL:584-588
    bad = list(args); bad[9] = "TRIPLE"
    assert subprocess.run([exe] + bad, stdout=subprocess.DEVNULL).returncode == 8                    # CL:624-628


def test_arena_range_bookkeeping_on_the_host(tmp_path):
    """DevPool's arena ranges (csrc/arena_ranges.hpp) with made-up base addresses that are 2 MB- but not 1 GB-aligned: nothing is
    handed out below or beyond the arena, everything is aligned on its absolute address, releases coalesce (round-2 advice)."""
    exe = str(tmp_path / "arena_ranges_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "arena_ranges_test.cpp")])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:]


def test_no_cpu_fallback_without_a_device():
    _built()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.CoraHipError):
        hip.init(0)


def test_split_file_parsing_matches_fixture_semantics():
    cnt, tab = hip.parse_split_file(helpers.split_path("phys", 10), 2)
    assert cnt == 10 and all(1 <= g <= 10 for g in tab)
    idx = lambda s: sum("ACGT".index(c) * 4 ** (len(s) - 1 - i) for i, c in enumerate(s))
    assert tab[idx("AA")] == tab[idx("TT")] and tab[idx("AC")] == tab[idx("GT")]                      # reverse complements share a group
    cnt4, tab4 = hip.parse_split_file(helpers.split_path("par", 8), 4)
    assert cnt4 == 8 and all(1 <= g <= 8 for g in tab4)
    assert tab4[idx("AAAC")] == tab4[idx("GTTT")]


def test_round_robin_splits_cover_every_kmer(tmp_path):
    for k, n in ((2, 1), (2, 10), (4, 8), (4, 24)):
        p = str(tmp_path / ("s%d_%d" % (k, n)))
        synth.write_round_robin_splits(p, k, n)
        cnt, tab = hip.parse_split_file(p, k)
        assert cnt == n and min(tab) >= 1 and max(tab) == n


def test_fasta_geometry_and_image(tmp_path):
    rs = np.random.RandomState(0)
    contigs = [("a", synth.random_seq(rs, 125)), ("b", synth.random_seq(rs, 60)), ("c", synth.random_seq(rs, 61))]
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, contigs, 60)
    image = open(fa, "rb").read()
    img2, W2, geom2 = synth.fasta_image(contigs, 60)
    assert img2.tobytes() == image
    W, geom = hip.fasta_geometry(image, fa + ".fai")
    assert W == 60 and geom == geom2
    for (off, ln), (_, seq) in zip(geom, contigs):
        got = bytes(image[off + i + i // W] for i in range(ln))
        assert got == seq.tobytes()
    # the .fai written is the 5-column form cora's faiGenerate produces (cora.cpp:1023)
    rows = [l.split("\t") for l in open(fa + ".fai").read().strip().split("\n")]
    assert [r[0] for r in rows] == ["a", "b", "c"] and [int(r[1]) for r in rows] == [125, 60, 61]
    assert all(int(r[3]) == 60 and int(r[4]) == 61 for r in rows)


def test_recipe_only_inputs_still_reproduce():
    for name, meta in helpers.MANIFEST["inputs"].items():
        if meta["recipe_only"] and meta["bases"] <= 10_000_000:
            helpers.input_path(name)        # asserts the sha256


# ------------------------------------------------------------------ oracle's std::sort restatement
class Rec(ctypes.Structure):
    _fields_ = [("key", ctypes.c_int32), ("idx", ctypes.c_int32)]


LESS = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(Rec), ctypes.POINTER(Rec))


@pytest.fixture(scope="module")
def sort_libs(tmp_path_factory):
    helpers.build_oracle()
    d = tmp_path_factory.mktemp("sortref")
    so = str(d / "libstd_sort_ref.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "support", "std_sort_ref.cpp")])
    ref = ctypes.CDLL(so)
    ora = ctypes.CDLL(os.path.join(helpers.ORACLE_DIR, "liboracle_sort.so"))
    ora.oracle_std_sort.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_size_t, LESS, ctypes.c_long]
    return ref, ora


def _recs(keys):
    arr = (Rec * len(keys))()
    for i, k in enumerate(keys):
        arr[i].key, arr[i].idx = int(k), i
    return arr


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 18, 33, 100, 257, 1000, 5000])
@pytest.mark.parametrize("distinct", [1, 2, 3, 7, 1000000])
def test_oracle_sort_equals_libstdcxx(sort_libs, n, distinct):
    ref, ora = sort_libs
    rs = np.random.RandomState(n * 31 + distinct)
    keys = rs.randint(0, distinct, size=n)
    for desc in (False, True):
        a, b = _recs(keys), _recs(keys)
        (ref.ref_std_sort_desc if desc else ref.ref_std_sort_asc)(a, n)
        less = LESS((lambda x, y: int(x[0].key > y[0].key)) if desc else (lambda x, y: int(x[0].key < y[0].key)))
        ora.oracle_std_sort(b, n, ctypes.sizeof(Rec), less, -1)
        assert [(r.key, r.idx) for r in a] == [(r.key, r.idx) for r in b]


def test_oracle_heap_fallback_equals_libstdcxx(sort_libs):
    ref, ora = sort_libs
    rs = np.random.RandomState(5)
    for n in (17, 40, 333):
        keys = rs.randint(0, 9, size=n)
        a, b = _recs(keys), _recs(keys)
        ref.ref_heap_sort_asc(a, n)
        ora.oracle_std_sort(b, n, ctypes.sizeof(Rec), LESS(lambda x, y: int(x[0].key < y[0].key)), 0)   # depth 0: heap sort at once
        assert [(r.key, r.idx) for r in a] == [(r.key, r.idx) for r in b]
