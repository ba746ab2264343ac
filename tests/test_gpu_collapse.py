"""GPU: the collapse path (cora_hip_collapse_*, through the C-ABI and through the drop-in executables) against the files
the real reference wrote for every mode combination (tests/golden/collapse_manifest.json) and against the oracle on
seeded inputs; forced hash collisions must not change a byte."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from test_collapse_oracle import MANIFEST, build_collapse_oracle, check_frag, check_outputs, check_seeded, mcf, mcg, mcs
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


@pytest.mark.parametrize("request_", ["1", "100", "3000"])
@pytest.mark.parametrize("name", sorted(mcg.CASES))
def test_order_after_the_hash_map_has_grown(name, request_, tmp_path, monkeypatch):
    """The reference's table re-buckets itself from 182 356 234 read-mers on (a human reference has 2.9e9 windows) and prints
    in an order that depends on the whole insertion history.  Here the table starts small (CORA_COLLAPSE_TABLE_REQUEST; the
    oracle gets the same history through CORA_COLLAPSE_GROWTH) and grows up to ten times during each golden case; the
    oracle's order is pinned against the real container (tests/test_collapse_growth.py).  References with letters other
    than A C G T N are among the cases: their windows are elements of the table under keys of their own (collapse.cpp:86-122)
    and shift every later element's number."""
    from test_collapse_growth import growth_env
    case = mcg.CASES[name]
    monkeypatch.setenv("CORA_COLLAPSE_TABLE_REQUEST", request_)
    monkeypatch.setenv("CORA_COLLAPSE_GROWTH", growth_env(int(request_), 1 << 32))
    os.makedirs(tmp_path / "o")
    want, r = mcg.run_case(build_collapse_oracle(), case, str(tmp_path / "o"), extra=(["L"] if case[6] else []))
    assert r.returncode == 0, r.stderr[-300:]
    os.makedirs(tmp_path / "g")
    got, r = mcg.run_case(COLLAPSE_EXE + ("_L" if case[6] else ""), case, str(tmp_path / "g"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])
    assert got == want, "%s: files differ from the oracle's after growth" % name


@pytest.mark.parametrize("name", ["p_full_generic", "s_half_generic", "multi_s_three", "s_full_bwa_d2"])
def test_reference_windows_hashed_without_the_n_test(name, monkeypatch):
    """the hashing loop for reference windows that leaves the N test out (a window with an N is
    never emitted, so the test is dead there).  Round 1 saw a variant like it differ from run to run; re-created, it is
    deterministic and gives the reference's bytes (profiles/collapse_clean_hash.py: 5 read-mer shapes x 4 runs)."""
    monkeypatch.setenv("CORA_COLLAPSE_CLEAN_HASH", "1")
    a, _ = _run_case(mcg.CASES[name])
    check_outputs(name, a)
    b, _ = _run_case(mcg.CASES[name])
    assert a == b


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
    bad = list(args); bad[12] = "AAAA_TTTT_"; bad[13] = "3"                    # signal length other than 2 or 4 (CL:557-561)
    r = subprocess.run([COLLAPSE_EXE] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 19                                                  # collapse.cpp:560


FQSPLIT_EXE = os.path.join(helpers.ROOT, "cora_amd", "bin", "fastqSplit")


@pytest.mark.parametrize("name", sorted(mcf.CASES))
def test_fragmented_chain_matches_the_reference(name, tmp_path):
    """fastqSplit -> collapse once per group (argv[13] = the group's signals, long-read-name numbers carried along), both
    drop-in executables on the GPU: the per-group files of the real reference's chain."""
    exe = os.path.join(helpers.ROOT, "cora_amd", "bin", "collapse_L" if mcf.CASES[name][4] else "collapse")
    outs, codes = mcf.run(FQSPLIT_EXE, exe, name, str(tmp_path))
    assert set(codes) == {0}, codes
    check_frag(name, outs)


@pytest.mark.parametrize("in_hbm", [False, True])
@pytest.mark.parametrize("name", ["f_paired_sig2_g10_bwamem", "f_threeway_paired_g8_L", "f_split_paired_g2_noref"])
def test_fragmented_mode_through_the_cabi(name, in_hbm):
    """in_hbm: the fragments go from fastqSplit to collapse as device memory (cora_hip_collapse_add_reads_device)."""
    from cora_amd import fastqsplit
    fcase, ref, D, mapper, Lvar = mcf.CASES[name]
    sets, mode, n, size = mcf.mfg.CASES[fcase]
    L = mcg.READ_SETS[sets[0]][1]
    sig = mcf.mfg.frag_text(n, size)
    fs = fastqsplit.FastqSplit(sig.encode(), L, mode)
    for s in sets:
        fs.add_reads(open(os.path.join(mcg.INP, s + "_1.fq"), "rb").read())
        fs.add_reads(open(os.path.join(mcg.INP, s + "_2.fq"), "rb").read())
    fs.run()
    tokens = sig.split("\n")[1:1 + n]
    outs, lrn = [], 0
    for g in range(n):
        c = ccol.Collapse(D, L, mcf.MODES[mode][0], mcf.MODES[mode][1], mapper, Lvar, lrn)
        c.set_fragment(tokens[g], size)
        if ref:
            c.set_reference(open(os.path.join(helpers.GOLDEN, "inputs", ref + ".fa"), "rb").read())
        if in_hbm:
            c.add_reads_device(*fs.group_device(g))
        else:
            c.add_reads(fs.group(g))
        c.run()
        o = c.outputs()
        c.close()
        outs.append(o)
        lrn += o[2].count(b"\n")
    fs.close()
    check_frag(name, outs)


def test_one_object_takes_group_after_group():
    """The chain with ONE collapse object (reference scanned and packed once) and the fragments left in HBM."""
    from cora_amd import fastqsplit
    name = "f_paired_g4_bt2"
    fcase, ref, D, mapper, Lvar = mcf.CASES[name]
    sets, mode, n, size = mcf.mfg.CASES[fcase]
    L = mcg.READ_SETS[sets[0]][1]
    sig = mcf.mfg.frag_text(n, size)
    fs = fastqsplit.FastqSplit(sig.encode(), L, mode)
    for s in sets:
        fs.add_reads(open(os.path.join(mcg.INP, s + "_1.fq"), "rb").read())
        fs.add_reads(open(os.path.join(mcg.INP, s + "_2.fq"), "rb").read())
    fs.set_device_only(True)
    fs.run()
    with pytest.raises(hip.CoraHipError):
        fs.group(0)
    tokens = sig.split("\n")[1:1 + n]
    c = ccol.Collapse(D, L, mcf.MODES[mode][0], mcf.MODES[mode][1], mapper, Lvar, 0)
    c.set_reference(open(os.path.join(helpers.GOLDEN, "inputs", ref + ".fa"), "rb").read())
    outs, lrn = [], 0
    for g in range(n):
        c.set_fragment(tokens[g], size)
        c.set_lrn_offset(lrn)
        c.add_reads_device(*fs.group_device(g))
        c.run()
        outs.append(c.outputs())
        lrn += outs[-1][2].count(b"\n")
    c.close()
    fs.close()
    check_frag(name, outs)


@pytest.mark.parametrize("world", [2, 3])
def test_read_chain_sharded_over_ranks(world, tmp_path):
    """Fragment groups dealt over ranks (cora_amd.dist.read_chain_sharded; gloo, all ranks on this GPU): every group's files are
    the reference chain's, long-read-name numbers included although no rank saw the groups before its own."""
    import sys
    cases = ["f_big_noref_g4", "f_paired_sig2_g10_bwamem", "f_threeway_paired_g8_L"]
    script = os.path.join(helpers.ROOT, "tests", "support", "read_chain_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world), CORA_TEST_CASES=",".join(cases), CORA_TEST_OUT=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    for name in cases:
        n = mcf.mfg.CASES[mcf.CASES[name][0]][2]
        got = [[open(os.path.join(str(tmp_path), "%s.%s.split%d" % (name, fn, g)), "rb").read() for fn in ("reads", "perfect", "lrn")] for g in range(n)]
        check_frag(name, got)


@pytest.mark.parametrize("ref,L,split", [("syn", 100, "FULL"), ("fam", 75, "FULL"), ("multi", 60, "FULL"), ("syn", 99, "THREEWAY"),
                                         ("fam", 150, "FULL"), ("syn", 298, "FULL"), ("multi", 129, "FULL")])
def test_every_reference_window_groups_as_in_the_oracle(ref, L, split, tmp_path):
    """No reads at all: the number of distinct reference read-mers must be the oracle's (every window's canonical form and
    hash take part, not only the windows some read matches), and the same on every repetition."""
    fq = tmp_path / "none.fq"
    fq.write_bytes(b"")
    lst = tmp_path / "list"
    lst.write_text("1\n%s\t0\n" % fq)
    fa = os.path.join(helpers.GOLDEN, "inputs", ref + ".fa")
    o = [str(tmp_path / x) for x in ("r", "p", "l")]
    r = subprocess.run([build_collapse_oracle(), str(lst), "3", o[0], fa, "60", str(L), o[1], "0", o[2], "SINGLE", split, "GENERIC", "NONE", "0",
                        str(tmp_path / "k"), "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-300:]
    want = int(r.stderr.split("distinct")[1].split()[0])
    got = []
    for _ in range(3):
        c = ccol.Collapse(3, L, "SINGLE", split, "GENERIC")
        c.set_reference(open(fa, "rb").read())
        c.add_reads(b"")
        c.run()
        got.append(c.stats()["distinct"])
        assert c.outputs() == [b"", b"", b""]
        c.close()
    assert got == [want] * 3, (got, want)


@pytest.mark.parametrize("request_,frag", [("50", None), ("700", None), ("20", ("AC_CA_GT_TG_AA_TT_", 2))])
def test_windows_with_other_letters_take_their_place_in_a_growing_table(request_, frag, tmp_path, monkeypatch):
    """IUPAC letters (both cases), next to each other, next to N runs, at contig ends: every N-free window over one is an
    element of the reference's table under a key of its own -- the window, or its "reverse complement" cut at the NUL that
    stands for such a letter (collapse.cpp:86-122) -- so it moves every later element's number, and with it the order
    after a growth step.  Oracle (which hashes those keys) against the drop-in, whole-input and one fragment group."""
    from test_collapse_growth import growth_env
    from cora_amd import synth
    rs = np.random.RandomState(int(request_))
    R = 40
    contigs = []
    for c in range(3):
        seq = bytearray(bytes(synth.random_seq(rs, 2500 + 300 * c)))
        for _ in range(25):
            p = int(rs.randint(0, len(seq)))
            seq[p] = ord("RYKMSWBDHVrykm"[rs.randint(14)])
        for p in (0, 1, len(seq) - 1, len(seq) - 2, 700, 701, 702):
            seq[p] = ord("RYKM"[rs.randint(4)])
        seq[1200:1203] = b"NNN"
        seq[1210] = ord("S")
        seq[1500:1520] = bytes(seq[1500:1520]).lower()
        contigs.append(("c%d" % c, np.frombuffer(bytes(seq), dtype=np.uint8)))
    fa = str(tmp_path / "o.fa")
    synth.write_fasta(fa, contigs, 60)
    reads = []
    for _ in range(1500):
        s = bytes(contigs[rs.randint(3)][1]).upper()
        p = int(rs.randint(0, len(s) - R))
        r = bytearray(s[p:p + R])
        for k in range(R):
            if chr(r[k]) not in "ACGTN":
                r[k] = ord("ACGT"[rs.randint(4)])
        if rs.rand() < .3:
            r[int(rs.randint(R))] = ord("ACGT"[rs.randint(4)])
        reads.append(r.decode())
    fq = tmp_path / "r.fq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * R) for i, r in enumerate(reads)))
    lst = tmp_path / "l.txt"
    lst.write_text("1\n%s\t%d\n" % (fq, len(reads)))
    monkeypatch.setenv("CORA_COLLAPSE_TABLE_REQUEST", request_)
    monkeypatch.setenv("CORA_COLLAPSE_GROWTH", growth_env(int(request_), 1 << 32))
    inp = str(lst)
    sig, siglen = "NONE", "0"
    if frag:
        # one group's fragment file, as fastqSplit writes it: 8 bytes (2 x the read-mer's number) + the read-mer
        sig, siglen = frag[0], str(frag[1])
        sigs = set(frag[0].split("_")) - {""}
        comp = str.maketrans("ACGTN", "TGCAN")
        with open(tmp_path / "frag.bin", "wb") as f:
            for i, r in enumerate(reads):
                canon = min(r, r[::-1].translate(comp))
                if canon[0] + canon[-1] in sigs:
                    f.write(np.uint64(2 * i).tobytes() + r.encode())
        inp = str(tmp_path / "frag.bin")
    outs = {}
    for who, exe in (("oracle", build_collapse_oracle()), ("gpu", COLLAPSE_EXE)):
        o = [str(tmp_path / (who + x)) for x in ("r", "p", "l")]
        r = subprocess.run([exe, inp, "3", o[0], fa, "60", str(R), o[1], "0", o[2], "SINGLE", "FULL", "GENERIC", sig, siglen, str(tmp_path / "k"), "0"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (who, r.stdout[-300:], r.stderr[-300:])
        outs[who] = [open(x, "rb").read() for x in o]
        if who == "oracle":
            assert int(r.stderr.decode().split("other-letter-keys")[1].split()[0]) > 100
    assert len(outs["oracle"][0]) + len(outs["oracle"][1]) > 1000
    assert outs["gpu"] == outs["oracle"]
