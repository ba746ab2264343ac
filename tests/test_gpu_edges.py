"""Edge inputs of the table build on the GPU: what the reference refuses or mis-handles must be refused here with the
documented status; what it handles (many tiny contigs, no repeats at all, long low-complexity stretches, IUPAC letters)
must match the oracle byte for byte, through the plain run and through the sharded stages."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from cora_amd import hip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    hip.init(0)


def _build(fa, K, d, p=1, t=1, L=False, sharded=False):
    job = hip.Job(K, d, helpers.split_path("phys", p), helpers.split_path("par", t), L)
    try:
        job.set_reference_file(fa)
        if sharded:
            import torch
            from cora_amd import dist as cdist
            cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0))).run()
        else:
            job.run()
        return job.tables()
    finally:
        job.close()


def test_ragged_fasta_is_refused_by_the_fai_path_and_read_by_the_scan(tmp_path):
    """The reference's loader assumes one line width (HTS:96-142) and silently mis-loads anything else.  The C-ABI
    entry that mirrors it (.fai + line width) refuses such a file; the executable, which scans the file on the GPU
    (cora_hip_fasta_scan), builds the tables of the same sequence in fixed-width layout."""
    rs = np.random.RandomState(1)
    seq = synth.random_seq(rs, 3000)
    seq[1000:1400] = seq[200:600]                # something to find
    fa = str(tmp_path / "ragged.fa")
    good = str(tmp_path / "good.fa")
    synth.write_fasta(fa, [("c1", seq)], 60)
    synth.write_fasta(good, [("c1", seq)], 60)
    text = open(fa).read().split("\n")
    text[5] = text[5] + text[6][:7]              # one long line, the next one short
    text[6] = text[6][7:]
    open(fa, "w").write("\n".join(text))
    with pytest.raises(hip.CoraHipError) as e:
        _build(fa, 50, 2)
    assert e.value.code == 51
    exe = os.path.join(hip.BIN_DIR, "homTable_setup")
    ex, inx, r = helpers.run_tool(exe, fa, 50, 2, helpers.split_path("phys", 1), helpers.split_path("par", 1), str(tmp_path / "a"))
    assert r.returncode == 0, r.stdout
    ex2, inx2, r2 = helpers.run_tool(exe, good, 50, 2, helpers.split_path("phys", 1), helpers.split_path("par", 1), str(tmp_path / "b"))
    assert r2.returncode == 0, r2.stdout
    assert helpers.read(ex) == helpers.read(ex2) and helpers.read(inx) == helpers.read(inx2) and len(helpers.read(ex)) > 0


@pytest.mark.parametrize("sharded", [False, True])
@pytest.mark.parametrize("kind", ["all_n", "short", "one_window_at_end"])
def test_contig_without_a_window_is_refused(kind, sharded, tmp_path):
    """HTS:2530 reads before its buffer for such a contig; here: status 54, from both entry points."""
    rs = np.random.RandomState(2)
    good = synth.random_seq(rs, 4000)
    bad = {"all_n": synth.seq_from_str("N" * 500),
           "short": synth.random_seq(rs, 40),
           "one_window_at_end": np.concatenate([synth.seq_from_str("N" * 100), synth.random_seq(rs, 50)])}[kind]   # Q4: no window
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, [("ok", good), ("bad", bad), ("ok2", synth.random_seq(rs, 2000))], 60)
    with pytest.raises(hip.CoraHipError) as e:
        _build(fa, 50, 2, sharded=sharded)
    assert e.value.code == 54


def test_many_contigs_need_the_wide_layout(tmp_path):
    rs = np.random.RandomState(3)
    unit = synth.random_seq(rs, 220)
    contigs = []
    for c in range(140):                                     # > 127 contigs: 1-byte contig ids do not suffice (cora.cpp:488-493)
        s = synth.random_seq(rs, 400)
        if c % 7 == 0:
            s[50:270] = synth.mutate(rs, unit, 0.01)         # homologies across contigs
        contigs.append(("c%d" % c, s))
    fa = str(tmp_path / "many.fa")
    synth.write_fasta(fa, contigs, 50)
    with pytest.raises(hip.CoraHipError) as e:
        _build(fa, 40, 2)
    assert e.value.code == 19
    oex, oinx = helpers.run_oracle(fa, 40, 2, helpers.split_path("phys", 2), helpers.split_path("par", 3), str(tmp_path / "o"), L=True)
    for sharded in (False, True):
        ex, en, inx, inn = _build(fa, 40, 2, 2, 3, L=True, sharded=sharded)
        assert ex == helpers.read(oex) and inx == helpers.read(oinx)
        assert en > 0 and inn > 0


@pytest.mark.parametrize("sharded", [False, True])
def test_reference_without_any_repeat_gives_empty_tables(sharded, tmp_path):
    fa = str(tmp_path / "u.fa")
    synth.write_fasta(fa, [("u", synth.random_seq(np.random.RandomState(4), 20000))], 60)
    oex, oinx = helpers.run_oracle(fa, 60, 1, helpers.split_path("phys", 1), helpers.split_path("par", 1), str(tmp_path / "o"))
    ex, en, inx, inn = _build(fa, 60, 1, sharded=sharded)
    assert (ex, inx) == (helpers.read(oex), helpers.read(oinx))
    assert (en, inn, len(ex), len(inx)) == (0, 0, 0, 0)


@pytest.mark.parametrize("K", [33, 47, 50, 63, 64])
def test_window_bitmap_32_positions_at_once_equals_the_definition(K, tmp_path):
    """k_kmer_count derives the validity of 32 window starts per thread from the N/pad bitmap by shifts; window_ok is the
    per-position definition (HTS:2619-2636, minus HTS:1954-1981's final stretch of exactly K bases).  Contigs full of
    N runs, with N-free stretches of K-1, K, K+1 bases before N runs and before contig ends, short contigs, and starts
    at every word phase."""
    rs = np.random.RandomState(100 + K)
    contigs = []
    for c in range(12):
        parts = []
        for _ in range(int(rs.randint(3, 40))):
            kind = rs.randint(0, 4)
            n_free = int(K + rs.randint(-2, 3)) if kind == 0 else int(rs.randint(1, 4 * K))
            parts.append(synth.random_seq(rs, n_free))
            parts.append(synth.seq_from_str("N" * int(rs.randint(1, 70))))
        tail = [K - 1, K, K + 1, 2 * K, 5][c % 5]                    # the final stretch: the Q4 case and its neighbours
        parts.append(synth.random_seq(rs, tail))
        if c % 3 == 0:
            parts.insert(0, synth.random_seq(rs, K))                # a contig that starts with exactly K bases, then N
        contigs.append(("c%d" % c, np.concatenate(parts)))
    contigs.append(("exact", synth.random_seq(rs, K + 1)))          # one window, ending one base before the end
    fa = str(tmp_path / "w.fa")
    synth.write_fasta(fa, contigs, 61)
    job = hip.Job(K, 1, helpers.split_path("phys", 1), helpers.split_path("par", 1))
    try:
        job.set_reference_file(fa)
        fast, slow = job.window_bits()
    finally:
        job.close()
    assert fast.size > 100 and slow.any()
    assert np.array_equal(fast, slow)


@pytest.mark.parametrize("K,d", [(36, 3), (50, 2)])
def test_low_complexity_and_iupac(K, d, tmp_path):
    """Tandem repeats make classes of thousands of entries, lists far beyond 16 items and runs full of tied targets
    (the host-finished part of <Inexact>); IUPAC letters and lower case break windows exactly as N / upper case do."""
    rs = np.random.RandomState(5)
    parts = [synth.random_seq(rs, 3000), synth.seq_from_str("AT" * 700), synth.random_seq(rs, 500), synth.seq_from_str("A" * 900),
             synth.random_seq(rs, 800), synth.seq_from_str("ACG" * 400), synth.seq_from_str("RYKM"), synth.random_seq(rs, 2500),
             synth.seq_from_str("AT" * 300), synth.seq_from_str("n"), synth.random_seq(rs, 1500)]
    seq = np.concatenate(parts)
    low = seq.copy()
    low[1000:1400] = np.frombuffer(bytes(low[1000:1400]).lower(), dtype=np.uint8)
    fa = str(tmp_path / "lc.fa")
    synth.write_fasta(fa, [("lc", low), ("rc", synth.revcomp(seq[2000:6200]))], 80)     # ACGT only up to the IUPAC letters
    oex, oinx = helpers.run_oracle(fa, K, d, helpers.split_path("phys", 4), helpers.split_path("par", 5), str(tmp_path / "o"))
    for sharded in (False, True):
        ex, en, inx, inn = _build(fa, K, d, 4, 5, sharded=sharded)
        assert ex == helpers.read(oex), "Exact differs (sharded=%s)" % sharded
        assert inx == helpers.read(oinx), "Inexact differs (sharded=%s)" % sharded


def test_long_classes_that_merge_into_blocks(tmp_path):
    """300 identical copies of a 400-base unit (every 3rd reverse-complemented): classes of 300 entries, consecutive
    ones merging into blocks -- the entry lists, the merge test and the record bytes of such classes are produced by
    whole workgroups instead of single threads (COOP_MIN in kernels.hip)."""
    contig = synth.genome_with_families(9, 400_000, [(400, 300, 0.0, 3), (130, 200, 0.0, 0)], name="dup")
    fa = str(tmp_path / "dup.fa")
    synth.write_fasta(fa, contig, 60)
    oex, oinx = helpers.run_oracle(fa, 50, 2, helpers.split_path("phys", 2), helpers.split_path("par", 3), str(tmp_path / "o"))
    assert os.path.getsize(oex) > 300 * 5 * 2
    for sharded in (False, True):
        ex, en, inx, inn = _build(fa, 50, 2, 2, 3, sharded=sharded)
        assert ex == helpers.read(oex), "Exact differs (sharded=%s)" % sharded
        assert inx == helpers.read(oinx), "Inexact differs (sharded=%s)" % sharded


def _tandem_reference(path):
    rs = np.random.RandomState(12)
    seq = np.concatenate([synth.random_seq(rs, 30000), synth.seq_from_str("AT" * 70000), synth.random_seq(rs, 20000),
                          synth.seq_from_str("A" * 70000), synth.random_seq(rs, 10000)])
    synth.write_fasta(path, [("tandem", seq)], 60)


def test_huge_classes_spread_over_the_grid(tmp_path):
    """140 kbp of (AT)n and 70 kbp of (A)n: two palindromic classes of ~140 000 entries and one of ~70 000, above
    HUGE_MIN (kernels.hip), so their entry lists and record bytes are written by the follow-up grid-wide kernels."""
    fa = str(tmp_path / "tandem.fa")
    _tandem_reference(fa)
    oex, oinx = helpers.run_oracle(fa, 50, 2, helpers.split_path("phys", 1), helpers.split_path("par", 1), str(tmp_path / "o"))
    assert os.path.getsize(oex) > 2 * 140000 * 5
    for sharded in (False, True):
        ex, en, inx, inn = _build(fa, 50, 2, 1, 1, sharded=sharded)
        assert ex == helpers.read(oex), "Exact differs (sharded=%s)" % sharded
        assert inx == helpers.read(oinx), "Inexact differs (sharded=%s)" % sharded


def test_long_equal_hash_run_is_repaired_in_place(tmp_path, monkeypatch):
    """With a 16-bit hash every one of the 65 536 hash values is shared by about two of the 60 000 distinct K-mers of
    this reference, so the three classes of 70 000+ members collide with others for certain: equal-hash runs far
    above 64 records with a handful of distinct keys, rebuilt by k_mixed_fix_long without the full-key sort."""
    fa = str(tmp_path / "tandem.fa")
    _tandem_reference(fa)
    oex, oinx = helpers.run_oracle(fa, 50, 2, helpers.split_path("phys", 1), helpers.split_path("par", 1), str(tmp_path / "o"))
    monkeypatch.setenv("CORA_HASH_BITS", "16")
    job = hip.Job(50, 2, helpers.split_path("phys", 1), helpers.split_path("par", 1))
    job.set_reference_file(fa)
    job.run()
    ex, en, inx, inn = job.tables()
    st = job.stats()
    job.close()
    assert ex == helpers.read(oex) and inx == helpers.read(oinx)
    assert st["hash_mixed_runs"] > 1000 and st["full_key_fallback"] == 0


def test_sharded_stages_at_full_size_match_reference_hash():
    """The multi-GPU stage functions on bench.py's workload (one rank): same sha256 as the real reference's tables."""
    import hashlib
    c = helpers.MANIFEST["cases"]["chr20_like_k50_d2"]
    contigs = synth.workload_chr20_like(20, 64_000_000)
    image, W, geom = synth.fasta_image(contigs, 60)
    import torch
    from cora_amd import dist as cdist
    job = hip.Job(c["K"], c["d"], helpers.split_path("phys", c["p"]), helpers.split_path("par", c["t"]))
    job.set_reference(image, W, geom)
    cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0))).run()
    e, en, i, inn = job.tables()
    assert (en, inn) == (c["exact_nodes"], c["inexact_nodes"])
    assert hashlib.sha256(e).hexdigest() == c["exact_sha256"] and hashlib.sha256(i).hexdigest() == c["inexact_sha256"]
    job.close()


def test_large_reference_is_semantically_consistent(tmp_path):
    """A 170 Mbp, three-contig reference: more than 2**27 K-mers, so the exact stage sorts on the 56-bit hash prefix
    (7 passes), and far beyond what the oracle can restate in test time.  Every record of both tables is checked
    against the bases by oracle/table_check (itself pinned on the real reference's tables, tests/test_table_check.py);
    plain run, sharded stages and a second plain run must give the same bytes."""
    import hashlib
    import subprocess
    check = os.path.join(helpers.ORACLE_DIR, "table_check")
    if not os.path.exists(check):
        subprocess.check_call(["make", "-C", helpers.ORACLE_DIR, "table_check"], stdout=subprocess.DEVNULL)
    contigs = [("big%d" % c, synth.workload_chr20_like(70 + c, n)[0][1]) for c, n in enumerate((90_000_000, 50_000_000, 30_000_000))]
    fa = str(tmp_path / "big.fa")
    synth.write_fasta(fa, contigs, 60)
    del contigs
    job = hip.Job(50, 2, helpers.split_path("phys", 10), helpers.split_path("par", 8))
    job.set_reference_file(fa)
    job.run()
    ex, en, inx, inn = job.tables()
    st = job.stats()
    assert st["kmers"] > (1 << 27) and st["full_key_fallback"] == 0
    ep, ip = str(tmp_path / "Exact"), str(tmp_path / "Inexact")
    open(ep, "wb").write(ex); open(ip, "wb").write(inx)
    open(ep + ".itemCount", "w").write("%d\n" % en); open(ip + ".itemCount", "w").write("%d\n" % inn)
    r = subprocess.run([check, fa, "50", "2", ep, ip], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert en > 50_000 and inn > 500_000, (en, inn, r.stdout)
    sha = (hashlib.sha256(ex).hexdigest(), hashlib.sha256(inx).hexdigest())
    del ex, inx
    import torch
    from cora_amd import dist as cdist
    cdist.ShardedJob(job, cdist.Comm(torch.device("cuda", 0))).run()
    e2, en2, i2, inn2 = job.tables()
    assert (hashlib.sha256(e2).hexdigest(), hashlib.sha256(i2).hexdigest(), en2, inn2) == sha + (en, inn)
    job.run()
    e3, _, i3, _ = job.tables()
    assert (hashlib.sha256(e3).hexdigest(), hashlib.sha256(i3).hexdigest()) == sha
    job.close()
