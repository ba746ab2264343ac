"""BASELINE.json's large configurations on the GPU: configs[2] (GRCh38-sized, d = 2), configs[3] (-H 3 -K 64, -t 24) and
configs[4] (repeat-rich).  No oracle reaches these sizes, so every table is checked record by record against the bases by
oracle/table_check (pinned on the real reference's own tables by tests/test_table_check.py), builds are repeated for
identical bytes, and each configuration has a small sibling that IS compared byte for byte: with the real reference
(tests/golden: syn_k64_d3_p10t24, grch38_like_s002_*) or with the oracle (the repeat-rich generator at 3 Mbp)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import helpers
from cora_amd import hip, synth

pytestmark = pytest.mark.gpu
TABLE_CHECK = os.path.join(helpers.ORACLE_DIR, "table_check")


def _fai(path, names, geom, width=60):
    with open(path + ".fai", "w") as f:
        for name, (off, n) in zip(names, geom):
            f.write("%s\t%d\t%d\t%d\t%d\n" % (name, n, off, width, width + 1))


def _build_and_check(contigs, K, d, p, t, tmp_path, repeat=True, L=False):
    names = [n for n, _ in contigs]
    image, W, geom = synth.fasta_image(contigs, 60)
    del contigs
    job = hip.Job(K, d, helpers.split_path("phys", p), helpers.split_path("par", t), L)
    job.set_reference(image, W, geom)
    job.run()
    e, en, x, xn = job.tables()
    st = job.stats()
    st["exact_sha256"], st["inexact_sha256"] = hashlib.sha256(e).hexdigest(), hashlib.sha256(x).hexdigest()
    if repeat:                                           # a second build of the same job: the same bytes
        job.run()
        e2, en2, x2, xn2 = job.tables()
        assert hashlib.sha256(e2).digest() == hashlib.sha256(e).digest() and hashlib.sha256(x2).digest() == hashlib.sha256(x).digest()
        del e2, x2
    job.close()
    fa = str(tmp_path / "ref.fa")
    with open(fa, "wb") as f:
        f.write(memoryview(image))
    _fai(fa, names, geom)
    ep, xp = str(tmp_path / "Exact"), str(tmp_path / "Inexact")
    for path, b, n in ((ep, e, en), (xp, x, xn)):
        with open(path, "wb") as f:
            f.write(b)
        with open(path + ".itemCount", "w") as f:
            f.write("%d\n" % n)
    del e, x, image
    r = subprocess.run([TABLE_CHECK, fa, str(K), str(d), ep, xp] + (["L"] if L else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for path in (fa, ep, xp):
        os.remove(path)
    assert r.returncode == 0, r.stdout[-2000:]
    return st, r.stdout


# the tables of workload_grch38_like(38, 1.0), K = 50, d = 2, -p 10 -t 8: what every build of it must produce (bench.py prints the
# same two hashes in `tables_of_last_step`); test_config2_grch38_sized_reference checks these very bytes with table_check
CONFIG2_SHA256 = ("2abe201b61f505d066eb1f6416c168e8a59d3c01d3c95c610f0020ba5c08e8cd", "3f366fdf22b788996899ddc09a15d42d2d78d63352be72e150d425d227c42be4")


def test_config2_grch38_sized_reference(tmp_path):
    """3.09 Gbp in 25 contigs (bench.py's workload): the exact sort on 40 hash bits with 2e6 collision places to repair, the sketches
    switched on by the size itself (not by CORA_SKETCH), per-seed record generation, the balanced candidate kernel."""
    assert "CORA_SKETCH" not in os.environ and "CORA_HASH_BITS" not in os.environ
    hip.init(0)
    st, out = _build_and_check(synth.workload_grch38_like(38, 1.0), 50, 2, 10, 8, tmp_path)
    assert st["kmers"] > 2_900_000_000 and st["representatives"] > (1 << 30)
    assert st["seed_sketch"] == 1 and st["full_key_fallback"] == 0 and st["hash_mixed_runs"] > 0
    assert st["verify_slots"] < st["candidate_pairs"] // 10          # the sketches dismissed the chance partners
    assert st["hom_pairs"] > 100_000_000 and st["inexact_nodes"] > 10_000_000
    assert (st["exact_sha256"], st["inexact_sha256"]) == CONFIG2_SHA256


@pytest.mark.parametrize("keep_own", ["1", "0"])
def test_config2_two_ranks_sharing_the_gpu_build_the_same_tables(keep_own, monkeypatch):
    """BASELINE configs[2] is quoted on 8 GPUs; a one-GPU box can only rehearse it: the in-process multi-GPU flow
    (cora_hip_job_run_multi) with two ranks on the one device, at the full 3.09 Gbp, must produce the bytes of the
    single-GPU build -- with every rank scanning all positions and keeping its own records (what 2 to 4 GPUs do) and with
    the records exchanged (what more GPUs do; here the "links" are device-local copies)."""
    hip.init(0)
    monkeypatch.setenv("CORA_SHARD_FILTER", keep_own)
    contigs = synth.workload_grch38_like(38, 1.0)
    image, W, geom = synth.fasta_image(contigs, 60)
    del contigs
    job = hip.Job(50, 2, helpers.split_path("phys", 10), helpers.split_path("par", 8))
    job.set_reference(image, W, geom)
    del image
    job.run_multi(2, [0, 0])
    e, en, x, xn = job.tables()
    st = job.stats()
    job.close()
    assert (hashlib.sha256(e).hexdigest(), hashlib.sha256(x).hexdigest()) == CONFIG2_SHA256
    assert st["kmers"] > 2_900_000_000 and st["hom_pairs"] > 100_000_000


def test_config2_eight_ranks_sharing_the_gpu_build_the_same_tables():
    """configs[2] as it is quoted: EIGHT ranks, records exchanged by all-to-all (what more than four GPUs do), at the full
    3.09 Gbp -- all eight on the one device of this box, every rank with its own arena cut from an eighth of the memory.  The other way
    of delivery (every rank keeping its own records out of a scan of all positions) is what 2 to 4 GPUs run and is covered above at two ranks: eight
    virtual ranks would hold eight copies of all 35 GB of K-mer records on the one GPU at once."""
    hip.init(0)
    contigs = synth.workload_grch38_like(38, 1.0)
    image, W, geom = synth.fasta_image(contigs, 60)
    del contigs
    job = hip.Job(50, 2, helpers.split_path("phys", 10), helpers.split_path("par", 8))
    job.set_reference(image, W, geom)
    del image
    job.run_multi(8, [0] * 8)
    e, en, x, xn = job.tables()
    st = job.stats()
    job.close()
    assert (hashlib.sha256(e).hexdigest(), hashlib.sha256(x).hexdigest()) == CONFIG2_SHA256
    assert st["kmers"] > 2_900_000_000 and st["hom_pairs"] > 100_000_000 and st["seed_sketch"] == 1


DEVICE_MEMORY = 288 * (1 << 30)


def test_config3_k64_d3_grch38_sized(tmp_path):
    """BASELINE configs[3] at its stated size: -H 3 -K 64 with -t 24 on the 3.09 Gbp GRCh38-sized reference.  d = 3 compacts contigs
    above 50 Mbp window by window (HTS:2852-2856, finish.cpp plan_windows) -- 22 such contigs here; four seeds, sketches of twelve
    bytes (20-byte records through the seed sorts).  The most the stage arrays hold at once must stay below the device's memory."""
    hip.init(0)
    contigs = synth.workload_grch38_like(38, 1.0)
    assert sum(len(s) > 50_000_000 for _, s in contigs) >= 20
    st, out = _build_and_check(contigs, 64, 3, 10, 24, tmp_path, repeat=False)
    assert st["kmers"] > 2_900_000_000 and st["seed_sketch"] == 1 and st["hom_pairs"] > 100_000_000 and st["full_key_fallback"] == 0
    assert 0 < st["pool_peak_bytes"] < DEVICE_MEMORY - (24 << 30), st["pool_peak_bytes"]


def _repeat_rich(seed, total, fams, tandem):
    contigs = synth.genome_with_families(seed, total, fams, name="rr")
    g = contigs[0][1]
    a0, a1 = total // 3, 2 * total // 3
    g[a0:a0 + tandem] = np.resize(np.frombuffer(b"AT", dtype=np.uint8), tandem)
    g[a1:a1 + tandem] = ord("A")
    return contigs


def test_config4_repeat_rich_100m(tmp_path):
    """Families of 10^4 and 3 x 10^4 copies (at 10 % and 15 % divergence, which keeps the pair output bounded: pairs grow
    with copies^2 x P(<= 2 mismatches in 50)), hundreds of copies at 1-3 %, and 1 Mbp each of (AT)n and (A)n: equal-code
    runs of thousands of records for the candidate walk, exact classes of 10^6 entries."""
    hip.init(0)
    fams = [(300, 10_000, 0.10, 3), (150, 30_000, 0.15, 3), (600, 300, 0.03, 3), (300, 900, 0.02, 2), (1200, 40, 0.01, 2)]
    st, out = _build_and_check(_repeat_rich(5, 100_000_000, fams, 1_000_000), 50, 2, 10, 8, tmp_path)
    assert st["candidate_pairs"] > 1_000_000_000 and st["hom_pairs"] > 10_000_000 and st["big_nodes"] > 1000


def test_config4_repeat_rich_gbp(tmp_path):
    """BASELINE configs[4] at its stated size: 1 Gbp, families of 3 x 10^5, 10^5 and 10^4 copies (each at a divergence that keeps its
    REPORTED pairs within the budget synth.family_budget states -- 2.4e8 pairs in all, 4.8e8 directed items < 2^32 -- while
    their equal-code runs reach 8 000 records and the comparisons 1.2e10 per seed), thousands of copies at 1-3 %, and 10 Mbp
    each of (AT)n and (A)n: exact classes of 5 x 10^6 (palindromic, listed twice) and 10^7 entries."""
    hip.init(0)
    pairs = sum(synth.family_budget(u, c, d)[0] for u, c, d in synth.REPEAT_RICH_GBP_FAMILIES)
    assert 1e8 < pairs < (1 << 31) and max(c for _, c, _ in synth.REPEAT_RICH_GBP_FAMILIES) >= 100_000
    st, out = _build_and_check(synth.workload_repeat_rich_gbp(5, 1_000_000_000), 50, 2, 10, 8, tmp_path, repeat=False)
    assert st["kmers"] > 950_000_000 and st["candidate_pairs"] > 10_000_000_000 and st["big_nodes"] > 100_000
    # the budget is an upper bound: it counts window pairs of all copies, the path pairs REPRESENTATIVES (at 1-3 % divergence a fifth of
    # a family's windows are exact duplicates of the unit's and collapse into one class before any pair is looked at)
    assert 10_000_000 < st["hom_pairs"] < pairs, (st["hom_pairs"], pairs)
    assert st["exact_classes"] > 0 and 0 < st["pool_peak_bytes"] < DEVICE_MEMORY - (24 << 30)


@pytest.mark.parametrize("sketch", ["0", "1"])
def test_config4_repeat_rich_small_equals_oracle(sketch, tmp_path, monkeypatch):
    """The same recipe at a size the oracle can do: byte for byte, with and without sketches (runs of 1 500 records are
    longer than the candidate kernel's LDS window, so the walk continues in global memory)."""
    monkeypatch.setenv("CORA_SKETCH", sketch)
    hip.init(0)
    fams = [(200, 1500, 0.10, 3), (120, 2500, 0.15, 3), (300, 60, 0.03, 3), (150, 90, 0.02, 2)]
    contigs = _repeat_rich(6, 3_000_000, fams, 20_000)
    fa = str(tmp_path / "rr.fa")
    synth.write_fasta(fa, contigs, 60)
    phys, par = helpers.split_path("phys", 10), helpers.split_path("par", 8)
    oex, oinx = helpers.run_oracle(fa, 50, 2, phys, par, str(tmp_path / "oracle"))
    e, en, i, inn, st = hip.build_tables(fa, 50, 2, phys, par)
    assert e == helpers.read(oex) and i == helpers.read(oinx)
    assert st["seed_sketch"] == int(sketch)


@pytest.mark.parametrize("name", ["grch38_like_s002_k50_d2", "grch38_like_s002_k64_d3"])
def test_bench_generator_scaled_down_matches_reference_hash(name):
    """bench.py's generator at 1/500 of its size (25 contigs, N runs, soft-masked capped families): the real reference's
    tables for it are pinned by sha256 (tests/golden/make_golden.py; cora's defaults -p 10 -t 8, and -t 24 for d = 3)."""
    hip.init(0)
    c = helpers.MANIFEST["cases"][name]
    e, en, i, inn, st = hip.build_tables(helpers.input_path(c["input"]), c["K"], c["d"], helpers.split_path("phys", c["p"]),
                                         helpers.split_path("par", c["t"]))
    assert (en, inn) == (c["exact_nodes"], c["inexact_nodes"])
    assert hashlib.sha256(e).hexdigest() == c["exact_sha256"] and hashlib.sha256(i).hexdigest() == c["inexact_sha256"]
