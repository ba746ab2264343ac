"""GPU: the FASTA scan (cora_hip_fasta_scan) writes the bytes `cora faiGenerate` writes, and the table build through
cora_hip_job_set_reference_fasta gives the same tables as the .fai + fixed-line-width path -- also for files the
reference's loader cannot read (ragged lines, blank lines, trailing words), where the expected tables are those of the
same sequences in 60-column layout."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_fai import fai_cases, build_fai_oracle
from cora_amd import hip, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    hip.init(0)


@pytest.mark.parametrize("fa,fai", fai_cases(), ids=lambda p: os.path.basename(p))
def test_fai_text_equals_the_real_command(fa, fai):
    f = hip.Fasta(open(fa, "rb").read())
    try:
        assert f.fai_text() == open(fai, "rb").read()
    finally:
        f.close()


def _fuzz_fasta(rs, n_bytes):
    """Random FASTA-like text: every kind of line the scan distinguishes, in random order."""
    parts, size = [], 0
    while size < n_bytes:
        kind = rs.randint(0, 12)
        n = int(rs.choice([0, 1, 7, 60, 60, 60, 61, 200, 5000]))
        body = "".join("ACGTNacgtn"[i] for i in rs.randint(0, 10, n))
        if kind == 0:
            line = ">" + "ctg%d" % rs.randint(0, 1 << 30) + ["", " descr", "\tdescr x", " "][rs.randint(0, 4)]
        elif kind == 1:
            line = [" ", "\t", "  \t "][rs.randint(0, 3)] + body
        elif kind == 2:
            line = body + [" x", "\tzz top", " "][rs.randint(0, 3)]
        elif kind == 3:
            line = ""
        elif kind == 4:
            line = " >hdr_after_blank"
        else:
            line = body
        parts.append(line + "\n")
        size += len(line) + 1
    text = "".join(parts)
    if rs.randint(0, 2):
        text = text.rstrip("\n")
    return text.encode()


@pytest.mark.parametrize("seed,size", [(1, 3000), (2, 20000), (3, 300000), (4, 300000), (5, 5_000_000)])
def test_fai_text_on_random_files_equals_the_oracle(seed, size, tmp_path):
    rs = np.random.RandomState(seed)
    image = _fuzz_fasta(rs, size)
    p = str(tmp_path / "f.fa")
    open(p, "wb").write(image)
    want = subprocess.run([build_fai_oracle(), p], stdout=subprocess.PIPE, check=True).stdout
    f = hip.Fasta(image)
    try:
        assert f.fai_text() == want
        assert f.info()["conforming"] == 0
    finally:
        f.close()


def test_conforming_is_recognised():
    for name in ("syn", "multi", "q4a"):
        f = hip.Fasta(open(helpers.input_path(name), "rb").read())
        info = f.info()
        f.close()
        assert info["conforming"] == 1 and info["line_width"] == 60, (name, info)
    for case in ("ragged", "blank_lines", "leading_blanks", "text_before_header", "consecutive_headers", "no_header", "empty"):
        f = hip.Fasta(open(os.path.join(helpers.GOLDEN, "fai_cases", case + ".fa"), "rb").read())
        info = f.info()
        f.close()
        assert info["conforming"] == 0, (case, info)


def _same(e, en, i, inn, case, want, label):
    c = helpers.MANIFEST["cases"][case]
    assert en == c["exact_nodes"] and inn == c["inexact_nodes"], (label, en, inn)
    assert e == helpers.read(os.path.join(want, "Exact")), "%s: Exact differs" % label
    assert i == helpers.read(os.path.join(want, "Inexact")), "%s: Inexact differs" % label


def _tables_via_fasta(image, K, d, p, t, L):
    job = hip.Job(K, d, helpers.split_path("phys", p), helpers.split_path("par", t), L)
    try:
        job.set_reference_fasta(image)
        job.run()
        e, en, i, inn = job.tables()
        return e, en, i, inn, job.fasta_info
    finally:
        job.close()


GOLDEN_CASES = ["syn_k50_d2", "multi_k50_d2_L", "pal_k50_d2", "q4a_k50_d2", "q4b_k50_d2", "lowcx_k36_d1", "fam_k48_d3", "q_k50_d2"]


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_conforming_file_without_fai_gives_the_golden_tables(case):
    meta, want = helpers.golden_case(case)
    image = open(helpers.input_path(meta["input"]), "rb").read()
    e, en, i, inn, info = _tables_via_fasta(image, meta["K"], meta["d"], meta["p"], meta["t"], meta["L"])
    assert info["conforming"] == 1
    _same(e, en, i, inn, case, want, case)


def _reshape(image: bytes, rs, style):
    """The same contigs and bases in a layout ReadReference cannot read."""
    out = []
    for rec in image.decode().split(">")[1:]:
        head, _, body = rec.partition("\n")
        seq = body.replace("\n", "")
        out.append(">" + head + (" length=%d" % len(seq) if style != "one_line" else "") + "\n")
        if style == "one_line":
            out.append(seq + "\n")
            continue
        at = 0
        while at < len(seq):
            n = int(rs.choice([1, 13, 60, 61, 70, 250]))
            chunk = seq[at:at + n]
            at += n
            pre = [" ", "\t", ""][rs.randint(0, 3)] if style == "messy" else ""
            post = [" trailing", "", "", "\t#"][rs.randint(0, 4)] if style == "messy" else ""
            out.append(pre + chunk + post + "\n")
            if style == "messy" and rs.randint(0, 5) == 0:
                out.append(["\n", "  \n"][rs.randint(0, 2)])
    return "".join(out).encode()


@pytest.mark.parametrize("style", ["ragged", "messy", "one_line"])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_any_layout_gives_the_golden_tables(case, style):
    meta, want = helpers.golden_case(case)
    image = open(helpers.input_path(meta["input"]), "rb").read()
    rs = np.random.RandomState(11)
    e, en, i, inn, info = _tables_via_fasta(_reshape(image, rs, style), meta["K"], meta["d"], meta["p"], meta["t"], meta["L"])
    if style != "one_line":            # a single line per contig can be a valid fixed-width layout
        assert info["conforming"] == 0
    _same(e, en, i, inn, case, want, case + "/" + style)


def test_executables(tmp_path):
    """cora_faiGenerate writes the command's .fai; homTable_setup then builds the golden tables from a ragged file."""
    meta, want = helpers.golden_case("syn_k50_d2")
    image = open(helpers.input_path(meta["input"]), "rb").read()
    fa = str(tmp_path / "r.fa")
    open(fa, "wb").write(_reshape(image, np.random.RandomState(3), "messy"))
    r = subprocess.run([os.path.join(hip.BIN_DIR, "cora_faiGenerate"), fa], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout
    oracle = subprocess.run([build_fai_oracle(), fa], stdout=subprocess.PIPE, check=True).stdout
    assert open(fa + ".fai", "rb").read() == oracle
    ex, inx, proc = helpers.run_tool(os.path.join(hip.BIN_DIR, "homTable_setup"), fa, meta["K"], meta["d"],
                                     helpers.split_path("phys", meta["p"]), helpers.split_path("par", meta["t"]), str(tmp_path / "o"))
    assert proc.returncode == 0, proc.stdout
    helpers.assert_same_tables(ex, inx, want, "exe/messy")
    # a .fai that describes another file is refused
    open(fa + ".fai", "w").write("x\t12345\t3\t60\t61\n")
    ex, inx, proc = helpers.run_tool(os.path.join(hip.BIN_DIR, "homTable_setup"), fa, meta["K"], meta["d"],
                                     helpers.split_path("phys", meta["p"]), helpers.split_path("par", meta["t"]), str(tmp_path / "o2"))
    assert proc.returncode == 51


def test_full_size_scan_equals_the_oracle(tmp_path):
    """The 64 Mbp bench reference: .fai bytes against the oracle, conforming layout recognised."""
    contigs = synth.workload_chr20_like()
    image = synth.fasta_image(contigs, 60)[0]
    image = bytes(image)
    p = str(tmp_path / "big.fa")
    open(p, "wb").write(image)
    want = subprocess.run([build_fai_oracle(), p], stdout=subprocess.PIPE, check=True).stdout
    f = hip.Fasta(image)
    try:
        assert f.fai_text() == want
        info = f.info()
        assert info["conforming"] == 1 and info["bases"] == 64_000_000
    finally:
        f.close()


def test_image_above_half_a_gigabyte_goes_up_through_the_staged_upload(tmp_path):
    """From 512 MB on the image is uploaded by six host threads through pinned buffers (fasta.hip: upload_pageable) instead of
    one hipMemcpy from pageable memory: the scan must see the same bytes (.fai against the oracle, base count), also for a
    size that is no multiple of the 8 MB chunks."""
    import mmap
    rs = np.random.RandomState(77)
    contigs = [("c%d" % i, synth.random_seq(rs, n)) for i, n in enumerate((190_000_001, 170_000_000, 151_234_567, 60_000_003, 977))]
    image = synth.fasta_image(contigs, 60)[0]
    total = sum(len(s) for _, s in contigs)
    del contigs
    p = str(tmp_path / "half_g.fa")
    open(p, "wb").write(bytes(image))
    del image
    want = subprocess.run([build_fai_oracle(), p], stdout=subprocess.PIPE, check=True).stdout
    with open(p, "rb") as fh:
        m = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        assert len(m) > 512 * 1024 * 1024 and len(m) % (8 << 20) != 0
        f = hip.Fasta(m)
        try:
            assert f.fai_text() == want
            info = f.info()
            assert info["conforming"] == 1 and info["bases"] == total
        finally:
            f.close()
            m.close()
