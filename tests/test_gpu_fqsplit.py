"""GPU: the fastqSplit path (cora_hip_fqsplit_*, through the C-ABI and through the drop-in executable) against the fragment
files the real reference wrote (tests/golden/fqsplit_manifest.json) and against the oracle on seeded reads."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_fqsplit_oracle import build_fqsplit_oracle, check_fragments, mfg
from cora_amd import fastqsplit, hip, synth

pytestmark = pytest.mark.gpu

FQSPLIT_EXE = os.path.join(helpers.ROOT, "cora_amd", "bin", "fastqSplit")


@pytest.fixture(scope="module", autouse=True)
def _init():
    hip.init(0)


@pytest.mark.parametrize("name", sorted(mfg.CASES))
def test_executable_matches_the_reference(name, tmp_path):
    outs, r = mfg.run(FQSPLIT_EXE, name, str(tmp_path))
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-300:])
    check_fragments(name, outs)


@pytest.mark.parametrize("name", ["paired_4_g4", "threeway_paired_2_g10", "single_4_g8_two_samples"])
def test_cabi_matches_the_reference(name):
    sets, mode, n, size = mfg.CASES[name]
    L = mfg.mcg.READ_SETS[sets[0]][1]
    fs = fastqsplit.FastqSplit(mfg.frag_text(n, size).encode(), L, mode)
    for s in sets:
        fs.add_reads(open(os.path.join(mfg.INP, s + "_1.fq"), "rb").read())
        if "PAIRED" in mode:
            fs.add_reads(open(os.path.join(mfg.INP, s + "_2.fq"), "rb").read())
    fs.run()
    check_fragments(name, fs.groups())
    fs.run()                                                   # a second run of the same object: the same bytes
    check_fragments(name, fs.groups())
    fs.close()


@pytest.mark.parametrize("L,mode,n,size,reads", [(100, "SINGLE", 8, 4, 300_000), (150, "SPLIT_PAIRED", 5, 4, 100_000),
                                                 (255, "THREEWAY_SINGLE", 10, 2, 60_000), (36, "THREEWAY_PAIRED", 3, 4, 200_000),
                                                 (50, "SINGLE", 1, 4, 100_000)])
def test_seeded_reads_against_the_oracle(L, mode, n, size, reads, tmp_path):
    rs = np.random.RandomState(L + n)
    ref = synth.random_seq(rs, 500_000)
    paired = "PAIRED" in mode
    imgs = [synth.fastq_image(1000 + L + m, ref, reads, L, n_frac=0.05) for m in range(2 if paired else 1)]
    paths = []
    for m, img in enumerate(imgs):
        p = str(tmp_path / ("r%d.fq" % m))
        open(p, "wb").write(img)
        paths.append(p)
    sig, lst, prefix = str(tmp_path / "sig"), str(tmp_path / "list"), str(tmp_path / "o")
    open(sig, "w").write(mfg.frag_text(n, size))
    open(lst, "w").write("1\n%s\t%d\n" % ("\t".join(paths), reads))
    r = subprocess.run([build_fqsplit_oracle(), sig, str(L), lst, prefix, mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-300:]
    fs = fastqsplit.FastqSplit(open(sig, "rb").read(), L, mode)
    for img in imgs:
        fs.add_reads(img)
    fs.run()
    st = fs.stats()
    assert st["read_mers"] == reads * (2 if paired else 1) * fastqsplit.MODES[mode][0]
    for g in range(n):
        assert fs.group(g) == open("%s.frag%d" % (prefix, g), "rb").read(), "fragment %d differs" % g
    fs.close()


def test_refusals(tmp_path):
    sig = mfg.frag_text(3, 4).encode()
    ok = b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n"
    fs = fastqsplit.FastqSplit(sig, 10, "SINGLE")
    fs.add_reads(ok + b"@r2\nACGTACG\n+\nIIIIIII\n")          # a sequence line shorter than the read length
    with pytest.raises(hip.CoraHipError) as e:
        fs.run()
    assert e.value.code == 78
    fs.close()
    fs = fastqsplit.FastqSplit(sig, 10, "SINGLE")
    fs.add_reads(ok + b"@r2\nacGTACGTAC\n+\nIIIIIIIIII\n")    # lower case in the signal: outside the reference's tables
    with pytest.raises(hip.CoraHipError) as e:
        fs.run()
    assert e.value.code == 78
    fs.close()
    fs = fastqsplit.FastqSplit(sig, 10, "SINGLE")
    fs.add_reads(ok + b"@r2\nACGTACGTAC\n+\n")                # three lines in the last record
    with pytest.raises(hip.CoraHipError) as e:
        fs.run()
    assert e.value.code == 7
    fs.close()
    fs = fastqsplit.FastqSplit(sig, 10, "SINGLE")              # an empty file: empty fragment files
    fs.add_reads(b"")
    fs.run()
    assert fs.groups() == [b"", b"", b""]
    fs.close()
    r = subprocess.run([FQSPLIT_EXE, "a", "b"], stdout=subprocess.PIPE)
    assert r.returncode == 1
