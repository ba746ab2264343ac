"""Deterministic synthetic reference genomes, FASTA/.fai writers and split files.

No real genome exists offline, so every workload and fixture is a seeded
synthetic genome.  Generation uses numpy's
frozen legacy ``RandomState`` so the bytes are stable across numpy versions.

The FASTA layout is the one the reference's loader assumes
(/root/reference/homTable_setup.cpp:96-202): a fixed line width, ``\\n`` line
ends; the ``.fai`` has the five columns ``cora faiGenerate`` writes
(/root/reference/cora.cpp:1023).
"""
from __future__ import annotations

import os
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def random_seq(rs: np.random.RandomState, n: int) -> np.ndarray:
    """n uniform random bases as uint8 ASCII."""
    return _ACGT[rs.randint(0, 4, size=n).astype(np.uint8)]


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(rs: np.random.RandomState, seq: np.ndarray, rate: float) -> np.ndarray:
    """Substitutions only (the tables model Hamming distance), each to a different base."""
    out = seq.copy()
    n = len(seq)
    hit = np.nonzero(rs.random_sample(n) < rate)[0]
    if len(hit):
        code = np.searchsorted(_ACGT, out[hit])  # A C G T are sorted in ASCII
        code = (code + rs.randint(1, 4, size=len(hit))) & 3
        out[hit] = _ACGT[code]
    return out


def seq_from_str(s: str) -> np.ndarray:
    return np.frombuffer(s.encode(), dtype=np.uint8).copy()


def write_fasta(path: str, contigs, width: int = 60) -> None:
    """contigs: list of (name, uint8 array).  Also writes path + '.fai'."""
    fai = []
    with open(path, "wb") as f:
        off = 0
        for name, seq in contigs:
            hdr = b">" + name.encode() + b"\n"
            f.write(hdr)
            off += len(hdr)
            n = len(seq)
            fai.append((name, n, off, width, width + 1))
            full = n // width
            if full:
                body = np.empty((full, width + 1), dtype=np.uint8)
                body[:, :width] = seq[: full * width].reshape(full, width)
                body[:, width] = 10
                f.write(body.tobytes())
            rest = n - full * width
            if rest:
                f.write(seq[full * width:].tobytes() + b"\n")
            off += n + full + (1 if rest else 0)
    with open(path + ".fai", "w") as f:
        for row in fai:
            f.write("%s\t%d\t%d\t%d\t%d\n" % row)


def fasta_image(contigs, width: int = 60):
    """In-memory FASTA file image: -> (uint8 array, line width, [(offset of first base, length)])."""
    parts, geom, off = [], [], 0
    for name, seq in contigs:
        hdr = np.frombuffer(b">" + name.encode() + b"\n", dtype=np.uint8)
        parts.append(hdr)
        off += len(hdr)
        n = len(seq)
        geom.append((off, n))
        full = n // width
        if full:
            body = np.empty((full, width + 1), dtype=np.uint8)
            body[:, :width] = seq[: full * width].reshape(full, width)
            body[:, width] = 10
            parts.append(body.reshape(-1))
        rest = n - full * width
        if rest:
            parts.append(seq[full * width:])
            parts.append(np.array([10], dtype=np.uint8))
        off += n + full + (1 if rest else 0)
    return np.concatenate(parts), width, geom


# --------------------------------------------------------------------------
# split files (what cora.cpp:295-314 hands to homTable_setup as argv[3]/argv[6])
# --------------------------------------------------------------------------
def _canonical_mers(k: int):
    out = []
    for i in range(4 ** k):
        s = "".join("ACGT"[(i >> (2 * (k - 1 - j))) & 3] for j in range(k))
        rc = "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(s))
        if s <= rc:
            out.append(s)
    return out


def write_round_robin_splits(path: str, k: int, n: int) -> None:
    """A valid split file with n groups: canonical k-mers dealt round-robin.

    The reference's tables are invariant to the partition,
    so any partition that covers every k-mer once is a legal argv[3] (k=2) or
    argv[6] (k=4).  cora itself uses its LPT_2.dat / LPT_4.dat lines."""
    mers = _canonical_mers(k)
    groups = [[] for _ in range(n)]
    for i, m in enumerate(mers):
        groups[i % n].append(m)
    with open(path, "w") as f:
        f.write("%d\n" % n)
        for g in groups:
            f.write(" ".join(g) + "\n")


# --------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------
def genome_with_families(seed: int, total_len: int, families, n_runs=(), lower=(), name="chrS"):
    """One contig of uniform random ACGT with repeat families injected.

    families: list of (unit_len, copies, divergence, rc_every) -- each copy is an
    independently mutated instance of a random unit, every ``rc_every``-th copy
    reverse-complemented (0 = never), dropped at random non-overlapping slots.
    n_runs: list of (start, length) stretches overwritten with N.
    lower: list of (start, length) stretches lower-cased."""
    rs = np.random.RandomState(seed)
    g = random_seq(rs, total_len)
    slot = max(u for u, _, _, _ in families) if families else 1
    nslots = total_len // slot
    need = sum(c for _, c, _, _ in families)
    assert need <= nslots, "families do not fit"
    slots = rs.permutation(nslots)[:need]
    si = 0
    for unit_len, copies, div, rc_every in families:
        unit = random_seq(rs, unit_len)
        for c in range(copies):
            inst = mutate(rs, unit, div)
            if rc_every and c % rc_every == rc_every - 1:
                inst = revcomp(inst)
            at = int(slots[si]) * slot
            si += 1
            g[at:at + unit_len] = inst
    for st, ln in n_runs:
        g[st:st + ln] = ord("N")
    for st, ln in lower:
        seg = g[st:st + ln]
        g[st:st + ln] = np.where(seg != ord("N"), seg | 0x20, seg)
    return [(name, g)]


def workload_ecoli_like(seed: int = 2024):
    """configs[0]: 4.6 Mbp, 5 families x 12 copies x 1.2 kbp at 1 % divergence."""
    return genome_with_families(seed, 4_600_000, [(1200, 12, 0.01, 3)] * 5, name="ecoli_like")


def workload_chr20_like(seed: int = 20, total_len: int = 64_000_000):
    """configs[1]: chr20-sized contig, 0.5 Mbp leading N-run, repeat families
    totalling about 5 % of the length at 2 % divergence."""
    scale = total_len / 64_000_000
    fams = []
    rs = np.random.RandomState(seed + 1)
    budget = int(0.05 * total_len)
    while budget > 0:
        unit = int(rs.choice([300, 600, 1200, 2400, 6000]))
        copies = int(rs.randint(4, 21))
        if unit * copies > budget:
            copies = max(2, budget // unit)
        fams.append((unit, copies, 0.02, 3))
        budget -= unit * copies
    nlead = int(500_000 * scale)
    return genome_with_families(seed, total_len, fams, n_runs=[(0, nlead)], name="chr20_like")


def workload_repeat_rich(seed: int = 5, total_len: int = 100_000_000, max_copies: int = 3000):
    """configs[4] in spirit, scaled to what ANY implementation of the reference's semantics can
    hold: random sequence + repeat families with copy numbers from 10^2 up to ``max_copies`` at 0.5-3 % divergence
    (pairs grow with copies^2 per position: 10^6 copies would mean 10^12 pair records per position, for the reference
    too) + two tandem low-complexity stretches, (AT)n and (A)n, of total_len/100 bases each: exact classes with
    ~10^6 entries, the (AT)n ones palindromic (listed twice, quirk Q2)."""
    rs = np.random.RandomState(seed + 7)
    fams = []
    copies = 100
    while copies <= max_copies:
        fams.append((int(rs.choice([150, 300, 600])), copies, float(rs.choice([0.005, 0.01, 0.02, 0.03])), 3))
        copies *= 3
    fams += [(1200, 40, 0.01, 2), (6000, 10, 0.005, 0)]
    contig = genome_with_families(seed, total_len, fams, name="repeat_rich")
    g = contig[0][1]
    run = total_len // 100
    a0, a1 = total_len // 3, 2 * total_len // 3
    g[a0:a0 + run] = np.resize(np.frombuffer(b"AT", dtype=np.uint8), run)
    g[a1:a1 + run] = ord("A")
    return contig


def family_budget(unit: int, copies: int, div: float, K: int = 50, d: int = 2):
    """What one repeat family costs ANY implementation of the reference's semantics, as (reported pairs, comparisons per seed):
    two of its copies differ per base with probability q = _pairwise_q(div); a window pair is reported when it has 1..d
    mismatches (Binomial(K, q)), and is compared under a seed when its 16 seed bases agree ((1-q)^16) -- for every one of
    the unit * copies^2 / 2 window pairs (homTable_setup.cpp:2753-2762 walks the whole chain of an equal code).  An UPPER bound:
    only class representatives are paired, and at low divergence many windows of a family are exact duplicates of each other
    (unit 150, 2 000 copies at 3 %: 1 007 distinct windows per position and 1.9e4 reported pairs among them, against 7.8e5 counted here)."""
    import math
    q = _pairwise_q(div)
    p_rep = sum(math.comb(K, k) * q ** k * (1 - q) ** (K - k) for k in range(1, d + 1))
    wp = unit * copies * (copies - 1) / 2.0
    return wp * p_rep, wp * (1 - q) ** 16


# BASELINE configs[4] at its stated size: copy numbers from 10^2 to 3 x 10^5.  The divergence of every family is
# chosen so that its REPORTED pairs stay within a budget (family_budget; the sum is 2.4e8 pairs = 4.8e8 directed items, below
# the 2^32 items the path's 32-bit item indices hold -- the reference's own pair files grow the same way): the more copies,
# the older the family, as in real genomes.  Equal-code runs among the representatives reach copies * (1-div)^16:
# 6 000 records for the 10^5-copy family, 8 000 for the 3 x 10^5-copy one; comparisons per seed 1.2e10 in all.
REPEAT_RICH_GBP_FAMILIES = [(150, 300_000, 0.20), (150, 100_000, 0.16), (300, 10_000, 0.10), (150, 2_000, 0.03), (300, 900, 0.02),
                            (600, 300, 0.03), (1200, 100, 0.01)]


def workload_repeat_rich_gbp(seed: int = 5, total_len: int = 1_000_000_000, families=None, tandem: int = 10_000_000):
    """configs[4]: one contig of total_len uniform random bases + the families above (every third copy reverse-complemented,
    instances at random non-overlapping slots) + ``tandem`` bases each of (AT)n and (A)n: exact classes of tandem / 2 and
    tandem entries, the (AT)n ones palindromic (quirk Q2).  Vectorised (10^5 copies per family)."""
    fams = REPEAT_RICH_GBP_FAMILIES if families is None else families
    rs = np.random.RandomState(seed)
    lut = np.zeros(256, dtype=np.uint32)
    for b in range(256):
        lut[b] = sum(int(_ACGT[(b >> (2 * k)) & 3]) << (8 * k) for k in range(4))
    g = lut[np.frombuffer(rs.bytes((total_len + 3) // 4), dtype=np.uint8)].view(np.uint8)[:total_len]
    slot = max(u for u, _, _ in fams)
    need = sum(c for _, c, _ in fams)
    # the two tandem stretches sit at 1/3 and 2/3 of the contig; slots overlapping them are not used
    a0, a1 = total_len // 3, 2 * total_len // 3
    ok = np.ones(total_len // slot, dtype=bool)
    for a in (a0, a1):
        ok[max(0, a // slot - 1):(a + tandem) // slot + 2] = False
    free = np.nonzero(ok)[0]
    assert need <= len(free), "families do not fit"
    slots = free[rs.permutation(len(free))[:need]]
    si = 0
    for unit, copies, div in fams:
        codes = rs.randint(0, 4, size=unit).astype(np.uint8)
        m = np.tile(codes, copies)
        nm = int(rs.binomial(unit * copies, div))
        idx = rs.randint(0, unit * copies, size=nm)
        m[idx] = (m[idx] + rs.randint(1, 4, size=nm).astype(np.uint8)) & 3
        rows = m.reshape(copies, unit)
        rows[2::3] = 3 - rows[2::3, ::-1]
        rows = _ACGT[rows]
        at = slots[si:si + copies].astype(np.int64) * slot
        si += copies
        g[(at[:, None] + np.arange(unit, dtype=np.int64)[None, :]).reshape(-1)] = rows.reshape(-1)
    g[a0:a0 + tandem] = np.resize(np.frombuffer(b"AT", dtype=np.uint8), tandem)
    g[a1:a1 + tandem] = ord("A")
    return [("repeat_rich", g)]


# GRCh38 primary assembly: chr1..22, X, Y, M (bases, incl. N)
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                  133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616,
                  64444167, 46709983, 50818468, 156040895, 57227415, 16569]
GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"]


def _pairwise_q(div: float) -> float:
    """Per-base mismatch probability between two copies that were each mutated at rate div from one unit."""
    return 2 * div * (1 - div) + div * div * 2 / 3


def plan_repeat_families(seed: int, young_bases: int, old_bases: int, cand_per_base: float = 3.6):
    """Repeat families of the GRCh38-like workload as (unit, copies, divergence) lists.

    young: 1-3 % divergence, 4-20 copies (segmental-duplication-like; nearly every window of a copy is within d=2 of its
    relatives, so these carry the pair output: about copies/2 pairs per base).
    old: 5-15 % divergence, high copy number (interspersed-repeat-like).  Their copy number is CAPPED: two copies share a
    16-base seed code with probability (1-q)^16, and every implementation of the reference's semantics -- the reference's
    chained hash lists (homTable_setup.cpp:2753-2762) included -- compares all members of an equal-code run pairwise, i.e.
    unit * copies^2 / 2 * (1-q)^16 comparisons per seed.  The cap keeps that at cand_per_base comparisons per repeat base
    (about 5e9 for the full-size reference, twice what chance collisions of 32-bit codes among 2.9e9 K-mers cost)."""
    import math
    rs = np.random.RandomState(seed)
    young, old = [], []
    mass = 0
    while mass < young_bases:
        unit, copies = int(rs.choice([300, 600, 1200, 2400, 6000])), int(rs.randint(4, 21))
        young.append((unit, copies, float(rs.uniform(0.01, 0.03))))
        mass += unit * copies
    mass = 0
    while mass < old_bases:
        unit, div = int(rs.choice([300, 600, 1200, 2400, 6000])), float(rs.uniform(0.05, 0.15))
        s = (1 - _pairwise_q(div)) ** 16
        copies = int(max(4, min(cand_per_base / (1.5 * s), 100000)))
        old.append((unit, copies, div))
        mass += unit * copies
    return young, old


def workload_grch38_like(seed: int = 38, scale: float = 1.0, young_share: float = 0.05, old_share: float = 0.40,
                         n_share: float = 0.05):
    """configs[2]: 25 contigs with the GRCh38 primary-assembly lengths (x scale), 5 % N in runs
    of >= 10 kbp (telomere ends, one centromere-like run per contig, shorter runs between), 45 % of the bases drawn from
    repeat families of 300 bp - 6 kbp units (plan_repeat_families: how many families and why the copy numbers are capped),
    every third copy reverse-complemented, old-family copies soft-masked (lower case) as in the real assembly.
    -> [(name, uint8 ASCII array)] (views of one array)."""
    rs = np.random.RandomState(seed)
    lens = [max(2000, int(round(l * scale))) for l in GRCH38_LENGTHS]
    total = int(sum(lens))
    lut = np.zeros(256, dtype=np.uint32)
    for b in range(256):
        lut[b] = sum(int(_ACGT[(b >> (2 * k)) & 3]) << (8 * k) for k in range(4))
    nb = (total + 3) // 4
    g = lut[np.frombuffer(rs.bytes(nb), dtype=np.uint8)].view(np.uint8)[:total]
    young, old = plan_repeat_families(seed + 1, int(young_share * total), int(old_share * total))
    fams = [(u, c, d, False) for u, c, d in young] + [(u, c, d, True) for u, c, d in old]
    # all instances in one shuffled sequence along the concatenated contigs, random gaps between them
    inst_len = np.concatenate([np.full(c, u, dtype=np.int64) for u, c, _, _ in fams]) if fams else np.zeros(0, dtype=np.int64)
    n_inst = len(inst_len)
    order = rs.permutation(n_inst)
    free = total - int(inst_len.sum())
    if n_inst and free > 0:
        gaps = rs.random_sample(n_inst + 1)
        gaps = np.floor(gaps / gaps.sum() * free).astype(np.int64)
        starts_in_order = np.cumsum(gaps[:-1] + np.concatenate([[0], inst_len[order][:-1]]))
        start = np.empty(n_inst, dtype=np.int64)
        start[order] = starts_in_order
        at = 0
        for unit, copies, div, soft in fams:
            codes = rs.randint(0, 4, size=unit).astype(np.uint8)
            m = np.tile(codes, copies)
            nm = int(rs.binomial(unit * copies, div))
            idx = rs.randint(0, unit * copies, size=nm)
            m[idx] = (m[idx] + rs.randint(1, 4, size=nm).astype(np.uint8)) & 3
            rows = m.reshape(copies, unit)
            rows[2::3] = 3 - rows[2::3, ::-1]
            rows = _ACGT[rows]
            if soft:
                rows |= 0x20
            for c in range(copies):
                a = int(start[at + c])
                g[a:a + unit] = rows[c]
            at += copies
    # N runs: per contig 10 kbp at either end, a centromere-like run of 3 %, shorter runs to n_share in all
    contigs, off = [], 0
    min_run = max(100, int(10000 * min(1.0, scale * 50)))
    for name, ln in zip(GRCH38_NAMES, lens):
        c = g[off:off + ln]
        off += ln
        if ln >= 20 * min_run:
            c[:min_run] = ord("N"); c[ln - min_run:] = ord("N")
            cen = int(0.03 * ln)
            c0 = int(ln * rs.uniform(0.3, 0.5))
            c[c0:c0 + cen] = ord("N")
            left = int(n_share * ln) - 2 * min_run - cen
            while left > 0:
                run = int(rs.randint(min_run, 5 * min_run + 1))
                a = int(rs.randint(0, ln - run))
                c[a:a + run] = ord("N")
                left -= run
        contigs.append((name, c))
    return contigs


def ensure_dir(p: str) -> str:
    os.makedirs(p, exist_ok=True)
    return p


def fastq_image(seed: int, ref: np.ndarray, n: int, L: int, dup_frac: float = 0.25, novel_frac: float = 0.1,
                mut_frac: float = 0.15, n_frac: float = 0.01) -> bytes:
    """n reads of length L as one FASTQ image (read side of the pipeline, /root/reference/collapse.cpp:923-935 reads four
    lines per record): stretches of ``ref`` (an N-free uint8 ASCII array) in either strand, a share of them taken from a
    small pool of positions (heavy duplicates), a share with one substitution, a share with one N, and a share of random
    reads that are not in the reference.  Headers are ``@r`` + nine digits, qualities ``I``."""
    rs = np.random.RandomState(seed)
    pos = rs.randint(0, len(ref) - L + 1, size=n).astype(np.int64)
    pool = rs.randint(0, len(ref) - L + 1, size=max(1, n // 200)).astype(np.int64)
    kind = rs.random_sample(n)
    dup = kind < dup_frac
    pos[dup] = pool[rs.randint(0, len(pool), size=int(dup.sum()))]
    reads = ref[pos[:, None] + np.arange(L)[None, :]]
    novel = kind > 1.0 - novel_frac
    reads[novel] = _ACGT[rs.randint(0, 4, size=(int(novel.sum()), L)).astype(np.uint8)]
    mut = np.nonzero((kind >= dup_frac) & (kind < dup_frac + mut_frac))[0]
    at = rs.randint(0, L, size=len(mut))
    code = (np.searchsorted(_ACGT, reads[mut, at]) + rs.randint(1, 4, size=len(mut))) & 3
    reads[mut, at] = _ACGT[code]
    withn = np.nonzero(rs.random_sample(n) < n_frac)[0]
    reads[withn, rs.randint(0, L, size=len(withn))] = ord("N")
    flip = np.nonzero(rs.randint(0, 2, size=n))[0]
    reads[flip] = _COMP[reads[flip, ::-1]]
    rec = np.empty((n, 12 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for k in range(9):
        rec[:, 10 - k] = 48 + (idx // 10 ** k) % 10
    rec[:, 11] = 10
    rec[:, 12:12 + L] = reads
    rec[:, 12 + L] = 10; rec[:, 13 + L] = ord("+"); rec[:, 14 + L] = 10
    rec[:, 15 + L:15 + 2 * L] = ord("I")
    rec[:, 15 + 2 * L] = 10
    return rec.tobytes()
