"""GPU: the HIP table loader (cora_amd/csrc/tables.hip, through the C-ABI) against the loader oracle and against the
pins taken from the real reference's loaders (tests/golden/manifest.json: loader_sha256).  Bit-exact, array by array."""
import os
import struct
import sys

import numpy as np
import pytest

import helpers as H
from cora_amd import hip, tables
from cora_amd.tables import ARRAYS

pytestmark = pytest.mark.gpu

CASES = sorted(n for n, c in H.MANIFEST["cases"].items() if not c.get("hash_only"))
LCASES = sorted(H.MANIFEST["loader_cases"])
sys.path.insert(0, H.GOLDEN)


@pytest.fixture(scope="module", autouse=True)
def _init():
    hip.init(0)


def oracle_arrays(dirn, c, name, exact_only=False):
    out = os.path.join(H.scratch(), "gl_%s%s.dump" % (name, "_eo" if exact_only else ""))
    r = H.run_loader_oracle(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], out, L=c["L"], exact_only=exact_only)
    assert r.returncode == 0, r.stdout
    return H.read_dump(out)


@pytest.mark.parametrize("name", CASES)
def test_golden_tables_load_like_the_reference(name):
    c, dirn = H.golden_case(name)
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], chr_short=c["L"])
    got = t.arrays()
    H.assert_same_arrays(got, oracle_arrays(dirn, c, name), name)
    assert H.dump_sha256(got) == c["loader_sha256"]
    info = t.info()
    assert info["serial_hub_replay"] == 0, "a builder-made table must take the parallel hub path"
    assert info["exact_records"] == c["exact_nodes"] and info["inexact_nodes"] == c["inexact_nodes"]
    t.close()


@pytest.mark.parametrize("name", LCASES)
def test_handmade_tables_load_like_the_reference(name):
    c = H.MANIFEST["loader_cases"][name]
    dirn = os.path.join(H.LOADER_CASES, name)
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], chr_short=c["L"])
    got = t.arrays()
    H.assert_same_arrays(got, H.read_dump(os.path.join(dirn, "reference.dump")), name)
    assert H.dump_sha256(got) == c["loader_sha256"]
    import make_loader_golden as mg
    assert t.info()["serial_hub_replay"] == int(_order_matters(*mg.LOADER_CASES[name]()[3:]))
    t.close()
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], chr_short=c["L"], exact_only=True)
    assert H.dump_sha256(t.arrays()) == c["loader_exact_only_sha256"]
    t.close()


def _lookup_bytes(t, lens):
    """RetrieveOffsetAndDirInfo for every position of every contig, in the 8-byte layout of the oracle's --lookup-all."""
    chr_codes = np.concatenate([np.full(n, c + 1, dtype=np.uint32) for c, n in enumerate(lens)]) if lens else np.zeros(0, np.uint32)
    pos = np.concatenate([np.arange(1, n + 1, dtype=np.uint32) for n in lens]) if lens else np.zeros(0, np.uint32)
    idx, off, dr = t.lookup_equiv(chr_codes, pos)
    rec = np.zeros(len(idx), dtype=[("idx", "<u4"), ("off", "<i2"), ("dir", "u1"), ("pad", "u1")])
    rec["idx"], rec["off"], rec["dir"] = idx, off, dr
    return rec.tobytes()


@pytest.mark.parametrize("name", CASES + ["loader:" + n for n in LCASES])
def test_class_lookup_of_every_position_matches_the_reference(name):
    import hashlib
    if name.startswith("loader:"):
        c = H.MANIFEST["loader_cases"][name[7:]]
        dirn = os.path.join(H.LOADER_CASES, name[7:])
    else:
        c, dirn = H.golden_case(name)
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], chr_short=c["L"])
    got = _lookup_bytes(t, c["contig_lengths"])
    assert hashlib.sha256(got).hexdigest() == c["lookup_sha256"]
    # keys the reference's maps could not hold are "not found"
    idx, off, dr = t.lookup_equiv([0, 1, len(c["contig_lengths"]) + 1, 1], [5, 0, 1, c["contig_lengths"][0] + 1])
    assert not idx.any() and not off.any() and not dr.any()
    t.close()


def _order_matters(records, nodes):
    """True if homHub depends on the order of the copies (MI:1123, 1133): a position copied from is also copied to
    by some other copy, or two copies put different values on one position."""
    first = {(c, p): i + 1 for i, (c, p, _) in enumerate(nodes)}
    copies = []
    for rc, rp, bl, members in records:
        for dr, mc, mp in members:
            for j in range(bl):
                t, s = (mc, mp + j if dr == 0 else mp - j), (rc, rp + j)
                copies.append((t, s))
    targets = {t for t, s in copies if t != s}
    if any(s in targets for t, s in copies):
        return True
    val = {}
    for t, s in copies:
        if val.setdefault(t, first.get(s, 0)) != first.get(s, 0):
            return True
    return False


def test_order_matters_only_in_the_handmade_tables():
    import make_loader_golden as mg
    got = {name: _order_matters(*mg.LOADER_CASES[name]()[3:]) for name in LCASES}
    assert got["weird_a"] and got["fuzz4_d2"] and not got["empty_inexact"] and not got["empty_both"]


def test_device_arrays_are_described():
    dirn = os.path.join(H.LOADER_CASES, "fuzz4_d2")
    c = H.MANIFEST["loader_cases"]["fuzz4_d2"]
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"])
    info = t.info()
    assert info["hub_positions"] == 8000 and info["d"] == 2 and info["chr_short"] == 0
    for name, dt in ARRAYS:
        p, n, eb = t.device_array(name)
        assert eb == np.dtype(dt).itemsize and (p or n == 0)
    assert t.device_array("hmi_offs")[1] == 2 * t.device_array("hmi_pos")[1]
    assert info["ms_total"] > 0
    t.close()


# ---- malformed input: clear errors, no crash ----

def _weird():
    import make_loader_golden as mg
    L, d, lens, records, nodes = mg.case_weird_b()
    return mg, L, d, lens, records, nodes


def test_truncated_and_miscounted_tables_are_rejected():
    mg, L, d, lens, records, nodes = _weird()
    ex, inx = mg.exact_bytes(records, L), mg.inexact_bytes(nodes, L, d)
    ok = tables.load_bytes(ex, len(records), inx, len(nodes), lens, d, chr_short=L)
    ok.close()
    for bad_ex, n_ex, bad_in, n_in in ((ex[:-3], len(records), inx, len(nodes)), (ex, len(records) + 1, inx, len(nodes)),
                                       (ex, len(records), inx[:-1], len(nodes)), (ex, len(records), inx, len(nodes) - 1),
                                       (ex, len(records), inx + b"\x00", len(nodes)), (ex + b"\xff", len(records), inx, len(nodes))):
        with pytest.raises(hip.CoraHipError) as e:
            tables.load_bytes(bad_ex, n_ex, bad_in, n_in, lens, d, chr_short=L)
        assert e.value.code == 91, str(e.value)


def test_nodes_out_of_order_or_empty_are_rejected():
    mg, L, d, lens, records, nodes = _weird()
    swapped = list(nodes)
    swapped[3], swapped[4] = swapped[4], swapped[3]
    with pytest.raises(hip.CoraHipError) as e:
        tables.load_bytes(mg.exact_bytes(records, L), len(records), mg.inexact_bytes(swapped, L, d), len(nodes), lens, d, chr_short=L)
    assert e.value.code == 92 and "order" in str(e.value)
    empty = list(nodes)
    empty[5] = (empty[5][0], empty[5][1], [])
    with pytest.raises(hip.CoraHipError) as e:
        tables.load_bytes(mg.exact_bytes(records, L), len(records), mg.inexact_bytes(empty, L, d), len(nodes), lens, d, chr_short=L)
    assert e.value.code == 92 and "no item" in str(e.value)


def test_positions_outside_the_contigs_are_rejected():
    mg, L, d, lens, records, nodes = _weird()
    ex, inx = mg.exact_bytes(records, L), mg.inexact_bytes(nodes, L, d)
    for short in ([300, 300, 100], [300, 300], [90, 300, 120]):
        with pytest.raises(hip.CoraHipError) as e:
            tables.load_bytes(ex, len(records), inx, len(nodes), short, d, chr_short=L)
        assert e.value.code == 93, str(e.value)


def test_missing_files_give_the_reference_exit_codes(tmp_path):
    dirn = os.path.join(H.LOADER_CASES, "weird_a")
    for code, missing in ((87, "Exact.itemCount"), (88, "Exact"), (89, "Inexact.itemCount"), (90, "Inexact")):
        d = tmp_path / str(code)
        d.mkdir()
        for fn in ("Exact", "Exact.itemCount", "Inexact", "Inexact.itemCount"):
            if fn != missing:
                (d / fn).write_bytes(H.read(os.path.join(dirn, fn)))
        with pytest.raises(hip.CoraHipError) as e:
            tables.load_homology_tables(str(d / "Exact"), str(d / "Inexact"), 2, [400])
        assert e.value.code == code


# ---- large: parser states across many workgroups, long reverse-lookup runs, tables straight from a job ----

def test_many_parser_workgroups_and_long_runs():
    """300 k nodes (5 MB of <Inexact>: 300 parser workgroups, so the workgroup-map scan loops) with a 40 k-node stretch
    of consecutive positions (one reverse-lookup run walked by a single thread) and 0xFF bytes inside positions."""
    rng = np.random.default_rng(11)
    G = 0x0200FFFF + 70000
    pos = np.unique(np.concatenate([rng.integers(1, 0x01000000, 200000), np.arange(0x00FFFF00, 0x00FFFF00 + 40000),
                                    rng.integers(0x01FFFF00, 0x0200FFFF, 60000)])).astype(np.int64)
    nodes, recs = [], []
    n = len(pos)
    r_dir, r_tgt = rng.integers(0, 2, (n, 2)).tolist(), rng.integers(1, G, (n, 2)).tolist()
    r_off, r_bl = np.sort(rng.integers(1, 64, (n, 2, 2)), axis=2).tolist(), rng.integers(1, 64, (n, 2)).tolist()
    for i, p in enumerate(pos.tolist()):
        nodes.append((1, p, [(r_dir[i][k], 1, r_tgt[i][k], r_off[i][k], r_bl[i][k]) for k in range(1 + (i % 3 == 0))]))
    r_bl, r_rep = rng.integers(1, 9, 30000).tolist(), rng.integers(1, 0x00800000, 30000).tolist()
    for i in range(30000):
        recs.append((1, r_rep[i], r_bl[i], [(0, 1, 0x01000000 + i * 16)] + ([(1, 1, 0x01800000 + i * 16 + 8)] if i % 2 else [])))
    import make_loader_golden as mg
    dirn = os.path.join(H.scratch(), "loader_large")
    mg.write_tables(dirn, recs, nodes, 0, 2)
    c = {"d": 2, "contig_lengths": [G], "L": 0}
    t = tables.load_homology_tables(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), 2, [G])
    got = t.arrays()
    H.assert_same_arrays(got, oracle_arrays(dirn, c, "large"), "large")
    assert t.info()["serial_hub_replay"] == 0
    assert got[10].max() > 20      # hm_maxrev
    t.close()


def test_tables_of_a_job_load_without_files():
    """chr20-like 64 Mbp, K=50, d=2: build on the GPU, hand the tables over in memory, compare with the reference's
    loaders run over the reference's own tables (pin from make_loader_golden.py --big)."""
    c = H.MANIFEST["cases"]["chr20_like_k50_d2"]
    fa = H.input_path(c["input"])
    job = hip.Job(c["K"], c["d"], H.split_path("phys", c["p"]), H.split_path("par", c["t"]))
    job.set_reference_file(fa)
    job.run()
    t = tables.from_job(job)
    info = t.info()
    assert info["exact_records"] == c["exact_nodes"] and info["inexact_nodes"] == c["inexact_nodes"]
    assert info["serial_hub_replay"] == 0
    assert H.dump_sha256(t.arrays()) == c["loader_sha256"]
    # every one of the 64 M positions through the class lookup, against the oracle's
    ex, en, inx, inn = job.tables()
    dirn = os.path.join(H.scratch(), "job_tables")
    os.makedirs(dirn, exist_ok=True)
    for fn, data in (("Exact", ex), ("Inexact", inx), ("Exact.itemCount", b"%d\n" % en), ("Inexact.itemCount", b"%d\n" % inn)):
        open(os.path.join(dirn, fn), "wb").write(data)
    lk = os.path.join(dirn, "lookup.bin")
    r = H.run_loader_oracle(os.path.join(dirn, "Exact"), os.path.join(dirn, "Inexact"), c["d"], c["contig_lengths"], os.path.join(dirn, "o.dump"), lookup_all=lk)
    assert r.returncode == 0, r.stdout
    assert _lookup_bytes(t, c["contig_lengths"]) == H.read(lk)
    print("loader: parse %.3f ms, reverse lookups %.3f ms, hubs %.3f ms, total %.3f ms, %.1f MB in HBM" % (
        info["ms_parse"], info["ms_lookups"], info["ms_hubs"], info["ms_total"], info["device_bytes"] / 1e6))
    t.close()
    job.close()
