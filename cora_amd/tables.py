"""Host-side mirror of the consumer's table loading (MI = /root/reference/mappingInference.cpp):
``load_homology_tables`` stands where ``LoadHomologyTables`` (MI:1084-1153) does, with the same inputs -- the two table
files, their ``.itemCount`` companions, d, and the mapping mode's "exact only" switch -- and the same failures (the
reference's exit codes 87-90 for missing files become ``CoraHipError`` codes).  All work happens in libcora_hip.so on
the GPU (cora_amd/csrc/tables.hip); there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hip

# order and element types of include/cora_hip.h's enum cora_hip_tables_array (= oracle/loader_dump.h's dump order)
ARRAYS = [
    ("eq_chr", np.uint16), ("eq_pos", np.uint32), ("eq_block", np.uint32), ("eq_off", np.uint64),
    ("eqi_dir", np.uint8), ("eqi_chr", np.uint16), ("eqi_pos", np.uint32),
    ("hm_chr", np.uint16), ("hm_pos", np.uint32), ("hm_off", np.uint64), ("hm_maxrev", np.uint8),
    ("hmi_dir", np.uint8), ("hmi_chr", np.uint16), ("hmi_pos", np.uint32), ("hmi_block", np.uint8), ("hmi_offs", np.uint8),
    ("equiv_hub", np.uint32), ("hom_hub", np.uint32),
]


def read_item_count(table_path, missing_code):
    """``<table>.itemCount`` holds one decimal number (MI:559-567, 657-665)."""
    p = table_path + ".itemCount"
    if not os.path.exists(p):
        raise hip.CoraHipError(missing_code, "item count file does not exist: " + p)
    txt = open(p).read().split()
    return int(txt[0]) if txt else 0


class LoadedTables:
    """The loaded image in HBM.  ``array(name)`` copies one array to the host; ``device_array(name)`` gives
    (device pointer, elements, element bytes) for the caller's own kernels."""

    def __init__(self, handle):
        self._h = handle

    def info(self):
        i = hip.TablesInfo()
        hip._check(hip.lib().cora_hip_tables_get_info(self._h, C.byref(i)))
        return i.as_dict()

    def device_array(self, name):
        which = [n for n, _ in ARRAYS].index(name)
        p, n, eb = C.c_void_p(), C.c_uint64(), C.c_uint32()
        hip._check(hip.lib().cora_hip_tables_array_ptr(self._h, which, C.byref(p), C.byref(n), C.byref(eb)))
        return p.value, n.value, eb.value

    def array(self, name):
        which = [n for n, _ in ARRAYS].index(name)
        _, n, eb = self.device_array(name)
        out = np.zeros(n, dtype=ARRAYS[which][1])
        assert out.itemsize == eb
        hip._check(hip.lib().cora_hip_tables_export(self._h, which, out.ctypes.data, out.nbytes))
        return out

    def arrays(self):
        return [self.array(n) for n, _ in ARRAYS]

    def lookup_equiv(self, chr_codes, positions):
        """RetrieveOffsetAndDirInfo (MI:6618-6674) for arrays of (contig code, position): -> (class index, block
        offset, direction) arrays; class index 0 = the position is in no exact class."""
        c = np.ascontiguousarray(chr_codes, dtype=np.uint32)
        p = np.ascontiguousarray(positions, dtype=np.uint32)
        assert c.shape == p.shape
        idx, off, dr = np.zeros(len(c), np.uint32), np.zeros(len(c), np.int16), np.zeros(len(c), np.uint8)
        hip._check(hip.lib().cora_hip_tables_lookup_equiv(self._h, c.ctypes.data, p.ctypes.data, len(c), 0,
                                                          idx.ctypes.data, off.ctypes.data, dr.ctypes.data))
        return idx, off, dr

    def close(self):
        if self._h:
            hip.lib().cora_hip_tables_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_bytes(exact: bytes, exact_count: int, inexact: bytes, inexact_count: int, contig_lengths, d: int,
               chr_short=False, exact_only=False) -> LoadedTables:
    lens = np.asarray(contig_lengths, dtype=np.uint32)
    h = C.c_void_p()
    hip._check(hip.lib().cora_hip_tables_load(exact, len(exact), exact_count, inexact, len(inexact), inexact_count,
                                               lens.ctypes.data, len(lens), d, int(bool(chr_short)), int(bool(exact_only)), C.byref(h)))
    return LoadedTables(h)


def load_homology_tables(perf_table, inexact_table, d, contig_lengths, chr_short=False, exact_only=False) -> LoadedTables:
    """LoadHomologyTables(perfTableFileName, inexactTableFileName, readLen, numHomTableMismatches, mapMode): the exact
    table first (MI:1088; 87 = no item count, 88 = no table), then, unless the mode is exact-only, the inexact one
    (MI:1090-1091; 89, 90).  readLen is not used by the loaders and is not asked for."""
    n_eq = read_item_count(perf_table, 87)
    if not os.path.exists(perf_table):
        raise hip.CoraHipError(88, "homology table file does not exist: " + perf_table)
    exact = open(perf_table, "rb").read()
    n_hm, inexact = 0, b""
    if not exact_only:
        n_hm = read_item_count(inexact_table, 89)
        if not os.path.exists(inexact_table):
            raise hip.CoraHipError(90, "could not find inexact homology table: " + inexact_table)
        inexact = open(inexact_table, "rb").read()
    return load_bytes(exact, n_eq, inexact, n_hm, contig_lengths, d, chr_short, exact_only)


def from_job(job: "hip.Job", exact_only=False) -> LoadedTables:
    """The tables of a job's last run, loaded without the trip through files."""
    h = C.c_void_p()
    hip._check(hip.lib().cora_hip_tables_from_job(job._h, int(bool(exact_only)), C.byref(h)))
    return LoadedTables(h)
