"""Host-side mirror of CORA's `collapse` (/root/reference/collapse.cpp, whole-input and fragmented mode) over libcora_hip.so:
argument meaning as the reference's argv (input mode, split mode, mapper mode, digit length, LRN offset).  Plumbing for
tests and the benchmark; the product boundary is the C-ABI (cora_hip_collapse_*) and cora_amd/bin/collapse[_L]."""
from __future__ import annotations

import ctypes as C

from . import hip

SPLITS = {"FULL": 1, "HALF": 2, "THREEWAY": 3}
MAPPERS = {"BWA": 1, "BWA_MEM": 2, "BOWTIE_2": 3}          # anything else is the generic mode (collapse.cpp:590-611)


def table_geometry():
    b, r = C.c_uint64(), C.c_uint64()
    hip._check(hip.lib().cora_hip_collapse_table_geometry(C.byref(b), C.byref(r)))
    return b.value, r.value


class Collapse:
    def __init__(self, digit_len, read_len, input_mode="SINGLE", split_mode="FULL", mapper="GENERIC", chr_short=False, lrn_offset=0):
        if split_mode not in SPLITS:
            raise ValueError("Split mode unrecognized")
        if input_mode not in ("SINGLE", "PAIRED"):
            raise ValueError("Input mode unrecognized")
        p = hip.CollapseParams()
        p.digit_len, p.read_len, p.splits = digit_len, read_len, SPLITS[split_mode]
        p.paired, p.mapper, p.chr_short, p.lrn_offset = int(input_mode == "PAIRED"), MAPPERS.get(mapper, 0), int(chr_short), lrn_offset
        self._h = C.c_void_p()
        hip._check(hip.lib().cora_hip_collapse_create(C.byref(p), C.byref(self._h)))

    def set_reference(self, fasta_image: bytes):
        fa = hip.Fasta(fasta_image)
        try:
            hip._check(hip.lib().cora_hip_collapse_set_reference(self._h, fa._h))
        finally:
            fa.close()

    def set_fragment(self, signals: str, signal_len: int):
        """Fragmented mode (collapse.cpp:542-563): `signals` = the group's line of the signals file; the one input added
        afterwards is that group's fragment file of fastqSplit."""
        f = hip.lib().cora_hip_collapse_set_fragment
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        hip._check(f(self._h, signals.encode(), signal_len))

    def set_lrn_offset(self, lrn_offset: int):
        f = hip.lib().cora_hip_collapse_set_lrn_offset
        f.argtypes = [C.c_void_p, C.c_uint64]
        hip._check(f(self._h, lrn_offset))

    def add_reads_device(self, dev_ptr: int, nbytes: int):
        """Fragmented mode: a fragment already in HBM (FastqSplit.group_device); borrowed until run() returns."""
        f = hip.lib().cora_hip_collapse_add_reads_device
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        hip._check(f(self._h, C.c_void_p(dev_ptr), nbytes))

    def add_reads(self, fastq_image: bytes):
        buf = (C.c_char * max(1, len(fastq_image))).from_buffer_copy(fastq_image or b"\0")
        hip._check(hip.lib().cora_hip_collapse_add_reads(self._h, C.cast(buf, C.c_void_p), len(fastq_image)))

    def run(self):
        hip._check(hip.lib().cora_hip_collapse_run(self._h))

    def outputs(self):
        ptr = [C.c_void_p() for _ in range(3)]
        n = [C.c_size_t() for _ in range(3)]
        hip._check(hip.lib().cora_hip_collapse_outputs(self._h, C.byref(ptr[0]), C.byref(n[0]), C.byref(ptr[1]), C.byref(n[1]),
                                                       C.byref(ptr[2]), C.byref(n[2])))
        return [C.string_at(p, k.value) if k.value else b"" for p, k in zip(ptr, n)]

    def stats(self):
        st = hip.CollapseStats()
        hip._check(hip.lib().cora_hip_collapse_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            hip.lib().cora_hip_collapse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
