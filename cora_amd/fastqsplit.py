"""Host-side mirror of CORA's `fastqSplit` (/root/reference/fastqSplit.cpp) over libcora_hip.so: argument meaning as the
reference's argv (signals file, read length, mapping mode).  Plumbing for tests and the benchmark; the product boundary is
the C-ABI (cora_hip_fqsplit_*) and cora_amd/bin/fastqSplit."""
from __future__ import annotations

import ctypes as C

from . import hip

MODES = {"SINGLE": (1, 0), "SPLIT_SINGLE": (2, 0), "THREEWAY_SINGLE": (3, 0),
         "PAIRED": (1, 1), "SPLIT_PAIRED": (2, 1), "THREEWAY_PAIRED": (3, 1)}       # fastqSplit.cpp:353-604


class FqSplitParams(C.Structure):
    _fields_ = [("read_len", C.c_int32), ("splits", C.c_int32), ("paired", C.c_int32), ("signal_size", C.c_int32),
                ("n_groups", C.c_int32), ("group_of", C.c_uint8 * 628)]


class FqSplitStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads", "read_mers", "record_bytes", "output_bytes", "image_bytes")] + [
        (n, C.c_float) for n in ("ms_parse", "ms_classify", "ms_order", "ms_write", "ms_device", "ms_total")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _bind(lib):
    if getattr(lib, "_fqsplit_bound", False):
        return lib
    lib.cora_hip_fqsplit_parse_signals.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FqSplitParams)]
    lib.cora_hip_fqsplit_create.argtypes = [C.POINTER(FqSplitParams), C.POINTER(C.c_void_p)]
    lib.cora_hip_fqsplit_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.cora_hip_fqsplit_run.argtypes = [C.c_void_p]
    lib.cora_hip_fqsplit_group.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.cora_hip_fqsplit_group_device.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.cora_hip_fqsplit_get_stats.argtypes = [C.c_void_p, C.POINTER(FqSplitStats)]
    lib.cora_hip_fqsplit_destroy.argtypes = [C.c_void_p]
    lib.cora_hip_fqsplit_destroy.restype = None
    lib._fqsplit_bound = True
    return lib


def parse_signals(text: bytes) -> FqSplitParams:
    """The signals file (written by cora.cpp:335-390) -> signal size, group count and the signal -> group table.  Host only."""
    p = FqSplitParams()
    hip._check(_bind(hip.lib()).cora_hip_fqsplit_parse_signals(text, len(text), C.byref(p)))
    return p


class FastqSplit:
    def __init__(self, signals_text: bytes, read_len: int, mode: str):
        if mode not in MODES:
            raise ValueError("ERROR: Mapping mode undefined... ")
        self._lib = _bind(hip.lib())
        p = parse_signals(signals_text)
        p.read_len = read_len
        p.splits, p.paired = MODES[mode]
        self.n_groups = p.n_groups
        self._h = C.c_void_p()
        hip._check(self._lib.cora_hip_fqsplit_create(C.byref(p), C.byref(self._h)))

    def add_reads(self, fastq_image: bytes):
        buf = (C.c_char * max(1, len(fastq_image))).from_buffer_copy(fastq_image or b"\0")
        hip._check(self._lib.cora_hip_fqsplit_add_reads(self._h, C.cast(buf, C.c_void_p), len(fastq_image)))

    def set_device_only(self, on: bool = True):
        f = self._lib.cora_hip_fqsplit_set_device_only
        f.argtypes = [C.c_void_p, C.c_int32]
        hip._check(f(self._h, int(on)))

    def run(self):
        hip._check(self._lib.cora_hip_fqsplit_run(self._h))

    def group(self, g: int) -> bytes:
        ptr, n = C.c_void_p(), C.c_size_t()
        hip._check(self._lib.cora_hip_fqsplit_group(self._h, g, C.byref(ptr), C.byref(n)))
        return C.string_at(ptr, n.value) if n.value else b""

    def group_device(self, g: int):
        """(device pointer, bytes) of the group's fragment file where the run left it in HBM."""
        ptr, n = C.c_void_p(), C.c_size_t()
        hip._check(self._lib.cora_hip_fqsplit_group_device(self._h, g, C.byref(ptr), C.byref(n)))
        return (ptr.value or 0), n.value

    def groups(self):
        return [self.group(g) for g in range(self.n_groups)]

    def stats(self):
        st = FqSplitStats()
        hip._check(self._lib.cora_hip_fqsplit_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            self._lib.cora_hip_fqsplit_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
