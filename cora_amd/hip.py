"""ctypes binding of libcora_hip.so (include/cora_hip.h) plus the host-side
set-up the reference's main() does before its hot loops: split files, .fai,
FASTA line width (/root/reference/homTable_setup.cpp:2407-2471, 144-202).

Python here is plumbing for the tests and the benchmark; the product boundary
is the C-ABI and the ``homTable_setup`` executable in ``cora_amd/bin``.
There is no CPU path: importing works anywhere (symbol checks), running needs
a gfx950 device and raises otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcora_hip.so")
BIN_DIR = os.path.join(_HERE, "bin")

EXPORTS = [
    "cora_hip_init", "cora_hip_shutdown", "cora_hip_last_error", "cora_hip_job_create", "cora_hip_job_destroy",
    "cora_hip_job_set_reference", "cora_hip_job_run", "cora_hip_job_run_multi", "cora_hip_device_count", "cora_hip_job_tables", "cora_hip_job_stats", "cora_hip_job_set_table_files", "cora_hip_job_finish_table_files",
    "cora_hip_radix_sort", "cora_hip_radix_sort_noalloc", "cora_hip_radix_sort_noalloc_max", "cora_hip_test_sort_host", "cora_hip_test_sort_chain_host", "cora_hip_test_sort_kept_host", "cora_hip_test_list_sort", "cora_hip_test_pack", "cora_hip_test_kmers", "cora_hip_test_window_bits",
    "cora_hip_shard_begin", "cora_hip_shard_stage", "cora_hip_shard_slot_info", "cora_hip_shard_slot_export",
    "cora_hip_shard_slot_import", "cora_hip_shard_slot_reserve", "cora_hip_shard_slot_ptr", "cora_hip_shard_seed_sketch", "cora_hip_shard_filter", "cora_hip_shard_get_first_singleton", "cora_hip_shard_set_first_singleton",
    "cora_hip_tables_load", "cora_hip_tables_from_job", "cora_hip_tables_get_info", "cora_hip_tables_array_ptr",
    "cora_hip_tables_export", "cora_hip_tables_lookup_equiv", "cora_hip_tables_destroy",
    "cora_hip_fasta_scan", "cora_hip_fasta_get_info", "cora_hip_fasta_fai_text", "cora_hip_fasta_contig",
    "cora_hip_fasta_destroy", "cora_hip_job_set_reference_fasta",
    "cora_hip_collapse_table_geometry", "cora_hip_collapse_table_growth", "cora_hip_collapse_create", "cora_hip_collapse_set_reference", "cora_hip_collapse_set_fragment", "cora_hip_collapse_set_lrn_offset", "cora_hip_collapse_add_reads", "cora_hip_collapse_add_reads_device",
    "cora_hip_collapse_run", "cora_hip_collapse_outputs", "cora_hip_collapse_get_stats", "cora_hip_collapse_destroy",
    "cora_hip_fqsplit_parse_signals", "cora_hip_fqsplit_create", "cora_hip_fqsplit_add_reads", "cora_hip_fqsplit_run",
    "cora_hip_fqsplit_group", "cora_hip_fqsplit_group_device", "cora_hip_fqsplit_set_device_only", "cora_hip_fqsplit_get_stats", "cora_hip_fqsplit_destroy",
]


class Contig(C.Structure):
    _fields_ = [("file_offset", C.c_uint64), ("length", C.c_uint32), ("reserved", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("K", C.c_int32), ("d", C.c_int32), ("chr_short", C.c_int32), ("line_width", C.c_int32),
                ("n_groups", C.c_int32), ("n_threads", C.c_int32), ("sig2", C.c_uint8 * 16), ("sig4", C.c_uint8 * 256)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "genome_bases", "kmers", "representatives", "exact_classes", "exact_nodes", "candidate_pairs", "hom_pairs",
        "inexact_items", "inexact_nodes", "dirty_nodes", "big_nodes")] + [(n, C.c_float) for n in (
        "ms_pack", "ms_exact", "ms_seeds", "ms_items", "ms_host_finish", "ms_total", "ms_sort_scatter")] + [
        ("n_sort_scatter", C.c_uint32), ("sort_scatter_bytes", C.c_uint64), ("hash_mixed_runs", C.c_uint64),
        ("full_key_fallback", C.c_uint64), ("seed_sketch", C.c_uint64), ("verify_slots", C.c_uint64), ("pool_peak_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class TablesInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "exact_records", "exact_entries", "inexact_nodes", "inexact_items", "hub_positions", "device_bytes",
        "serial_hub_replay")] + [("d", C.c_int32), ("chr_short", C.c_int32)] + [(n, C.c_float) for n in (
        "ms_parse", "ms_lookups", "ms_hubs", "ms_total")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FastaInfo(C.Structure):
    _fields_ = [("image_bytes", C.c_uint64), ("lines", C.c_uint64), ("contigs", C.c_uint64), ("bases", C.c_uint64),
                ("conforming", C.c_int32), ("line_width", C.c_int32), ("ms_scan", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CollapseParams(C.Structure):
    _fields_ = [("digit_len", C.c_int32), ("read_len", C.c_int32), ("splits", C.c_int32), ("paired", C.c_int32),
                ("mapper", C.c_int32), ("chr_short", C.c_int32), ("lrn_offset", C.c_uint64)]


class CollapseStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "reads", "read_mers", "ref_windows", "distinct", "printed", "lrn_names", "hash_collisions", "reads_bytes",
        "perfect_bytes", "lrn_bytes")] + [(n, C.c_float) for n in (
        "ms_parse", "ms_keys", "ms_group", "ms_order", "ms_text", "ms_device", "ms_total", "ms_sort_scatter")] + [
        ("n_sort_scatter", C.c_uint32), ("sort_scatter_bytes", C.c_uint64), ("table_growth_steps", C.c_uint32), ("reserved0", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CoraHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcora_hip: %s (status %d)" % (msg, code))
        self.code = code


_lib = None


def lib():
    """Loads libcora_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s is missing: build it with `make -C cora_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        # PyTorch ships its own HIP/HSA runtime libraries.  If it is imported AFTER this library has pulled in
        # /opt/rocm's, its runtime binds to the wrong HSA and reports "No HIP GPUs are available" (seen when the
        # sharded driver, which moves records with torch.distributed, ran after plain table builds in one
        # process).  Loading torch's libraries first is harmless for callers that never use it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.cora_hip_last_error.restype = C.c_char_p
        L.cora_hip_job_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.cora_hip_job_destroy.argtypes = [C.c_void_p]
        L.cora_hip_job_destroy.restype = None
        L.cora_hip_job_set_reference.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Contig), C.c_int32]
        L.cora_hip_job_run.argtypes = [C.c_void_p]
        L.cora_hip_job_run_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cora_hip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.cora_hip_job_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.cora_hip_job_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.cora_hip_job_set_table_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.cora_hip_job_finish_table_files.argtypes = [C.c_void_p]
        L.cora_hip_radix_sort.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_int, C.c_int]
        L.cora_hip_radix_sort_noalloc.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_int, C.c_int]
        L.cora_hip_radix_sort_noalloc_max.restype = C.c_uint64
        L.cora_hip_test_sort_host.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_int, C.c_int]
        L.cora_hip_test_sort_chain_host.argtypes = [C.c_void_p] * 2 + [C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.cora_hip_test_sort_kept_host.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p]
        L.cora_hip_test_list_sort.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.cora_hip_test_pack.argtypes = [C.c_void_p] * 5
        L.cora_hip_test_kmers.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
        L.cora_hip_test_window_bits.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p]
        L.cora_hip_shard_begin.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cora_hip_shard_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cora_hip_shard_slot_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.cora_hip_shard_slot_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cora_hip_shard_slot_import.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.cora_hip_shard_slot_ptr.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cora_hip_shard_slot_reserve.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
        L.cora_hip_shard_seed_sketch.argtypes = [C.c_void_p, C.c_void_p]
        L.cora_hip_shard_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cora_hip_shard_get_first_singleton.argtypes = [C.c_void_p, C.c_void_p]
        L.cora_hip_shard_set_first_singleton.argtypes = [C.c_void_p, C.c_uint64]
        L.cora_hip_tables_load.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint64,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.cora_hip_tables_from_job.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.cora_hip_tables_get_info.argtypes = [C.c_void_p, C.POINTER(TablesInfo)]
        L.cora_hip_tables_array_ptr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cora_hip_tables_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.cora_hip_tables_lookup_equiv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cora_hip_tables_destroy.argtypes = [C.c_void_p]
        L.cora_hip_tables_destroy.restype = None
        L.cora_hip_fasta_scan.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.cora_hip_fasta_get_info.argtypes = [C.c_void_p, C.POINTER(FastaInfo)]
        L.cora_hip_fasta_fai_text.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.cora_hip_fasta_contig.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cora_hip_fasta_destroy.argtypes = [C.c_void_p]
        L.cora_hip_fasta_destroy.restype = None
        L.cora_hip_job_set_reference_fasta.argtypes = [C.c_void_p, C.c_void_p]
        L.cora_hip_collapse_table_geometry.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cora_hip_collapse_create.argtypes = [C.POINTER(CollapseParams), C.POINTER(C.c_void_p)]
        L.cora_hip_collapse_set_reference.argtypes = [C.c_void_p, C.c_void_p]
        L.cora_hip_collapse_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.cora_hip_collapse_run.argtypes = [C.c_void_p]
        L.cora_hip_collapse_outputs.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.cora_hip_collapse_get_stats.argtypes = [C.c_void_p, C.POINTER(CollapseStats)]
        L.cora_hip_collapse_destroy.argtypes = [C.c_void_p]
        L.cora_hip_collapse_destroy.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise CoraHipError(rc, lib().cora_hip_last_error().decode())


def init(device: int = 0):
    _check(lib().cora_hip_init(device))


# --------------------------------------------------------------------------
# host set-up mirrored from the reference's main()
# --------------------------------------------------------------------------
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_RC = {"A": "T", "C": "G", "G": "C", "T": "A"}


def parse_split_file(path: str, k: int):
    """-> (count, table) with table[index of k-mer] = group id (HTS:2295-2346, 2499-2509).
    Every token and its reverse complement take the group of their line; later lines win."""
    with open(path) as f:
        lines = f.read().split("\n")
    count = int(lines[0])
    table = [0] * (4 ** k)
    body = lines[1:]
    if k == 2:      # read until EOF (HTS:2499); a trailing empty string after the last '\n' is not a line
        if body and body[-1] == "":
            body = body[:-1]
        if len(body) != count:
            raise ValueError("2-char split file announces %d groups, holds %d" % (count, len(body)))
    else:
        body = body[:count]
    for g, line in enumerate(body, 1):
        toks = line.split()
        if k == 2 and not toks:
            raise ValueError("empty 2-char signal group")
        for t in toks:
            if len(t) != k:
                raise ValueError("bad %d-char signal %r" % (k, t))
            for s in (t, "".join(_RC[c] for c in reversed(t))):
                idx = 0
                for c in s:
                    idx = idx * 4 + _CODE[c]
                table[idx] = g
    return count, table


def fasta_geometry(image: bytes, fai_path: str):
    """Line width (length of the image's 2nd token, HTS:155-166) and, per .fai row, the offset of the
    contig's first base under the fixed-line-width layout the reference's loader assumes (HTS:96-142)."""
    toks = image[:65536].split(None, 2)
    if len(toks) < 2 or not toks[0].startswith(b">"):
        raise ValueError("reference is not FASTA")
    W = len(toks[1])
    contigs = []
    cursor = 0
    with open(fai_path) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 2:
                continue
            size = int(parts[1])
            if cursor >= len(image) or image[cursor:cursor + 1] != b">":
                raise ValueError("contig %d does not start where the .fai lengths and the line width put it" % (len(contigs) + 1))
            nl = image.index(b"\n", cursor)
            contigs.append((nl + 1, size))
            cursor = nl + 1 + size + (size + W - 1) // W
    return W, contigs


class Fasta:
    """A FASTA image scanned on the device: the contig index of `cora faiGenerate` (cora.cpp:1006-1029)."""

    def __init__(self, image: bytes):
        self._h = C.c_void_p()
        self._n = len(image)
        buf = (C.c_char * max(len(image), 1)).from_buffer_copy(image or b"\0")
        _check(lib().cora_hip_fasta_scan(C.cast(buf, C.c_void_p), len(image), C.byref(self._h)))

    def info(self):
        fi = FastaInfo()
        _check(lib().cora_hip_fasta_get_info(self._h, C.byref(fi)))
        return fi.as_dict()

    def fai_text(self) -> bytes:
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().cora_hip_fasta_fai_text(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def contig(self, i):
        name, ln, off, lb = C.c_char_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().cora_hip_fasta_contig(self._h, i, C.byref(name), C.byref(ln), C.byref(off), C.byref(lb)))
        return name.value, ln.value, off.value, lb.value

    def close(self):
        if self._h:
            lib().cora_hip_fasta_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Job:
    """One reference, one device: set_reference() uploads, run() builds both tables."""

    def __init__(self, K, d, phys_path, par_path, chr_short=False):
        self.K, self.d, self.chr_short = K, d, bool(chr_short)
        self.p_count, sig2 = parse_split_file(phys_path, 2)
        self.t_count, sig4 = parse_split_file(par_path, 4)
        self._sig2, self._sig4 = sig2, sig4
        self._h = C.c_void_p()
        self._image = None

    def _create(self, line_width):
        prm = Params()
        prm.K, prm.d, prm.chr_short, prm.line_width = self.K, self.d, int(self.chr_short), line_width
        prm.n_groups, prm.n_threads = self.p_count, self.t_count
        for i, v in enumerate(self._sig2):
            prm.sig2[i] = v
        for i, v in enumerate(self._sig4):
            prm.sig4[i] = v
        _check(lib().cora_hip_job_create(C.byref(prm), C.byref(self._h)))

    def set_reference_file(self, fasta_path):
        with open(fasta_path, "rb") as f:
            image = f.read()
        W, contigs = fasta_geometry(image, fasta_path + ".fai")
        self.set_reference(image, W, contigs)

    def set_reference(self, image: bytes, line_width: int, contigs):
        if not self._h:
            self._create(line_width)
        arr = (Contig * len(contigs))()
        for i, (off, ln) in enumerate(contigs):
            arr[i].file_offset, arr[i].length = off, ln
        self._image = image
        self.n_contigs = len(contigs)
        buf = (C.c_char * len(image)).from_buffer_copy(image) if not isinstance(image, np.ndarray) else None
        ptr = C.cast(buf, C.c_void_p) if buf is not None else C.c_void_p(image.ctypes.data)
        _check(lib().cora_hip_job_set_reference(self._h, ptr, len(image), arr, len(contigs)))

    def set_reference_fasta(self, image: bytes):
        """No .fai, no fixed line width: contigs come from the device scan of the image."""
        if not self._h:
            self._create(1)
        fa = Fasta(image)
        try:
            self.fasta_info = fa.info()
            self.n_contigs = self.fasta_info["contigs"]
            self._image = image
            _check(lib().cora_hip_job_set_reference_fasta(self._h, fa._h))
        finally:
            fa.close()

    def run(self):
        _check(lib().cora_hip_job_run(self._h))

    def run_multi(self, world, devices=None):
        """One build over `world` ranks of THIS process (cora_hip_job_run_multi); devices may repeat (ranks sharing a GPU)."""
        arr = (C.c_int * world)(*devices) if devices is not None else None
        _check(lib().cora_hip_job_run_multi(self._h, world, arr))

    def tables(self):
        ex, inx = C.c_void_p(), C.c_void_p()
        exb, inb = C.c_size_t(), C.c_size_t()
        exn, inn = C.c_uint64(), C.c_uint64()
        _check(lib().cora_hip_job_tables(self._h, C.byref(ex), C.byref(exb), C.byref(exn), C.byref(inx), C.byref(inb), C.byref(inn)))
        e = C.string_at(ex, exb.value) if exb.value else b""
        i = C.string_at(inx, inb.value) if inb.value else b""
        return e, exn.value, i, inn.value

    def stats(self):
        st = Stats()
        _check(lib().cora_hip_job_stats(self._h, C.byref(st)))
        return st.as_dict()

    # ---- stage-level helpers for tests ----
    def pack(self):
        cap = (len(self._image) + 64 * (self.n_contigs + 2)) // 32 + 64
        packed = np.zeros(cap, dtype=np.uint64)
        bad = np.zeros(cap, dtype=np.uint32)
        starts = np.zeros(self.n_contigs, dtype=np.uint64)
        G = C.c_uint64()
        _check(lib().cora_hip_test_pack(self._h, packed.ctypes.data, bad.ctypes.data, starts.ctypes.data, C.byref(G)))
        nw = G.value // 32
        return packed[:nw], bad[:nw], starts, G.value

    def kmers(self):
        cap = len(self._image)
        hi = np.zeros(cap, dtype=np.uint64)
        lo = np.zeros(cap, dtype=np.uint64)
        gp = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64()
        _check(lib().cora_hip_test_kmers(self._h, hi.ctypes.data, lo.ctypes.data, gp.ctypes.data, cap, C.byref(n)))
        return hi[:n.value], lo[:n.value], gp[:n.value]

    def window_bits(self):
        """(fast, slow): the window-validity bitmap by the bit-parallel path and by the per-position definition"""
        cap = len(self._image) // 32 + 64 * 1024
        fast = np.zeros(cap, dtype=np.uint32)
        slow = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64()
        _check(lib().cora_hip_test_window_bits(self._h, fast.ctypes.data, slow.ctypes.data, cap, C.byref(n)))
        return fast[:n.value], slow[:n.value]

    def close(self):
        if self._h:
            lib().cora_hip_job_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sort_host(key_hi, key_lo=None, val=None, bit_lo=0, bit_hi=64):
    """Sorts numpy arrays in place through the device radix sort (test helper)."""
    n = len(key_hi)
    _check(lib().cora_hip_test_sort_host(key_hi.ctypes.data, key_lo.ctypes.data if key_lo is not None else None,
                                         val.ctypes.data if val is not None else None, n, bit_lo, bit_hi))


def sort_chain_host(key_hi, key_lo, bit_lo, bit_hi, chain):
    """The two-word sort with the second word climbing through `chain` arrays (test helper); in place."""
    _check(lib().cora_hip_test_sort_chain_host(key_hi.ctypes.data, key_lo.ctypes.data, len(key_hi), bit_lo, bit_hi, chain))


def sort_kept_host(key_hi, key_lo, val, bit_lo, bit_hi, owner_mode, world, me):
    """The sort of one rank of the few-GPU flow (test helper): of all records those owned by `me`, sorted; returns new arrays."""
    import numpy as np
    n = len(key_hi)
    out_hi = np.zeros(max(n, 1), dtype=np.uint64)
    out_lo = np.zeros(max(n, 1), dtype=np.uint64) if key_lo is not None else None
    out_val = np.zeros(max(n, 1), dtype=np.uint32) if val is not None else None
    kept = C.c_uint64()
    _check(lib().cora_hip_test_sort_kept_host(key_hi.ctypes.data, key_lo.ctypes.data if key_lo is not None else None,
                                              val.ctypes.data if val is not None else None, n, bit_lo, bit_hi, owner_mode, world, me,
                                              out_hi.ctypes.data, out_lo.ctypes.data if out_lo is not None else None,
                                              out_val.ctypes.data if out_val is not None else None, n, C.byref(kept)))
    k = kept.value
    return out_hi[:k], (out_lo[:k] if out_lo is not None else None), (out_val[:k] if out_val is not None else None)


def build_tables(fasta_path, K, d, phys_path, par_path, chr_short=False):
    """In-process equivalent of `homTable_setup ref K phys Exact Inexact par BOTH d FULL`."""
    job = Job(K, d, phys_path, par_path, chr_short)
    try:
        job.set_reference_file(fasta_path)
        job.run()
        e, en, i, inn = job.tables()
        return e, en, i, inn, job.stats()
    finally:
        job.close()
