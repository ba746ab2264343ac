"""Multi-GPU table build: one process per GPU, ``torch.distributed`` moves the records.

The per-rank compute is libcora_hip.so's sharded stages (cora_amd/csrc/shard.hip); this module is the
exchange plumbing between them: ``all_to_all_single`` of K-mer records by key-hash range and of seed records
by code hash (above 4 ranks; up to 4 every rank scans all positions itself and keeps the records it owns, because a
GPU pair has one xGMI link: ``cora_hip_shard_filter``), of directed items by source interval, a sum ``all_reduce`` of the position bitmap of later class
members (bits are disjoint, so sum = or) and of per-position item counts near the interval starts, a min
``all_reduce`` of the first-singleton key, and gathers to rank 0 for the two final tables.  With the
``nccl`` backend (= RCCL over xGMI) tensors stay in HBM; with ``gloo`` (CPU rehearsal of the same code
path, e.g. two ranks sharing one GPU) they are staged through host memory.

The reference has no counterpart: it is a single process.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import hip

ST_KMERS, ST_EXACT, ST_PARTS, ST_MERGE, ST_REPS, ST_SEED, ST_VERIFY, ST_ITEMS, ST_FINISH, ST_ITEM_CUT, ST_COLLECT = range(1, 12)
(SLOT_SEND_A, SLOT_SEND_C, SLOT_RECV_A, SLOT_RECV_C, SLOT_REPMASK, SLOT_X_CLS, SLOT_X_CNT, SLOT_X_ENT, SLOT_X_DIR,
 SLOT_RX_CLS, SLOT_RX_CNT, SLOT_RX_ENT, SLOT_RX_DIR, SLOT_WCOUNT, SLOT_F2, SLOT_RX_F2, SLOT_STATS, SLOT_SEND_B, SLOT_RECV_B) = range(19)

_DTYPE = {8: torch.int64, 4: torch.int32, 1: torch.uint8}
_TYPESTR = {8: "<i8", 4: "<i4", 1: "|u1"}


class _DeviceArray:
    """A slot of the library seen through ``__cuda_array_interface__``: torch wraps it without copying."""

    def __init__(self, ptr, n, elem_bytes):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": _TYPESTR[elem_bytes], "data": (ptr, False), "version": 2}
NO_SINGLETON = (1 << 62)        # exchanged instead of ~0 (int64 min-reduce)


class Comm:
    """The four collectives the flow needs, on device tensors, over whatever backend the group has."""

    def __init__(self, device, always_collective=False):
        self.device = device
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # world 1 skips the collectives unless asked not to (a test drives RCCL with a single rank that way)
        self.local = self.world == 1 and not (always_collective and dist.is_initialized())
        self.staged = dist.is_initialized() and dist.get_backend() != "nccl"      # gloo: through host memory

    def _w(self, t):
        return t.cpu() if self.staged else t

    def exchange_counts(self, counts):
        """counts[r] = what I send to rank r  ->  what every rank sends to me."""
        if self.local:
            return list(counts)
        send = self._w(torch.tensor(counts, dtype=torch.int64, device=self.device))
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        return [int(x) for x in recv.tolist()]

    def all_to_all_into(self, recv, send, send_counts, recv_counts):
        """Device tensors only (nccl): the records land in `recv`, which the caller owns."""
        dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))

    def all_to_all(self, send, send_counts, recv_counts):
        if self.local:
            return send
        recv = torch.empty(sum(recv_counts), dtype=send.dtype, device="cpu" if self.staged else self.device)
        dist.all_to_all_single(recv, self._w(send), output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
        return recv.to(self.device) if self.staged else recv

    def all_reduce_sum(self, t):
        if self.local or t.numel() == 0:
            return t
        w = self._w(t)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        return w.to(self.device) if self.staged else w

    def all_reduce_min_int(self, v):
        if self.local:
            return v
        w = self._w(torch.tensor([v], dtype=torch.int64, device=self.device))
        dist.all_reduce(w, op=dist.ReduceOp.MIN)
        return int(w.item())

    def barrier(self):
        if not self.local:
            dist.barrier()


class ShardedJob:
    """Drives one sharded table build; every rank calls run(); rank 0 ends up with the tables."""

    def __init__(self, job: hip.Job, comm: Comm, filter=None):
        self.job, self.comm = job, comm
        self.filter = filter          # None: the library's choice (world <= 4, CORA_SHARD_FILTER); True / False: forced
        self.kept_own = False         # what the last run did
        self.report = None            # ... and where its time went (run())
        self.lib = hip.lib()
        self.h = job._h

    # ---- slots <-> torch tensors
    def _view(self, slot):
        """The slot's records as a tensor over the library's own memory (valid until the next stage / import)."""
        n, eb, ptr = C.c_uint64(), C.c_uint32(), C.c_void_p()
        hip._check(self.lib.cora_hip_shard_slot_info(self.h, slot, C.byref(n), C.byref(eb)))
        hip._check(self.lib.cora_hip_shard_slot_ptr(self.h, slot, C.byref(ptr)))
        if n.value == 0 or not ptr.value:
            return torch.empty(0, dtype=_DTYPE[eb.value], device=self.comm.device)
        return torch.as_tensor(_DeviceArray(ptr.value, n.value, eb.value), device=self.comm.device)

    def _export(self, slot):
        n, eb = C.c_uint64(), C.c_uint32()
        hip._check(self.lib.cora_hip_shard_slot_info(self.h, slot, C.byref(n), C.byref(eb)))
        t = torch.empty(n.value, dtype=_DTYPE[eb.value], device=self.comm.device)
        hip._check(self.lib.cora_hip_shard_slot_export(self.h, slot, C.c_void_p(t.data_ptr() if n.value else 0)))
        return t

    def _import(self, slot, t):
        t = t.contiguous()
        torch.cuda.synchronize()
        hip._check(self.lib.cora_hip_shard_slot_import(self.h, slot, C.c_void_p(t.data_ptr() if t.numel() else 0), t.numel()))

    def _stage(self, stage, arg=0):
        counts = (C.c_uint64 * 256)()
        hip._check(self.lib.cora_hip_shard_stage(self.h, stage, arg, counts))
        return [int(counts[r]) for r in range(self.comm.world)]

    def _reserve(self, slot, n, like):
        """The receive slot sized for n elements, as a tensor over the library's memory."""
        ptr = C.c_void_p()
        hip._check(self.lib.cora_hip_shard_slot_reserve(self.h, slot, n, C.byref(ptr)))
        if n == 0 or not ptr.value:
            return torch.empty(0, dtype=like.dtype, device=self.comm.device)
        return torch.as_tensor(_DeviceArray(ptr.value, n, like.element_size()), device=self.comm.device)

    def _move(self, src, dst, counts, recv_counts):
        send = self._view(src)
        if self.comm.local or self.comm.staged:
            self._import(dst, self.comm.all_to_all(send, counts, recv_counts))
            return
        # RCCL: straight from the library's send slot into its receive slot, no tensor in between
        recv = self._reserve(dst, sum(recv_counts), send)
        self.comm.all_to_all_into(recv, send, counts, recv_counts)
        torch.cuda.synchronize()

    def _route(self, pairs, counts):
        """pairs: [(send slot, recv slot)]; counts[r] records of each send slot go to rank r."""
        recv_counts = self.comm.exchange_counts(counts)
        for src, dst in pairs:
            self._move(src, dst, counts, recv_counts)

    def _gather_to_root(self, pairs):
        """Each (send slot, recv slot) may have its own length: everything goes to rank 0."""
        for src, dst in pairs:
            n = self._view(src).numel()
            counts = [n] + [0] * (self.comm.world - 1)
            self._move(src, dst, counts, self.comm.exchange_counts(counts))

    def run(self):
        """One sharded build.  Leaves in self.report what this rank spent where (wall ms per part of the flow, the waits for slower
        ranks inside the collectives included) and the bytes it sent to OTHER ranks per exchange: bench.py prints them per rank."""
        import time
        c, lib, h = self.comm, self.lib, self.h
        ms, sent = {}, {}
        t_last = [time.perf_counter()]

        def lap(name):
            now = time.perf_counter()
            ms[name] = ms.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

        def route(name, pairs, counts):
            recv_counts = self.comm.exchange_counts(counts)
            away = sum(n for r, n in enumerate(counts) if r != c.rank)
            for src, dst in pairs:
                sent[name] = sent.get(name, 0) + away * self._elem_bytes(src)
                self._move(src, dst, counts, recv_counts)

        torch.cuda.synchronize()
        hip._check(lib.cora_hip_shard_begin(h, c.rank, c.world))
        # K-mer and seed records reach their owner by all-to-all, or -- 2 to 4 ranks, where a GPU pair has one xGMI link
        # (shard.hip) -- every rank scans all positions and keeps its own, and nothing is moved.  The ranks must agree.
        on = C.c_int()
        hip._check(lib.cora_hip_shard_filter(h, -1 if self.filter is None else int(self.filter), C.byref(on)))
        keep_own = bool(c.all_reduce_min_int(on.value))
        if keep_own != bool(on.value):
            hip._check(lib.cora_hip_shard_filter(h, int(keep_own), C.byref(on)))
        self.kept_own = keep_own
        lap("pack")
        # exact: K-mer records to the owner of their hash range
        counts = self._stage(ST_KMERS)
        lap("kmer_records")
        if not keep_own:
            route("kmer_records", [(SLOT_SEND_A, SLOT_RECV_A), (SLOT_SEND_C, SLOT_RECV_C)], counts)
            lap("kmer_exchange")
        self._stage(ST_EXACT)
        lap("exact_classes")
        key = C.c_uint64()
        hip._check(lib.cora_hip_shard_get_first_singleton(h, C.byref(key)))
        mine = NO_SINGLETON if key.value == 0xFFFFFFFFFFFFFFFF else key.value
        glob = c.all_reduce_min_int(mine)
        hip._check(lib.cora_hip_shard_set_first_singleton(h, 0xFFFFFFFFFFFFFFFF if glob == NO_SINGLETON else glob))
        mask = self._export(SLOT_REPMASK)
        if not c.local:
            sent["bitmap_allreduce"] = 2 * mask.numel() * mask.element_size() * (c.world - 1) // c.world     # a ring moves that much per rank
        self._import(SLOT_REPMASK, c.all_reduce_sum(mask))      # disjoint bits: sum = or
        del mask
        lap("bitmap_allreduce")
        self._stage(ST_PARTS)
        if c.rank != 0:
            sent["class_parts"] = sum(self._slot_bytes(sl) for sl in (SLOT_X_CLS, SLOT_X_CNT, SLOT_X_ENT, SLOT_X_DIR))
        self._gather_to_root([(SLOT_X_CLS, SLOT_RX_CLS), (SLOT_X_CNT, SLOT_RX_CNT), (SLOT_X_ENT, SLOT_RX_ENT),
                              (SLOT_X_DIR, SLOT_RX_DIR)])
        lap("class_parts_to_rank0")
        if c.rank == 0:
            self._stage(ST_MERGE)
            lap("exact_table_rank0")
        # inexact: seed records to the owner of their code hash, verification there
        self._stage(ST_REPS)
        lap("representatives")
        for seed in range(1, self.job.d + 2):
            counts = self._stage(ST_SEED, seed)
            lap("seed_records")
            pairs = [(SLOT_SEND_A, SLOT_RECV_A)]
            on = C.c_int()
            hip._check(lib.cora_hip_shard_seed_sketch(h, C.byref(on)))
            if on.value:                                  # large references: the records' sketches travel along (same answer on every rank)
                pairs.append((SLOT_SEND_B, SLOT_RECV_B))
            if not keep_own:
                route("seed_records", pairs, counts)
                lap("seed_exchange")
            self._stage(ST_VERIFY, seed)
            lap("seed_sort_verify")
        # lists are finished by the owner of their source position; intervals are cut where a position holds no list
        self._stage(ST_ITEMS)
        if not c.local:
            self._import(SLOT_WCOUNT, c.all_reduce_sum(self._export(SLOT_WCOUNT)))
        lap("items")
        route("items", [(SLOT_SEND_A, SLOT_RECV_A), (SLOT_SEND_C, SLOT_RECV_C)], self._stage(ST_ITEM_CUT))
        lap("item_exchange")
        self._stage(ST_FINISH)
        lap("lists_and_own_table_bytes")
        if c.rank != 0:
            sent["table_bytes"] = self._slot_bytes(SLOT_F2)
        self._gather_to_root([(SLOT_F2, SLOT_RX_F2)])
        self._import(SLOT_STATS, c.all_reduce_sum(self._export(SLOT_STATS)))
        if c.rank == 0:
            self._stage(ST_COLLECT)
        torch.cuda.synchronize()
        lap("table_bytes_to_rank0")
        self.report = {"rank": c.rank, "kept_own": keep_own, "ms": {k: round(v, 3) for k, v in ms.items()}, "sent_bytes": sent}

    def _elem_bytes(self, slot):
        n, eb = C.c_uint64(), C.c_uint32()
        hip._check(self.lib.cora_hip_shard_slot_info(self.h, slot, C.byref(n), C.byref(eb)))
        return int(eb.value)

    def _slot_bytes(self, slot):
        n, eb = C.c_uint64(), C.c_uint32()
        hip._check(self.lib.cora_hip_shard_slot_info(self.h, slot, C.byref(n), C.byref(eb)))
        return int(n.value) * int(eb.value)


# ---------------------------------------------------------------------------------------------------------------
# The read side over several GPUs (fastqSplit -> collapse in its fragmented mode).
# Fragment groups are independent: rank r takes the groups g with g % world == r, and no read-mer ever crosses a link.
# The only thing that chains the groups is the numbering of long read names (cora.cpp:1762-1776 counts the lines of every
# group's list before it starts the next group); which names are long does not depend on the numbers, so every rank first
# runs its groups numbered from 0, the counts are summed over the ranks (one small all-reduce), and only the groups that
# hold long names AND start at another number are run again.
# ---------------------------------------------------------------------------------------------------------------
def lrn_offsets(first, counts):
    """counts[g] = long read names of group g  ->  the number each group's names start after (cora.cpp:1776)."""
    out, at = [], first
    for n in counts:
        out.append(at)
        at += n
    return out


def read_chain_sharded(comm, signals_text, read_len, mode, fastq_images, reference_image, digit_len, mapper="GENERIC",
                       chr_short=False, lrn_offset=0):
    """Every rank calls this with the same inputs; returns {group: [reads, perfect, lrn] bytes} for the groups this rank
    owns.  Group g's files are the bytes the reference's chain writes as <file>.split<g>."""
    from . import collapse as ccol
    from . import fastqsplit
    fs = fastqsplit.FastqSplit(signals_text, read_len, mode)
    for img in fastq_images:
        fs.add_reads(img)
    fs.set_device_only(True)
    fs.run()
    splits, paired = fastqsplit.MODES[mode]
    text = signals_text.decode()
    size = int(text.split()[1])
    lines = text.split("\n")[1:1 + fs.n_groups]
    col = ccol.Collapse(digit_len, read_len, "PAIRED" if paired else "SINGLE", {1: "FULL", 2: "HALF", 3: "THREEWAY"}[splits], mapper,
                        chr_short, lrn_offset)
    if reference_image is not None:
        col.set_reference(reference_image)
    own = [g for g in range(fs.n_groups) if g % comm.world == comm.rank]

    def run_group(g, first):
        col.set_fragment(lines[g], size)
        col.set_lrn_offset(first)
        col.add_reads_device(*fs.group_device(g))
        col.run()
        return col.outputs(), col.stats()["lrn_names"]

    outs, counts = {}, torch.zeros(fs.n_groups, dtype=torch.int64, device=comm.device)
    for g in own:
        outs[g], n = run_group(g, lrn_offset)
        counts[g] = n
    counts = [int(x) for x in comm.all_reduce_sum(counts).tolist()]
    first = lrn_offsets(lrn_offset, counts)
    for g in own:
        if counts[g] and first[g] != lrn_offset:
            outs[g], _ = run_group(g, first[g])
    col.close()
    fs.close()
    return outs
