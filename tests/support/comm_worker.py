"""One rank of the CPU rehearsal of cora_amd.dist.Comm (gloo, no GPU): tests/test_dist_cpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from cora_amd import dist as cdist

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)
comm = cdist.Comm(torch.device("cpu"))
assert comm.staged and not comm.local and comm.world == world

# route: records tagged (source rank, destination rank, serial); every destination must see its records in
# source-rank order and, inside one source, in serial order (what the stable sorts downstream rely on)
g = torch.Generator().manual_seed(100 + rank)
counts = [int(x) for x in torch.randint(0, 50, (world,), generator=g)]
if rank == 0:
    counts[-1] = 0                                  # an empty chunk
send = torch.cat([torch.tensor([(rank << 40) | (r << 20) | i for i in range(counts[r])], dtype=torch.int64) for r in range(world)])
recv_counts = comm.exchange_counts(counts)
for dt in (torch.int64, torch.int32, torch.uint8):
    got = comm.all_to_all(send.to(dt) if dt != torch.int64 else send, counts, recv_counts)
    assert got.dtype == dt and got.numel() == sum(recv_counts)
got = comm.all_to_all(send, counts, recv_counts)
at = 0
for src in range(world):
    chunk = got[at:at + recv_counts[src]]
    at += recv_counts[src]
    assert all(int(v) == ((src << 40) | (rank << 20) | i) for i, v in enumerate(chunk.tolist())), (rank, src)

# or of disjoint bitmaps through a sum
bits = torch.zeros(64, dtype=torch.int32)
bits[rank::world] = 1 << rank
tot = comm.all_reduce_sum(bits.clone())
assert all(int(tot[i]) == 1 << (i % world) for i in range(64))

# first singleton: min with "none" encoded as NO_SINGLETON
mine = cdist.NO_SINGLETON if rank != 1 else (3 << 32) | 77
assert comm.all_reduce_min_int(mine) == ((3 << 32) | 77 if world > 1 else mine)
assert comm.all_reduce_min_int(cdist.NO_SINGLETON) == cdist.NO_SINGLETON

# block slices of shard_begin: contiguous, disjoint, complete
n_blocks = 1001
cuts = [(n_blocks * r // world, n_blocks * (r + 1) // world) for r in range(world)]
assert cuts[0][0] == 0 and cuts[-1][1] == n_blocks and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
comm.barrier()
dist.destroy_process_group()
print("ok", rank)
