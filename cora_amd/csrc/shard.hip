// shard.hip -- multi-GPU flow: one process per GPU, every rank holds the packed reference, works on 1/R of
// the positions and owns 1/R of the key space.  The library does the per-rank compute; the caller moves the
// "slots" between ranks (torch.distributed all_to_all_single / all_reduce over RCCL: cora_amd/dist.py).
//
//   KMERS      own position slice -> hashed K-mer records, partitioned by hash range        [all-to-all]
//   EXACT      received records  -> sort, classes, later members of the owned classes        [all-reduce of that
//                                                                       position bitmap and of the first singleton]
//   PARTS      entry lists of the owned classes with >= 2 entries                            [gather to rank 0]
//   MERGE      rank 0: all classes by representative, block merge, <Exact> bytes
//   REPS       representatives inside the own position slice
//   SEED s     their seed-s codes, partitioned by code hash                                  [all-to-all]
//   VERIFY s   received codes -> sort, equal runs, verification against the local replica
//   ITEMS      pair records -> directed items                                                [gather to rank 0]
//   FINISH     rank 0: item sort, lists, block merge, final order, <Inexact> bytes
//
// Equal keys (codes) meet on one rank; inside each received chunk they are in increasing position and chunks
// arrive in rank = position order, so the stable sorts see them exactly as the single-GPU run does.
#include "job_internal.hpp"

using namespace cora;

struct ShardState {
    int rank = 0, world = 1;
    u32 blk_lo = 0, blk_n = 0;
    u64 word_lo = 0, word_hi = 0;
    DBuf<u64> send_a, recv_a;
    DBuf<u32> send_c, recv_c;
    u64 n_send = 0, n_recv = 0;
    DBuf<u32> repmask_g;
    ExactLocal x;
    ExactParts parts, rparts;
    DBuf<u32> reps;
    u64 n_reps = 0;
    PairStore ps;
    FinishStats fst;
    double host_ms = 0;
    std::chrono::steady_clock::time_point t_begin;
};

void shard_state_free(cora_hip_job *job)
{
    if (!job->shard) return;
    PoolScope scope(&job->pool);
    delete job->shard;
    job->shard = nullptr;
}

enum { SLOT_SEND_A = 0, SLOT_SEND_C, SLOT_RECV_A, SLOT_RECV_C, SLOT_REPMASK, SLOT_X_CLS, SLOT_X_CNT, SLOT_X_ENT, SLOT_X_DIR,
       SLOT_RX_CLS, SLOT_RX_CNT, SLOT_RX_ENT, SLOT_RX_DIR, SLOT_COUNT };
enum { ST_KMERS = 1, ST_EXACT, ST_PARTS, ST_MERGE, ST_REPS, ST_SEED, ST_VERIFY, ST_ITEMS, ST_FINISH };

// stable partition of (a[, c]) by destination rank; counts[r] = records for rank r
static int partition_by_dest(cora_hip_job *job, DBuf<u64> &a, DBuf<u32> *c, u64 n, int mode, int world, uint64_t *counts)
{
    hipStream_t s = job->stream;
    for (int r = 0; r < world; r++) counts[r] = 0;
    if (world == 1 || n == 0) { counts[0] = (world == 1) ? n : 0; return 0; }
    DBuf<u64> dest, tdest, ta;
    DBuf<u32> tc;
    DBuf<unsigned long long> dcnt;
    RCQ(dest.alloc(n)); RCQ(tdest.alloc(n)); RCQ(ta.alloc(n)); RCQ(dcnt.alloc(256));
    if (c) RCQ(tc.alloc(n));
    HIPQ(hipMemsetAsync(dcnt.p, 0, 256 * sizeof(unsigned long long), s));
    launch_dest_keys(s, a.p, n, mode, (u32)world, dest.p);
    launch_count_dest(s, dest.p, n, (u32)world, dcnt.p);
    SortArrays keys, tmp;
    keys.a = dest.p; keys.b = a.p; keys.c = c ? c->p : nullptr;
    tmp.a = tdest.p; tmp.b = ta.p; tmp.c = c ? tc.p : nullptr;
    if (radix_sort(s, keys, tmp, n, 64, 72, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
    unsigned long long h[256];
    HIPQ(hipMemcpyAsync(h, dcnt.p, sizeof h, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    if (keys.b != a.p) { std::swap(a.p, ta.p); if (c) std::swap(c->p, tc.p); }     // one pass: the result sits in the tmp arrays
    for (int r = 0; r < world; r++) counts[r] = h[r];
    return 0;
}

extern "C" int cora_hip_shard_begin(cora_hip_job *job, int rank, int world)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    if (world < 1 || world > 256 || rank < 0 || rank >= world) return fail(19, "bad rank %d of %d", rank, world);
    PoolScope scope(&job->pool);
    shard_state_free(job);
    ShardState *sh = new ShardState();
    job->shard = sh;
    sh->rank = rank; sh->world = world;
    const DevParams &dp = job->dp;
    const u64 n_blocks = (dp.G + 255) / 256;                 // G is a multiple of 64; a block covers 8 bitmap words
    const u64 b0 = n_blocks * rank / world, b1 = n_blocks * (rank + 1) / world;
    sh->blk_lo = (u32)b0; sh->blk_n = (u32)(b1 - b0);
    sh->word_lo = b0 * 8; sh->word_hi = std::min<u64>(b1 * 8, dp.n_words);
    memset(&job->stats, 0, sizeof job->stats);
    job->stats.genome_bases = job->genome_bases;
    job->sort_timing = SortTiming();
    job->sort_timing.enabled = true;
    sh->t_begin = std::chrono::steady_clock::now();
    RCQ(st_pack(job));
    // every contig must own at least one window (HTS:2530); no rank sees every window, so ask the validity bitmap
    launch_contig_scan(job->stream, dp, job->bad.p, job->ctr.p);
    DevCounters h;
    RCQ(read_counters(job, h));
    if (h.empty_contigs)
        return fail(54, "%llu contig(s) hold no findable %d-mer: the reference's behaviour is undefined there (HTS:2530)",
                    (unsigned long long)h.empty_contigs, dp.K);
    return 0;
}

extern "C" int cora_hip_shard_stage(cora_hip_job *job, int stage, int arg, uint64_t *counts)
{
    if (!job || !job->shard) return fail(19, "cora_hip_shard_begin has not run");
    PoolScope scope(&job->pool);
    ShardState &sh = *job->shard;
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    uint64_t local_counts[256];
    if (!counts) counts = local_counts;
    switch (stage) {
    case ST_KMERS: {
        RCQ(st_kmers(job, sh.blk_lo, sh.blk_n, sh.send_a, sh.send_c, sh.n_send));
        RCQ(partition_by_dest(job, sh.send_a, &sh.send_c, sh.n_send, 0, sh.world, counts));
        return 0;
    }
    case ST_EXACT: {
        sh.x.release_all();
        sh.x.N = sh.n_recv;
        std::swap(sh.x.rec.p, sh.recv_a.p); std::swap(sh.x.rec.n, sh.recv_a.n); std::swap(sh.x.rec.pool, sh.recv_a.pool);
        std::swap(sh.x.gp.p, sh.recv_c.p); std::swap(sh.x.gp.n, sh.recv_c.n); std::swap(sh.x.gp.pool, sh.recv_c.pool);
        if (!sh.x.rec.p) { RCQ(sh.x.rec.alloc(1)); RCQ(sh.x.gp.alloc(1)); }
        sh.send_a.release(); sh.send_c.release();
        job->stats.kmers = sh.x.N;
        RCQ(st_exact_classes(job, sh.x));
        job->stats.representatives = sh.x.NR; job->stats.exact_classes = sh.x.NNS;
        return 0;
    }
    case ST_PARTS: {
        RCQ(st_exact_parts(job, sh.x, sh.parts));
        sh.x.release_all();
        return 0;
    }
    case ST_MERGE: {
        RCQ(st_exact_finish(job, sh.rparts, sh.fst, sh.host_ms));
        sh.rparts.release_all(); sh.parts.release_all();
        return 0;
    }
    case ST_REPS: {
        if (!sh.repmask_g.p) return fail(19, "the global representative bitmap has not been imported");
        RCQ(st_reps(job, sh.repmask_g.p, sh.word_lo, sh.word_hi, sh.reps, sh.n_reps));
        return 0;
    }
    case ST_SEED: {
        RCQ(sh.send_a.alloc(sh.n_reps));
        sh.n_send = sh.n_reps;
        launch_seed_codes(s, dp, job->pk.p, sh.reps.p, sh.n_reps, arg, sh.send_a.p);
        HIPQ(hipStreamSynchronize(s));
        RCQ(partition_by_dest(job, sh.send_a, nullptr, sh.n_send, 1, sh.world, counts));
        return 0;
    }
    case ST_VERIFY: {
        if (sh.n_recv) RCQ(st_seed_verify(job, sh.recv_a, sh.n_recv, arg, sh.ps));
        sh.recv_a.release(); sh.send_a.release();
        job->stats.hom_pairs = sh.ps.n;
        return 0;
    }
    case ST_ITEMS: {
        const u64 M = 2 * sh.ps.n;
        RCQ(sh.send_a.alloc(M)); RCQ(sh.send_c.alloc(M));
        launch_items_expand(s, dp, sh.ps.pairs.p, sh.ps.n, sh.send_a.p, sh.send_c.p);
        HIPQ(hipStreamSynchronize(s));
        sh.ps.pairs.release(); sh.ps.n = sh.ps.cap = 0;
        sh.reps.release(); sh.repmask_g.release();
        sh.n_send = M;
        for (int r = 0; r < sh.world; r++) counts[r] = 0;
        counts[0] = M;                                       // every item goes to rank 0
        return 0;
    }
    case ST_FINISH: {
        if (!sh.recv_a.p) { RCQ(sh.recv_a.alloc(1)); RCQ(sh.recv_c.alloc(1)); }
        RCQ(st_items_finish(job, sh.recv_a, sh.recv_c, sh.n_recv, sh.fst, sh.host_ms));
        sh.recv_a.release(); sh.recv_c.release(); sh.send_a.release(); sh.send_c.release();
        cora_hip_stats &st = job->stats;
        st.exact_nodes = sh.fst.exact_nodes; st.inexact_nodes = sh.fst.inexact_nodes;
        st.dirty_nodes = sh.fst.dirty_nodes; st.big_nodes = sh.fst.big_nodes;
        st.ms_host_finish = (float)sh.host_ms;
        st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t_begin).count();
        st.ms_sort_scatter = job->sort_timing.scatter_ms;
        st.n_sort_scatter = job->sort_timing.scatter_launches;
        st.sort_scatter_bytes = job->sort_timing.scatter_bytes;
        return 0;
    }
    default:
        return fail(19, "unknown shard stage %d", stage);
    }
}

static int slot_ref(ShardState &sh, const DevParams &dp, int slot, void **p, u64 *n, u32 *eb)
{
    switch (slot) {
    case SLOT_SEND_A: *p = sh.send_a.p; *n = sh.n_send; *eb = 8; return 0;
    case SLOT_SEND_C: *p = sh.send_c.p; *n = sh.n_send; *eb = 4; return 0;
    case SLOT_RECV_A: *p = sh.recv_a.p; *n = sh.n_recv; *eb = 8; return 0;
    case SLOT_RECV_C: *p = sh.recv_c.p; *n = sh.n_recv; *eb = 4; return 0;
    case SLOT_REPMASK: *p = sh.x.nonrep.p; *n = dp.n_words; *eb = 4; return 0;
    case SLOT_X_CLS: *p = sh.parts.cls.p; *n = sh.parts.NNS; *eb = 8; return 0;
    case SLOT_X_CNT: *p = sh.parts.ce.p; *n = sh.parts.NNS; *eb = 4; return 0;
    case SLOT_X_ENT: *p = sh.parts.ent.p; *n = sh.parts.NE; *eb = 4; return 0;
    case SLOT_X_DIR: *p = sh.parts.edir.p; *n = sh.parts.NE; *eb = 1; return 0;
    default: return 1;
    }
}

extern "C" int cora_hip_shard_slot_info(cora_hip_job *job, int slot, uint64_t *n_elems, uint32_t *elem_bytes)
{
    if (!job || !job->shard) return fail(19, "cora_hip_shard_begin has not run");
    void *p; u64 n; u32 eb;
    if (slot_ref(*job->shard, job->dp, slot, &p, &n, &eb)) return fail(19, "slot %d cannot be exported", slot);
    if (n_elems) *n_elems = n;
    if (elem_bytes) *elem_bytes = eb;
    return 0;
}

extern "C" int cora_hip_shard_slot_export(cora_hip_job *job, int slot, void *dst_device)
{
    if (!job || !job->shard) return fail(19, "cora_hip_shard_begin has not run");
    void *p; u64 n; u32 eb;
    if (slot_ref(*job->shard, job->dp, slot, &p, &n, &eb)) return fail(19, "slot %d cannot be exported", slot);
    if (n && (!p || !dst_device)) return fail(19, "slot %d is empty or the destination is null", slot);
    if (n) HIPQ(hipMemcpyAsync(dst_device, p, n * eb, hipMemcpyDeviceToDevice, job->stream));
    HIPQ(hipStreamSynchronize(job->stream));
    return 0;
}

extern "C" int cora_hip_shard_slot_import(cora_hip_job *job, int slot, const void *src_device, uint64_t n)
{
    if (!job || !job->shard) return fail(19, "cora_hip_shard_begin has not run");
    PoolScope scope(&job->pool);
    ShardState &sh = *job->shard;
    hipStream_t s = job->stream;
    void *dst = nullptr;
    u32 eb = 0;
    switch (slot) {
    case SLOT_RECV_A: RCQ(sh.recv_a.alloc(n)); sh.n_recv = n; dst = sh.recv_a.p; eb = 8; break;
    case SLOT_RECV_C: RCQ(sh.recv_c.alloc(n)); if (n != sh.n_recv) return fail(19, "RECV_C must match RECV_A"); dst = sh.recv_c.p; eb = 4; break;
    case SLOT_REPMASK:
        if (n != job->dp.n_words) return fail(19, "bitmap size mismatch");
        RCQ(sh.repmask_g.alloc(n)); dst = sh.repmask_g.p; eb = 4; break;
    case SLOT_RX_CLS: RCQ(sh.rparts.cls.alloc(n)); sh.rparts.NNS = n; dst = sh.rparts.cls.p; eb = 8; break;
    case SLOT_RX_CNT: RCQ(sh.rparts.ce.alloc(n)); if (n != sh.rparts.NNS) return fail(19, "RX_CNT must match RX_CLS"); dst = sh.rparts.ce.p; eb = 4; break;
    case SLOT_RX_ENT: RCQ(sh.rparts.ent.alloc(n)); sh.rparts.NE = n; dst = sh.rparts.ent.p; eb = 4; break;
    case SLOT_RX_DIR: RCQ(sh.rparts.edir.alloc(n)); if (n != sh.rparts.NE) return fail(19, "RX_DIR must match RX_ENT"); dst = sh.rparts.edir.p; eb = 1; break;
    default: return fail(19, "slot %d cannot be imported", slot);
    }
    if (n) {
        if (!src_device) return fail(19, "null source");
        HIPQ(hipMemcpyAsync(dst, src_device, n * eb, hipMemcpyDeviceToDevice, s));
    }
    if (slot == SLOT_REPMASK)       // what arrives is the union of all ranks' non-representatives: turn it into the representative bitmap
        launch_rep_mask(s, job->dp, job->bad.p, sh.repmask_g.p, sh.repmask_g.p);
    HIPQ(hipStreamSynchronize(s));
    return 0;
}

// first singleton of this rank's classes in the reference's class-id order (Q3); ~0 = none
extern "C" int cora_hip_shard_get_first_singleton(cora_hip_job *job, uint64_t *key)
{
    if (!job || !job->shard || !key) return fail(19, "bad argument");
    DevCounters h;
    RCQ(read_counters(job, h));
    *key = h.first_singleton_key;
    return 0;
}

extern "C" int cora_hip_shard_set_first_singleton(cora_hip_job *job, uint64_t key)
{
    if (!job || !job->shard) return fail(19, "bad argument");
    DevCounters h;
    RCQ(read_counters(job, h));
    h.first_singleton_key = key;
    RCQ(write_counters(job, h));
    return 0;
}
