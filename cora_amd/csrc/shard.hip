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
//   ITEMS      pair records -> directed items; items per source position near the slice starts [all-reduce]
//   ITEM_CUT   owner intervals cut where a position holds no list; items partitioned by owner  [all-to-all]
//   FINISH     every rank: item sort, lists, block merge, final order, <Inexact> bytes of its interval
//                                                                                  [gather of the bytes to rank 0, sum of the counts]
//   COLLECT    rank 0: the ranks' byte ranges, in rank = position order, are the table
//
// Equal keys (codes) meet on one rank; inside each received chunk they are in increasing position and chunks
// arrive in rank = position order, so the stable sorts see them exactly as the single-GPU run does.
//
// Two ways to bring a rank its K-mer and seed records (`filter` below; the same tables either way):
//   exchange   as above: 1/R of the positions scanned, the records partitioned by owner and moved by all-to-all.  28 B per k-mer
//              and 16 B per representative and seed cross the links: fine over 7 links at once, link-bound on 2 or 4 GPUs,
//              where a GPU pair has ONE xGMI link between them.
//   filter     every rank holds the whole packed reference anyway: it scans ALL positions itself and keeps the records it
//              owns -- in the first pass of the sort they are generated for, which reads all records and writes its own
//              (radix_sort's SortSource).  Nothing crosses a link for KMERS and SEED and there is no partition pass; the price
//              is a scan that does not shrink with R.  Chosen for world <= 4 (CORA_SHARD_FILTER=0/1 overrides).
#include "job_internal.hpp"

using namespace cora;

// Rank 0's finish of the exact table (ST_MERGE: all classes by representative, block merge, <Exact> bytes; 9-11 ms of small,
// latency-bound kernels at 3.09 Gbp) runs BESIDE the stages that follow instead of holding rank 0 back while the other ranks
// go on to their seeds and then wait for it at the next exchange: a helper thread with a context of its own -- stream,
// counters, sort scratch, device pool, host buffers; it shares the device arrays of the reference -- that is joined before
// rank 0 finishes its lists.  (Round-2 review: "remove the rank-0 funnel".  Splitting the merge by representative range would
// need one more exchange of the class parts with cuts chosen where no block crosses; hiding it costs no exchange.)
struct MergeSide {
    cora_hip_job *ctx = nullptr;
    std::thread th;
    int rc = 0;
    std::string err;
    FinishStats fst;
    double host_ms = 0;
};

struct ShardState {
    int rank = 0, world = 1;
    bool filter = false;                // K-mer and seed records: scan everything and keep the own ones, instead of the exchange
    u32 blk_lo = 0, blk_n = 0;
    u64 word_lo = 0, word_hi = 0;
    DBuf<u64> send_a, recv_a, send_b, recv_b;       // b: the seed records' sketches (large references), moved like a
    DBuf<u32> send_c, recv_c;
    bool seed_sketch = false;
    u64 n_send = 0, n_recv = 0;
    DBuf<u32> repmask_g;
    ExactLocal x;
    ExactParts parts, rparts;
    DBuf<u32> reps;
    u64 n_reps = 0;
    PairStore ps;
    DBuf<u32> wcount, bnd_dev;          // items per source position in a window behind every nominal slice start
    std::vector<u32> bnd;               // nominal slice starts of ranks 1..world-1 (positions)
    u32 win = 0;
    DBuf<u8> f2_dev, f2_all;            // this rank's <Inexact> bytes / all ranks' bytes (rank 0)
    u64 n_f2 = 0, n_f2_all = 0;
    DBuf<u64> stat_dev;                 // counts summed over ranks
    FinishStats fst;
    double host_ms = 0;
    std::chrono::steady_clock::time_point t_begin;
    MergeSide *side = nullptr;          // rank 0: the exact table's finish beside the seeds stage
    bool host_gather = false;           // in-process ranks: <Inexact> bytes stay in the host buffer (shard_set_host_gather)
    int b_chain = 1;                    // arrays the seed sketches' second word climbs through during its sort (radix_sort.hip)
};

// arrays of the chain the seed sketches' second word climbs through during a rank's sort: three when the rank cuts its arrays from an
// arena (large references), else the plain ping-pong; CORA_B_CHAIN=n forces n (tests run the chained sorts on small references)
static int shard_b_chain(const cora_hip_job *job)
{
    const int env = getenv("CORA_B_CHAIN") ? atoi(getenv("CORA_B_CHAIN")) : 0;
    if (env >= 1 && env <= 5) return env;
    return job->pool.arena ? 3 : 1;
}

void shard_set_host_gather(cora_hip_job *job, bool on)
{
    if (job->shard) job->shard->host_gather = on;
}

// waits for the exact table's finish (if one is running), takes its result over; the class parts are released by the caller's thread
static int join_merge(cora_hip_job *job, ShardState &sh)
{
    if (!sh.side || !sh.side->th.joinable()) return 0;
    MergeSide &m = *sh.side;
    m.th.join();
    sh.rparts.release_all(); sh.parts.release_all();
    if (m.rc) { g_err = m.err; return m.rc; }
    ByteBuf &a = job->exact_bytes, &b = m.ctx->exact_bytes;
    std::swap(a.p, b.p); std::swap(a.n, b.n); std::swap(a.cap, b.cap); std::swap(a.mapped, b.mapped);
    sh.fst.exact_nodes = m.fst.exact_nodes;
    sh.host_ms += m.host_ms;
    return 0;
}

void shard_state_free(cora_hip_job *job)
{
    if (!job->shard) return;
    PoolScope scope(&job->pool);
    if (job->shard->side) {
        if (job->shard->side->th.joinable()) job->shard->side->th.join();
        job->shard->rparts.release_all(); job->shard->parts.release_all();
        if (job->shard->side->ctx) { PoolScope none(nullptr); cora_hip_job_destroy(job->shard->side->ctx); }
        delete job->shard->side;
    }
    delete job->shard;
    job->shard = nullptr;
}

enum { SLOT_SEND_A = 0, SLOT_SEND_C, SLOT_RECV_A, SLOT_RECV_C, SLOT_REPMASK, SLOT_X_CLS, SLOT_X_CNT, SLOT_X_ENT, SLOT_X_DIR,
       SLOT_RX_CLS, SLOT_RX_CNT, SLOT_RX_ENT, SLOT_RX_DIR, SLOT_WCOUNT, SLOT_F2, SLOT_RX_F2, SLOT_STATS, SLOT_SEND_B, SLOT_RECV_B, SLOT_COUNT };
enum { ST_KMERS = 1, ST_EXACT, ST_PARTS, ST_MERGE, ST_REPS, ST_SEED, ST_VERIFY, ST_ITEMS, ST_FINISH, ST_ITEM_CUT, ST_COLLECT };
constexpr u32 N_STATS = 8;              // inexact_nodes, dirty, big, inexact_items, hom_pairs, candidate_pairs, 2 spare

template <typename T>
static void swap_buf(DBuf<T> &x, DBuf<T> &y)
{
    std::swap(x.p, y.p); std::swap(x.n, y.n); std::swap(x.pool, y.pool);
}

// stable partition of (a[, c][, b2]) by destination rank, all arrays in one scatter pass (radix_partition); counts[r] =
// records for rank r.  mode 0: the top bits of a as they are (hash ranges), 1: hashed (seed codes).
static int partition_by_dest(cora_hip_job *job, DBuf<u64> &a, DBuf<u32> *c, u64 n, int mode, int world, uint64_t *counts, DBuf<u64> *b2 = nullptr)
{
    hipStream_t s = job->stream;
    for (int r = 0; r < world; r++) counts[r] = 0;
    if (world == 1 || n == 0) { counts[0] = (world == 1) ? n : 0; return 0; }
    DBuf<u64> ta, tb;
    DBuf<u32> tc;
    RCQ(ta.alloc(n));
    if (c) RCQ(tc.alloc(n));
    if (b2) RCQ(tb.alloc(n));
    SortArrays keys, tmp;
    keys.a = a.p; keys.b = b2 ? b2->p : nullptr; keys.c = c ? c->p : nullptr;
    tmp.a = ta.p; tmp.b = b2 ? tb.p : nullptr; tmp.c = c ? tc.p : nullptr;
    if (radix_partition(s, keys, tmp, n, mode == 0 ? 1 : 2, (u32)world, job->sort_scratch, &job->sort_timing, counts)) return fail(100, "partition failed");
    HIPQ(hipGetLastError());
    // one pass: the result sits in what were the spare arrays
    swap_buf(a, ta);
    if (c) swap_buf(*c, tc);
    if (b2) swap_buf(*b2, tb);
    return 0;
}

// filter mode: what the KMERS / SEED stage generated for ALL positions stays in the send arrays; the owner's sort reads it
// from there and keeps its own records in its first pass (radix_sort's SortSource)
static SortSource source_of(const ShardState &sh, int fmode, bool with_b, bool with_c)
{
    SortSource f;
    f.src.a = sh.send_a.p; f.src.b = with_b ? sh.send_b.p : nullptr; f.src.c = with_c ? sh.send_c.p : nullptr;
    f.n = sh.n_send; f.fmode = fmode; f.world = (u32)sh.world; f.me = (u32)sh.rank;
    return f;
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
    // one xGMI link per GPU pair: on 2 or 4 GPUs the all-to-all of the records costs more than scanning everything locally
    sh->filter = world > 1 && world <= 4;
    if (const char *e = getenv("CORA_SHARD_FILTER")) sh->filter = world > 1 && atoi(e) != 0;
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
    // a large reference: this rank's stage arrays are cut from one allocation, as in the single-GPU run (DevPool, job.hip)
    reserve_job_arena(job, world, sh->filter);
    job->pool.ranges.big_from_high = false;
    RCQ(st_pack(job));
    // every contig must own at least one window (HTS:2530); no rank sees every window, so ask the validity bitmap
    launch_contig_scan(job->stream, dp, job->bad.p, job->ctr.p);
    DevCounters h;
    RCQ(read_counters(job, h));
    FORMAT_ERROR_CHECK(h);
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
        if (sh.filter) {                // all positions; EXACT's sort keeps this rank's records
            RCQ(st_kmers(job, 0, (u32)((dp.G + 255) / 256), sh.send_a, sh.send_c, sh.n_send));
            for (int r = 0; r < sh.world; r++) counts[r] = 0;
            return 0;
        }
        RCQ(st_kmers(job, sh.blk_lo, sh.blk_n, sh.send_a, sh.send_c, sh.n_send));
        RCQ(partition_by_dest(job, sh.send_a, &sh.send_c, sh.n_send, 0, sh.world, counts));
        return 0;
    }
    case ST_EXACT: {
        sh.x.release_all();
        if (sh.filter) {
            const SortSource from = source_of(sh, 1, false, true);
            u64 kept = 0;
            if (radix_sort_count_kept(s, from, 64 - exact_sort_bits(job), 64, job->sort_scratch, &kept)) return fail(100, "radix sort failed");
            sh.x.N = kept;
            RCQ(sh.x.rec.alloc(kept)); RCQ(sh.x.gp.alloc(kept));
            job->stats.kmers = kept;
            RCQ(st_exact_classes(job, sh.x, &from));
            sh.send_a.release(); sh.send_c.release(); sh.n_send = 0;
            job->stats.representatives = sh.x.NR; job->stats.exact_classes = sh.x.NNS;
            return 0;
        }
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
        if (getenv("CORA_MERGE_INLINE")) {                  // (debugging: on the job's own stream, as before round 3)
            RCQ(st_exact_finish(job, sh.rparts, sh.fst, sh.host_ms));
            sh.rparts.release_all(); sh.parts.release_all();
            return 0;
        }
        RCQ(join_merge(job, sh));
        if (!sh.side) sh.side = new MergeSide();
        MergeSide &m = *sh.side;
        if (!m.ctx) {
            PoolScope none(nullptr);
            RCQ(job_side_context(job, &m.ctx));
        }
        m.ctx->prm = job->prm; m.ctx->hg = job->hg; m.ctx->dp = job->dp; m.ctx->host_threads = job->host_threads;
        m.rc = 0; m.err.clear(); m.fst = FinishStats(); m.host_ms = 0;
        HIPQ(hipStreamSynchronize(s));                      // the class parts are all there
        ShardState *shp = &sh;
        m.th = std::thread([job, shp] {
            MergeSide &mm = *shp->side;
            if (hipSetDevice(job->device) != hipSuccess) { mm.rc = 100; mm.err = "hipSetDevice failed in the exact table's helper thread"; return; }
            PoolScope own(&mm.ctx->pool);
            mm.rc = st_exact_finish(mm.ctx, shp->rparts, mm.fst, mm.host_ms);
            if (mm.rc) mm.err = g_err;
        });
        return 0;
    }
    case ST_REPS: {
        if (!sh.repmask_g.p) return fail(19, "the global representative bitmap has not been imported");
        if (sh.filter) RCQ(st_reps(job, sh.repmask_g.p, 0, dp.n_words, sh.reps, sh.n_reps));
        else RCQ(st_reps(job, sh.repmask_g.p, sh.word_lo, sh.word_hi, sh.reps, sh.n_reps));
        return 0;
    }
    case ST_SEED: {
        RCQ(sh.send_a.alloc(sh.n_reps));
        sh.n_send = sh.n_reps;
        // large references: the records carry the two following seeds' codes as a sketch (kernels.hip, k_seed_codes_all); every
        // rank takes the same decision, from the size of the reference
        sh.seed_sketch = dp.nseeds >= 3 && (job->sketch_mode >= 0 ? job->sketch_mode != 0 : dp.G > (1ull << 28));
        if (sh.seed_sketch) {
            RCQ(sh.send_b.alloc(sh.n_reps));
            u64 *outs[4] = {nullptr, nullptr, nullptr, nullptr};
            u32 *hists[4] = {nullptr, nullptr, nullptr, nullptr};
            SeedSketches sk;
            memset(&sk, 0, sizeof sk);
            sk.on = 1;
            outs[arg - 1] = sh.send_a.p; sk.b[arg - 1] = sh.send_b.p;
            launch_seed_codes_all(s, dp, job->pk.p, sh.reps.p, sh.n_reps, outs, hists, sk, RS_TILE_MIN, arg - 1);
        } else {
            sh.send_b.release();
            launch_seed_codes(s, dp, job->pk.p, sh.reps.p, sh.n_reps, arg, sh.send_a.p);
        }
        HIPQ(hipStreamSynchronize(s));
        if (sh.filter) {                // VERIFY's sort keeps this rank's records
            for (int r = 0; r < sh.world; r++) counts[r] = 0;
            return 0;
        }
        RCQ(partition_by_dest(job, sh.send_a, nullptr, sh.n_send, 1, sh.world, counts, sh.seed_sketch ? &sh.send_b : nullptr));
        return 0;
    }
    case ST_VERIFY: {
        if (sh.filter) {
            const SortSource from = source_of(sh, 2, sh.seed_sketch, false);
            u64 kept = 0;
            // (the bit range must be fixed before the kept records have been counted: a rank keeps about its share of them)
            const bool lds = seed_lds_finish(sh.n_send / (u64)sh.world);
            if (radix_sort_count_kept(s, from, seed_sort_lo(sh.seed_sketch, lds), seed_sort_hi(sh.seed_sketch), job->sort_scratch, &kept)) return fail(100, "radix sort failed");
            if (kept) {
                RCQ(sh.recv_a.alloc(kept));
                // the sketches' second word climbs through a chain of arrays when the arena has the room (job.hip: why)
                int chain = 1;
                if (sh.seed_sketch) {
                    chain = shard_b_chain(job);
                    if (chain > 1 && sh.recv_b.alloc(kept * (u64)chain)) { chain = 1; (void)hipGetLastError(); }
                    if (chain == 1) RCQ(sh.recv_b.alloc(kept));
                }
                RCQ(st_seed_verify(job, sh.recv_a, kept, arg, sh.ps, nullptr, sh.seed_sketch ? sh.recv_b.p : nullptr, nullptr, &from, chain, lds));
            }
            sh.recv_a.release(); sh.send_a.release(); sh.recv_b.release(); sh.send_b.release();
            sh.n_send = sh.n_recv = 0;
            job->stats.hom_pairs = sh.ps.n;
            return 0;
        }
        if (sh.seed_sketch && sh.n_recv && !sh.recv_b.p) return fail(19, "the seed records' sketches have not been delivered");
        if (sh.n_recv) RCQ(st_seed_verify(job, sh.recv_a, sh.n_recv, arg, sh.ps, nullptr, sh.seed_sketch ? sh.recv_b.p : nullptr, nullptr, nullptr, sh.seed_sketch ? sh.b_chain : 1,
                                              seed_lds_finish(sh.n_recv)));
        sh.recv_a.release(); sh.send_a.release(); sh.recv_b.release(); sh.send_b.release();
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
        // items per source position in a window behind each nominal slice start (summed over ranks by the caller)
        sh.bnd.clear();
        const u64 n_blocks = (dp.G + 255) / 256;
        for (int r = 1; r < sh.world; r++) sh.bnd.push_back((u32)(n_blocks * r / sh.world * 256));
        sh.win = 1u << 16;
        const u64 nb = sh.bnd.size();
        RCQ(sh.wcount.alloc(nb * sh.win)); RCQ(sh.bnd_dev.alloc(nb));
        if (nb) {
            HIPQ(hipMemsetAsync(sh.wcount.p, 0, nb * sh.win * sizeof(u32), s));
            std::vector<u32> starts(sh.bnd);
            for (u32 &b : starts) if (b == 0) b = 1;          // fewer blocks than ranks: the kernel needs a start >= 1; the cut is 0 anyway
            HIPQ(hipMemcpyAsync(sh.bnd_dev.p, starts.data(), nb * sizeof(u32), hipMemcpyHostToDevice, s));
            launch_window_counts(s, sh.send_a.p, M, sh.bnd_dev.p, (u32)nb, sh.win, sh.wcount.p);
            HIPQ(hipStreamSynchronize(s));
            HIPQ(hipGetLastError());
        }
        return 0;
    }
    case ST_ITEM_CUT: {
        // cut r = first position >= the nominal start of rank r whose predecessor holds no list; every rank computes the
        // same cuts from the same summed counts.  No such position inside a window: rank 0 takes every item.
        const u64 nb = sh.bnd.size();
        std::vector<u32> wc(nb * sh.win), cuts(nb, 0xFFFFFFFFu);
        if (nb) {
            HIPQ(hipMemcpyAsync(wc.data(), sh.wcount.p, nb * sh.win * sizeof(u32), hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            bool all = true;
            for (u64 b = 0; b < nb; b++) {
                if (sh.bnd[b] == 0) { cuts[b] = 0; continue; }
                u32 j = 0;
                while (j < sh.win && wc[b * sh.win + j]) j++;
                if (j == sh.win) { all = false; break; }
                cuts[b] = sh.bnd[b] + j;
            }
            if (!all) std::fill(cuts.begin(), cuts.end(), 0xFFFFFFFFu);
        }
        for (int r = 0; r < sh.world; r++) counts[r] = 0;
        if (sh.world == 1 || sh.n_send == 0) { counts[0] = sh.world == 1 ? sh.n_send : 0; return 0; }
        DBuf<u64> dest, tdest, ta;
        DBuf<u32> tc, cuts_dev;
        DBuf<unsigned long long> dcnt;
        const u64 n = sh.n_send;
        RCQ(dest.alloc(n)); RCQ(tdest.alloc(n)); RCQ(ta.alloc(n)); RCQ(tc.alloc(n)); RCQ(dcnt.alloc(256)); RCQ(cuts_dev.alloc(nb));
        HIPQ(hipMemcpyAsync(cuts_dev.p, cuts.data(), nb * sizeof(u32), hipMemcpyHostToDevice, s));
        HIPQ(hipMemsetAsync(dcnt.p, 0, 256 * sizeof(unsigned long long), s));
        launch_dest_by_cuts(s, sh.send_a.p, n, cuts_dev.p, (u32)nb, dest.p);
        launch_count_dest(s, dest.p, n, (u32)sh.world, dcnt.p);
        SortArrays keys, tmp;
        keys.a = dest.p; keys.b = sh.send_a.p; keys.c = sh.send_c.p;
        tmp.a = tdest.p; tmp.b = ta.p; tmp.c = tc.p;
        if (radix_sort(s, keys, tmp, n, 64, 72, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        unsigned long long h[256];
        HIPQ(hipMemcpyAsync(h, dcnt.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        HIPQ(hipGetLastError());
        if (keys.b != sh.send_a.p) { std::swap(sh.send_a.p, ta.p); std::swap(sh.send_c.p, tc.p); }
        for (int r = 0; r < sh.world; r++) counts[r] = h[r];
        sh.wcount.release(); sh.bnd_dev.release();
        return 0;
    }
    case ST_FINISH: {
        RCQ(join_merge(job, sh));                           // rank 0: the exact table, finished beside the seeds stage
        if (!sh.recv_a.p) { RCQ(sh.recv_a.alloc(1)); RCQ(sh.recv_c.alloc(1)); }
        RCQ(st_items_finish(job, sh.recv_a, sh.recv_c, sh.n_recv, sh.fst, sh.host_ms));
        sh.recv_a.release(); sh.recv_c.release(); sh.send_a.release(); sh.send_c.release();
        cora_hip_stats &st = job->stats;
        st.exact_nodes = sh.fst.exact_nodes; st.inexact_nodes = sh.fst.inexact_nodes;
        st.dirty_nodes = sh.fst.dirty_nodes; st.big_nodes = sh.fst.big_nodes;
        // this rank's bytes and counts, for the gather / the sum.  Rank 0's own bytes are the head of the table and are in its
        // host buffer already: they stay there (COLLECT puts the other ranks' bytes behind them) and do not make the round
        // trip through the device
        sh.n_f2 = (sh.rank == 0 || sh.host_gather) ? 0 : job->inexact_bytes.size();
        RCQ(sh.f2_dev.alloc(sh.n_f2)); RCQ(sh.stat_dev.alloc(N_STATS));
        if (sh.n_f2) HIPQ(hipMemcpyAsync(sh.f2_dev.p, job->inexact_bytes.data(), sh.n_f2, hipMemcpyHostToDevice, s));
        const u64 hs[N_STATS] = {st.inexact_nodes, st.dirty_nodes, st.big_nodes, st.inexact_items, st.hom_pairs, st.candidate_pairs, 0, 0};
        HIPQ(hipMemcpyAsync(sh.stat_dev.p, hs, sizeof hs, hipMemcpyHostToDevice, s));
        HIPQ(hipStreamSynchronize(s));
        st.ms_host_finish = (float)sh.host_ms;
        st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t_begin).count();
        st.ms_sort_scatter = job->sort_timing.scatter_ms;
        st.n_sort_scatter = job->sort_timing.scatter_launches;
        st.sort_scatter_bytes = job->sort_timing.scatter_bytes;
        st.pool_peak_bytes = job->pool.peak_live;
        return 0;
    }
    case ST_COLLECT: {                  // rank 0: the table is the ranks' byte ranges in rank order; counts are the sums
        cora_hip_stats &st = job->stats;
        ByteBuf &tb = job->inexact_bytes;
        const size_t own = tb.size(), total = own + sh.n_f2_all;
        if (!tb.grow_keep(total + 16))                  // (first build of a job only: later ones find the room of the last table)
            return fail(101, "host allocation of %llu bytes for <Inexact> failed", (unsigned long long)total + 16);
        if (sh.n_f2_all) HIPQ(hipMemcpyAsync(tb.p + own, sh.f2_all.p, sh.n_f2_all, hipMemcpyDeviceToHost, s));
        u64 hs[N_STATS] = {0};
        if (sh.stat_dev.p) HIPQ(hipMemcpyAsync(hs, sh.stat_dev.p, sizeof hs, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        tb.n = total;
        st.inexact_nodes = hs[0]; st.dirty_nodes = hs[1]; st.big_nodes = hs[2]; st.inexact_items = hs[3]; st.hom_pairs = hs[4];
        st.candidate_pairs = hs[5];
        sh.f2_all.release(); sh.f2_dev.release();
        st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sh.t_begin).count();
        return 0;
    }
    default:
        return fail(19, "unknown shard stage %d", stage);
    }
}

// room for n sketches arriving from the other ranks: the first of sh.b_chain arrays when the arena has the room for the chain
// their second word climbs through during the sort (VERIFY), else one array
static int alloc_recv_b(cora_hip_job *job, ShardState &sh, u64 n)
{
    sh.b_chain = shard_b_chain(job);
    if (sh.b_chain > 1 && sh.recv_b.alloc((n ? n : 1) * (u64)sh.b_chain)) { sh.b_chain = 1; (void)hipGetLastError(); }
    if (sh.b_chain == 1) RCQ(sh.recv_b.alloc(n));
    return 0;
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
    case SLOT_WCOUNT: *p = sh.wcount.p; *n = (u64)sh.bnd.size() * sh.win; *eb = 4; return 0;
    case SLOT_F2: *p = sh.f2_dev.p; *n = sh.n_f2; *eb = 1; return 0;
    case SLOT_STATS: *p = sh.stat_dev.p; *n = sh.stat_dev.p ? N_STATS : 0; *eb = 8; return 0;
    case SLOT_SEND_B: *p = sh.send_b.p; *n = sh.send_b.p ? sh.n_send : 0; *eb = 8; return 0;
    case SLOT_RECV_B: *p = sh.recv_b.p; *n = sh.recv_b.p ? sh.n_recv : 0; *eb = 8; return 0;
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

// Device address of a slot's data (valid until the job's next stage / import call): lets the caller hand the
// records to its collective library without a copy.
extern "C" int cora_hip_shard_slot_ptr(cora_hip_job *job, int slot, void **ptr)
{
    if (!job || !job->shard || !ptr) return fail(19, "cora_hip_shard_begin has not run");
    void *p; u64 n; u32 eb;
    if (slot_ref(*job->shard, job->dp, slot, &p, &n, &eb)) return fail(19, "slot %d cannot be exported", slot);
    HIPQ(hipStreamSynchronize(job->stream));
    *ptr = n ? p : nullptr;
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
    case SLOT_RECV_B: if (n != sh.n_recv) return fail(19, "RECV_B must match RECV_A"); RCQ(alloc_recv_b(job, sh, n)); dst = sh.recv_b.p; eb = 8; break;
    case SLOT_REPMASK:
        if (n != job->dp.n_words) return fail(19, "bitmap size mismatch");
        RCQ(sh.repmask_g.alloc(n)); dst = sh.repmask_g.p; eb = 4; break;
    case SLOT_RX_CLS: RCQ(sh.rparts.cls.alloc(n)); sh.rparts.NNS = n; dst = sh.rparts.cls.p; eb = 8; break;
    case SLOT_RX_CNT: RCQ(sh.rparts.ce.alloc(n)); if (n != sh.rparts.NNS) return fail(19, "RX_CNT must match RX_CLS"); dst = sh.rparts.ce.p; eb = 4; break;
    case SLOT_RX_ENT: RCQ(sh.rparts.ent.alloc(n)); sh.rparts.NE = n; dst = sh.rparts.ent.p; eb = 4; break;
    case SLOT_RX_DIR: RCQ(sh.rparts.edir.alloc(n)); if (n != sh.rparts.NE) return fail(19, "RX_DIR must match RX_ENT"); dst = sh.rparts.edir.p; eb = 1; break;
    case SLOT_WCOUNT:
        if (n != (u64)sh.bnd.size() * sh.win || !sh.wcount.p) return fail(19, "window count size mismatch");
        dst = sh.wcount.p; eb = 4; break;
    case SLOT_RX_F2: RCQ(sh.f2_all.alloc(n)); sh.n_f2_all = n; dst = sh.f2_all.p; eb = 1; break;
    case SLOT_STATS:
        if (n != N_STATS || !sh.stat_dev.p) return fail(19, "stats size mismatch");
        dst = sh.stat_dev.p; eb = 8; break;
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

// Sizes a receive slot for n elements and returns its device address, so that the caller's collective can deliver straight
// into the library's memory (no staging tensor, no copy).  The data must be there before the next stage call.
extern "C" int cora_hip_shard_slot_reserve(cora_hip_job *job, int slot, uint64_t n, void **ptr)
{
    if (!job || !job->shard || !ptr) return fail(19, "cora_hip_shard_begin has not run");
    PoolScope scope(&job->pool);
    ShardState &sh = *job->shard;
    switch (slot) {
    case SLOT_RECV_A: RCQ(sh.recv_a.alloc(n)); sh.n_recv = n; *ptr = sh.recv_a.p; break;
    case SLOT_RECV_C: if (n != sh.n_recv) return fail(19, "RECV_C must match RECV_A"); RCQ(sh.recv_c.alloc(n)); *ptr = sh.recv_c.p; break;
    case SLOT_RECV_B: if (n != sh.n_recv) return fail(19, "RECV_B must match RECV_A"); RCQ(alloc_recv_b(job, sh, n)); *ptr = sh.recv_b.p; break;
    case SLOT_RX_CLS: RCQ(sh.rparts.cls.alloc(n)); sh.rparts.NNS = n; *ptr = sh.rparts.cls.p; break;
    case SLOT_RX_CNT: if (n != sh.rparts.NNS) return fail(19, "RX_CNT must match RX_CLS"); RCQ(sh.rparts.ce.alloc(n)); *ptr = sh.rparts.ce.p; break;
    case SLOT_RX_ENT: RCQ(sh.rparts.ent.alloc(n)); sh.rparts.NE = n; *ptr = sh.rparts.ent.p; break;
    case SLOT_RX_DIR: if (n != sh.rparts.NE) return fail(19, "RX_DIR must match RX_ENT"); RCQ(sh.rparts.edir.alloc(n)); *ptr = sh.rparts.edir.p; break;
    case SLOT_RX_F2: RCQ(sh.f2_all.alloc(n)); sh.n_f2_all = n; *ptr = sh.f2_all.p; break;
    default: return fail(19, "slot %d cannot be reserved", slot);
    }
    HIPQ(hipStreamSynchronize(job->stream));
    return 0;
}

// How this build brings a rank its K-mer and seed records (see the top of the file): *on = 1 when every rank scans all
// positions and keeps its own records -- the caller then moves NOTHING after the KMERS and SEED stages (their records are in the
// receive slots already) -- 0 when the caller runs the all-to-all.  set >= 0 overrides the choice cora_hip_shard_begin made
// (all ranks must agree; before the KMERS stage).
extern "C" int cora_hip_shard_filter(cora_hip_job *job, int set, int *on)
{
    if (!job || !job->shard) return fail(19, "cora_hip_shard_begin has not run");
    if (set >= 0) job->shard->filter = job->shard->world > 1 && set != 0;
    if (on) *on = job->shard->filter ? 1 : 0;
    return 0;
}

// after a SEED stage: do the seed records of this build carry sketches (SEND_B / RECV_B move like SEND_A / RECV_A)?  The same
// answer on every rank: it follows from the size of the reference.
extern "C" int cora_hip_shard_seed_sketch(cora_hip_job *job, int *on)
{
    if (!job || !job->shard || !on) return fail(19, "bad argument");
    *on = job->shard->seed_sketch ? 1 : 0;
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
