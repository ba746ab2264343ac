// job_internal.hpp -- state of one table-building job and the small host utilities shared by
// job.hip (single-GPU orchestration) and shard.hip (multi-GPU stages).
#pragma once
#include "../../include/cora_hip.h"
#include "arena_ranges.hpp"
#include "finish.hpp"
#include "kernels.hpp"
#include "radix_sort.hpp"

#include <chrono>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

using namespace cora;

inline thread_local std::string g_err;
inline int g_device = -1;

inline int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code ? code : 1;
}

#define HIPQ(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(100, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define RCQ(x) do { int r_ = (x); if (r_) return r_; } while (0)

// Device memory pool of one job: stages allocate and release dozens of arrays per run and
// hipMalloc/hipFree of large blocks cost milliseconds each (hipFree also synchronises the device),
// so released blocks are kept and handed out again (same sizes recur every run).
struct DevPool {
    struct Block { void *p; size_t bytes; bool free; };
    std::vector<Block> blocks;
    size_t n_malloc = 0, malloc_bytes = 0;               // since the pool was made (CORA_TRACE prints them per run)
    size_t live_bytes = 0, peak_live = 0;                // bytes handed out now / at most
    // The arena (large references, single-GPU run): ONE allocation made before the first stage, from which the arrays
    // are cut -- arrays of 256 MB and more at multiples of 1 GB from its low end, smaller ones at multiples of 2 MB from
    // its high end.  The sort's scatter runs at the pace its buffers allow, and that differed by 15 % between arrays that
    // came from separate hipMalloc calls (tools/sortbench, profiles/sort_alloc_size_sweep.sh: 19.9 ms per pass over 2.6e9
    // two-word records in the first process on a box, 17.2-19.6 in later ones, the two directions of the ping-pong
    // differently; shifting the arrays inside their allocations changed nothing) while arrays cut from one allocation
    // at multiples of 1 GB ran at 17.1-17.6 ms in every process (at multiples of 2 MB: 17.0-18.6).  What the arena cannot
    // hold comes from the blocks above, as before.
    char *arena = nullptr;
    size_t arena_bytes = 0;
    ArenaRanges ranges;                                  // which parts of the arena are free / handed out (arena_ranges.hpp)
    // The last tail_bytes of the allocation are not the pool's to cut arrays from: they hold the digit bytes of the sorts
    // (NextDigitLease).  As an allocation of their own those 3 GB lay in the device's last third, which is slow to give back: the
    // executable took 0.88 instead of 0.79 s (profiles/digit_lease_ab.sh).
    char *tail = nullptr;
    size_t tail_bytes = 0;
    bool reserve_arena(size_t bytes, size_t tail_want = 0)
    {
        if (arena) return true;
        tail_want = (tail_want + ArenaRanges::ALIGN - 1) & ~(ArenaRanges::ALIGN - 1);
        if (tail_want >= bytes / 2) tail_want = 0;
        // (the tail is taken out of the arena's size, not added: on top of 190 GiB it made the allocation slow to give back -- the
        //  executable took 0.885 instead of 0.795 s -- for a build 0-4 ms faster or slower, depending on the configuration)
        bytes = ((bytes - tail_want) + ArenaRanges::ALIGN_BIG - 1) & ~(ArenaRanges::ALIGN_BIG - 1);
        void *p = nullptr;
        if (hipMalloc(&p, bytes + tail_want) != hipSuccess) { (void)hipGetLastError(); return false; }
        arena = (char *)p; arena_bytes = bytes;
        tail = tail_want ? arena + bytes : nullptr; tail_bytes = tail_want;
        ranges.reset((size_t)p, bytes);
        n_malloc++; malloc_bytes += bytes + tail_want;
        return true;
    }
    void release_arena()                                 // (only when nothing of it is in use)
    {
        if (!arena || !ranges.live.empty()) return;
        (void)hipFree(arena);
        arena = nullptr; arena_bytes = 0; tail = nullptr; tail_bytes = 0; ranges.clear();
    }
    void *arena_get(size_t bytes)
    {
        const size_t at = ranges.get(bytes);
        return at == ArenaRanges::NONE ? nullptr : arena + at;
    }
    bool arena_put(void *p)
    {
        if (!arena || (char *)p < arena || (char *)p >= arena + arena_bytes) return false;
        live_bytes -= ranges.put((size_t)((char *)p - arena));      // (0: inside the arena but not live -- released twice; nothing to do)
        return true;
    }
    void *get(size_t bytes)
    {
        bytes = (bytes + (ArenaRanges::ALIGN - 1)) & ~(ArenaRanges::ALIGN - 1);
        if (arena) {
            void *q = arena_get(bytes);
            if (q) { live_bytes += bytes; if (live_bytes > peak_live) peak_live = live_bytes; return q; }
        }
        // smallest free block that holds the request.  A close fit (within 25 %) is the common case: the same sizes recur
        // in every run.  A LARGE request also takes a larger block rather than growing the pool: the stages of a
        // GRCh38-sized build ask for arrays of 12-47 GB each, and a pool that only ever adds blocks ended at 277 GB.
        int best = -1;
        const bool any_fit = bytes >= ((size_t)256 << 20);
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].free && blocks[i].bytes >= bytes && (any_fit || blocks[i].bytes <= bytes + bytes / 4 + (4u << 20)) &&
                (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = (int)i;
        if (best >= 0) {
            blocks[best].free = false;
            live_bytes += blocks[best].bytes; if (live_bytes > peak_live) peak_live = live_bytes;
            return blocks[best].p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            trim();                                          // give cached blocks back and retry once
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        }
        blocks.push_back({p, bytes, false});
        n_malloc++; malloc_bytes += bytes;
        live_bytes += bytes; if (live_bytes > peak_live) peak_live = live_bytes;
        return p;
    }
    void put(void *p)
    {
        if (arena_put(p)) return;
        for (auto &b : blocks) if (b.p == p) { if (!b.free) live_bytes -= b.bytes; b.free = true; return; }
    }
    void trim()
    {
        std::vector<Block> keep;
        for (auto &b : blocks) { if (b.free) (void)hipFree(b.p); else keep.push_back(b); }
        blocks.swap(keep);
        release_arena();
    }
    void destroy()
    {
        for (auto &b : blocks) (void)hipFree(b.p);
        blocks.clear();
        if (arena) (void)hipFree(arena);
        arena = nullptr; arena_bytes = 0; tail = nullptr; tail_bytes = 0; ranges.clear();
    }
};
inline thread_local DevPool *g_pool = nullptr;       // pool of the job the calling thread is working for

template <typename T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DevPool *pool = nullptr;
    int alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        pool = g_pool;
        if (pool) p = (T *)pool->get(count * sizeof(T));
        else if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) p = nullptr;
        if (!p) return fail(101, "device allocation of %zu bytes failed", count * sizeof(T));
        n = count;
        return 0;
    }
    void release()
    {
        if (p) { if (pool) pool->put(p); else (void)hipFree(p); }
        p = nullptr; n = 0;
    }
    ~DBuf() { release(); }
    DBuf() {}
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
};

// grow-only pinned host array for results copied back every run
template <typename T>
struct HBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t count)
    {
        if (count <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = count + count / 8 + 1024;
        if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) { p = nullptr; return fail(101, "pinned host allocation of %zu bytes failed", want * sizeof(T)); }
        cap = want;
        return 0;
    }
    ~HBuf() { if (p) (void)hipHostFree(p); }
};

// Lends the sort scratch pool memory for its digit bytes (radix_sort.hip, "the next digit travels ahead") while sorts of up to n records
// run; a lease inside another one does nothing.  Without room in the pool the sorts read key words for their histograms, as before.
struct NextDigitLease {
    SortScratch &sc;
    DBuf<u8> buf;
    bool mine = false;
    NextDigitLease(SortScratch &scratch, u64 n) : sc(scratch)
    {
        if (sc.lease_next_digit || !radix_sort_next_digit_wanted(sc, n)) return;
        if (g_pool && g_pool->tail && n <= g_pool->tail_bytes) { sc.lease_next_digit = (unsigned char *)g_pool->tail; sc.cap_lease_next_digit = g_pool->tail_bytes; mine = true; return; }
        if (n > RS_OWN_NEXT_DIGIT_MAX && !buf.alloc(n)) { sc.lease_next_digit = buf.p; sc.cap_lease_next_digit = (size_t)n; mine = true; }
        else (void)hipGetLastError();               // (a sort of up to RS_OWN_NEXT_DIGIT_MAX records: the scratch's own array will do)
    }
    ~NextDigitLease() { if (mine) { sc.lease_next_digit = nullptr; sc.cap_lease_next_digit = 0; sc.next_digit = sc.own_next_digit; } }
    NextDigitLease(const NextDigitLease &) = delete;
    NextDigitLease &operator=(const NextDigitLease &) = delete;
};

struct PoolScope {                                       // binds a job's pool to the calling thread for one API call
    DevPool *prev;
    explicit PoolScope(DevPool *p) : prev(g_pool) { g_pool = p; }
    ~PoolScope() { g_pool = prev; }
};

constexpr int STREAM_PIECES = 8;          // the byte stream of <Inexact> is written and sent in this many pieces (st_items_finish)
constexpr int EV_WRITTEN = 0, EV_ARRIVED = STREAM_PIECES, EV_PACKED_READY = 2 * STREAM_PIECES, EV_PACKED_ARRIVED = 2 * STREAM_PIECES + 1;

struct cora_hip_job {
    cora_hip_params prm;
    HostGenome hg;
    DevParams dp;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // Copies to the host that run beside the kernels (their own stream): the packed reference, which the host part of
    // <Inexact> reads, leaves right after k_pack; the <Inexact> byte stream leaves in pieces while k_node_write writes the rest.
    hipStream_t copy_stream = nullptr;
    // events of the copy stream: piece c of the <Inexact> byte stream written (EV_WRITTEN + c) / arrived on the host (EV_ARRIVED + c),
    // the packed reference ready to leave / arrived
    hipEvent_t ev_copy[2 * STREAM_PIECES + 2] = {};
    bool packed_in_flight = false;        // h_packed is being filled (or is full) for this run
    bool have_ref = false;
    uint64_t genome_bases = 0;
    DBuf<u8> raw;
    DBuf<u64> d_gstart, d_off;
    DBuf<u32> d_len, d_window;
    DBuf<u64> pk;
    DBuf<u32> bad;
    DBuf<u32> valid;              // window-validity bitmap of the positions k_kmer_count has covered in this run
    DBuf<DevCounters> ctr;
    SortScratch sort_scratch;
    SortTiming sort_timing;
    DevPool pool;
    bool arena_tried = false;           // the job has asked for the pool's arena (large references: reserve_job_arena)
    int arena_world = 0;                // ... for a build over that many ranks (1: the single-GPU run)
    HBuf<u64> h_key, h_packed;
    HBuf<u32> h_meta, h_node_start;
    HBuf<u8> h_stream, h_exact;   // landing areas of the <Inexact> byte stream and of the <Exact> bytes
    HBuf<HostNode> h_hnodes;
    HBuf<int16_t> h_blen;
    int host_threads = 8;
    struct ShardState *shard = nullptr;   // multi-GPU stage state (shard.hip)
    struct MultiState *multi = nullptr;   // the other ranks' jobs of an in-process multi-GPU run (multi.hip)
    int device = 0;                       // HIP device the job's memory and stream live on
    int ranks_on_device = 1;              // ranks of a sharded build that share this job's device (rehearsals on one GPU: they share its memory)
    std::vector<cora_hip_contig> ref_contigs;   // what set_reference_impl was given (a multi-GPU run replicates the reference)
    u32 ref_W = 0;
    size_t ref_image_bytes = 0;
    int sketch_mode = -1;         // -1 = by size (job.hip: SKETCH_MIN_REPS); CORA_SKETCH=0/1 overrides
    int hash_bits = 0;            // 0 = by size; CORA_HASH_BITS overrides (tests force collisions with small values)
    ByteBuf exact_bytes, inexact_bytes;
    // First build of a job on a large reference: a helper thread pins h_stream and faults in the pages of inexact_bytes
    // beside the kernels (job.hip); whoever touches either buffer calls join_warm first.
    std::thread warm;
    std::thread warm_exact;               // ... and <Exact>'s two buffers, wanted first (st_exact_finish joins this one only)
    // cora_hip_job_set_table_files: <Inexact> is assembled in a mapping of its file (inexact_bytes is then file-backed), <Exact> is
    // written by a helper thread as soon as it is finished, beside the inexact stages
    std::string exact_path;
    std::thread exact_writer;
    bool exact_written = false, exact_write_failed = false;
    cora_hip_stats stats;
};

inline void join_warm_exact(cora_hip_job *job) { if (job->warm_exact.joinable()) job->warm_exact.join(); }       // before exact_bytes / h_exact are touched
inline void join_warm(cora_hip_job *job) { join_warm_exact(job); if (job->warm.joinable()) job->warm.join(); }
inline void join_exact_writer(cora_hip_job *job) { if (job->exact_writer.joinable()) job->exact_writer.join(); }

inline int read_counters(cora_hip_job *job, DevCounters &h)
{
    HIPQ(hipMemcpyAsync(&h, job->ctr.p, sizeof h, hipMemcpyDeviceToHost, job->stream));
    HIPQ(hipStreamSynchronize(job->stream));
    return 0;
}
inline int write_counters(cora_hip_job *job, const DevCounters &h)
{
    HIPQ(hipMemcpyAsync(job->ctr.p, &h, sizeof h, hipMemcpyHostToDevice, job->stream));
    HIPQ(hipStreamSynchronize(job->stream));
    return 0;
}

// zeroes n consecutive 8-byte counters, in stream order (no host round trip)
#define ZERO_COUNTERS(job, field, n) HIPQ(hipMemsetAsync((char *)(job)->ctr.p + offsetof(DevCounters, field), 0, 8 * (n), (job)->stream))
#define FORMAT_ERROR_CHECK(h) do { if ((h).format_errors) return fail(51, "FASTA image does not have the fixed line width the reference's loader assumes (%llu misplaced line ends)", (unsigned long long)(h).format_errors); } while (0)

// CORA_TRACE=2: wall time between marks, each mark synchronising the stream first (for finding where a run loses time)
inline void trace_mark(cora_hip_job *job, const char *what, int idx = -1)
{
    static const int level = getenv("CORA_TRACE") ? atoi(getenv("CORA_TRACE")) : 0;
    static const int mask = getenv("CORA_SYNC_MASK") ? atoi(getenv("CORA_SYNC_MASK")) : 0;
    if (idx >= 0 && ((mask >> idx) & 1)) (void)hipStreamSynchronize(job->stream);
    if (level < 2) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    (void)hipStreamSynchronize(job->stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[cora]   %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
}

inline int bits_for(u64 G)
{
    int b = 1;
    while (b < 32 && (1ull << b) < G) b++;
    return b;
}

template <typename T>
inline int d2h(cora_hip_job *job, HBuf<T> &dst, View<T> &view, const T *src, size_t n)
{
    RCQ(dst.reserve(n));
    if (n) HIPQ(hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyDeviceToHost, job->stream));
    view.p = dst.p; view.n = n;
    return 0;
}


// ---- stage functions (stages.hip), shared by the single-GPU run and the sharded flow ----
struct ExactLocal {                  // device results of class detection over one rank's records
    DBuf<u64> rec, trec;             // records and sort ping-pong; rec_s points at the sorted copy (flag byte in the low 8 bits)
    DBuf<u32> gp, tgp;
    const u64 *rec_s = nullptr;
    const u32 *gp_s = nullptr;
    DBuf<u32> cls_m;                 // [NNS] records of each class with >= 2 entries
    DBuf<u32> nonrep;                // [n_words] later members of this rank's classes (every valid window not in here is a representative)
    DBuf<u64> cls;                   // [NNS] rep<<32 | index of the first record, classes with >= 2 entries
    u64 N = 0, NR = 0, NNS = 0;
    bool first_digit_ready = false;  // st_kmers left the records' first sort digit in the sort scratch (consumed by st_exact_classes)
    void release_all() { rec.release(); trec.release(); gp.release(); tgp.release(); cls_m.release(); nonrep.release(); cls.release(); rec_s = nullptr; gp_s = nullptr; }
};
struct ExactParts {                  // entry lists of classes, ent laid out in the order of cls
    DBuf<u64> cls;
    DBuf<u32> ce, ent;
    DBuf<u8> edir;
    u64 NNS = 0, NE = 0;
    void release_all() { cls.release(); ce.release(); ent.release(); edir.release(); NNS = NE = 0; }
};
struct PairStore {
    DBuf<PairRec> pairs;
    u64 n = 0, cap = 0;
};

int st_pack(cora_hip_job *job);
int st_kmers(cora_hip_job *job, u32 block_lo, u32 n_blocks, DBuf<u64> &rec, DBuf<u32> &gp, u64 &n_out, bool *first_digit_ready = nullptr);
int st_exact_classes(cora_hip_job *job, ExactLocal &x, const SortSource *from = nullptr);
inline int exact_sort_bits(const cora_hip_job *job) { return job->hash_bits ? job->hash_bits : 40; }     // the exact sort runs on the top bits of the hash
// A seed sort of many records leaves the last code byte to the candidate kernels, which order every bucket of equal top 24 bits in
// LDS (kernels.hip, "the last sort digit in LDS"): three global passes instead of four.  That pays when a bucket holds some tens of
// records (2^24 buckets; a 64 Mbp reference, 4 records per bucket, spent twice the time in its seeds stage that way).  Measured on
// the bench generator (profiles/seed_lds_threshold_ab.sh; seeds stage, four passes against three + LDS): 4.9e8 records 72.5 / 78.6 ms,
// 9.5e8 141.3 / 138.5, 1.45e9 203.0 / 192.0, 2.9e9 413 / 377: from SEED_LDS_MIN_RECORDS on.  CORA_SEED_LDS=0: never (the four
// passes of before round 3), 2: always (tests), a larger value: from that many records on.
constexpr u64 SEED_LDS_MIN_RECORDS = 900000000ull;
inline bool seed_lds_finish(u64 n_records)
{
    const char *e = getenv("CORA_SEED_LDS");
    const long long v = e ? atoll(e) : 1;
    if (v == 0) return false;
    if (v == 2) return true;
    return n_records >= (v > 2 ? (u64)v : SEED_LDS_MIN_RECORDS);
}
inline int seed_code_lo(bool with_sketch) { return with_sketch ? 96 : 32; }                             // a seed record's code: 32 bits from here
inline int seed_sort_lo(bool with_sketch, bool lds) { return seed_code_lo(with_sketch) + (lds ? 8 : 0); }   // the bits its sort runs on start here
inline int seed_sort_hi(bool with_sketch) { return seed_code_lo(with_sketch) + 32; }
inline int seed_first_digit_shift(bool lds) { return lds ? 8 : 0; }                                     // ... as a shift of the code (k_seed_codes_all's histograms)
int st_exact_parts(cora_hip_job *job, ExactLocal &x, ExactParts &p);
int st_exact_finish(cora_hip_job *job, ExactParts &p, FinishStats &fst, double &host_ms);
int st_reps(cora_hip_job *job, const u32 *repmask, u64 word_lo, u64 word_hi, DBuf<u32> &reps, u64 &n_out);
int st_seed_verify(cora_hip_job *job, DBuf<u64> &codes, u64 n, int seed, PairStore &ps, u32 *first_hist = nullptr,
                   u64 *sk_b = nullptr, u32 *sk_c = nullptr /* [n] per-record sketches for the candidate kernels' pre-filter; sorted along */,
                   const SortSource *from = nullptr, int b_chain_n = 1 /* sk_b is the first of that many arrays of n words (radix_sort's chain) */,
                   bool lds = false /* seed_lds_finish: the sort stops above the last code byte (first_hist / from were made for that range) */);
int st_items_finish(cora_hip_job *job, DBuf<u64> &ik, DBuf<u32> &im, u64 M, FinishStats &fst, double &host_ms);
void shard_state_free(cora_hip_job *job);
// a second context on the job's device for work that runs beside the job's own stream (shard.hip: the exact table's finish):
// own stream, events, counters, sort scratch and pool; no reference of its own (the caller copies dp / hg from the job)
int job_side_context(cora_hip_job *job, cora_hip_job **out);
// in-process ranks (multi.hip): FINISH leaves this rank's <Inexact> bytes in its host buffer only -- the ranks share the address
// space, so they are copied host to host into rank 0's table instead of going through the device (SLOT_F2 stays empty)
void shard_set_host_gather(cora_hip_job *job, bool on);
// the arena of a large build (DevPool): reserved once per job, sized for what `world` ranks leave each of them to hold
void reserve_job_arena(cora_hip_job *job, int world, bool keep_own);
void multi_state_free(cora_hip_job *job);
int set_reference_impl(cora_hip_job *job, const cora_hip_contig *contigs, int32_t n_contigs, u32 W, size_t image_bytes,
                       const uint8_t *host_image, const uint8_t *dev_image, DBuf<u8> *adopt);
