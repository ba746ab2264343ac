// multi.hip -- one table build over several GPUs from ONE process: cora_hip_job_run_multi.
//
// The reference has no counterpart (it is a single process on one host).  This is the sharded flow
// of shard.hip -- every rank holds the packed reference, scans 1/R of the positions and owns 1/R of the key space --
// driven by one host thread per rank instead of one process per rank, so that the drop-in executable
// (`cora coraIndex` -> homTable_setup) uses every GPU of the node without a launcher.  The records move between the ranks'
// device buffers by peer-to-peer copies (hipMemcpyPeerAsync: the SDMA engines over xGMI, every receiver pulling from every
// sender at once, so each GPU pair uses its own link); ranks that share a device (the one-GPU rehearsal of the tests)
// copy device to device.  The multi-process flow over RCCL (cora_amd/dist.py, bench.py --gpus N under
// torch.distributed.run) moves the same slots with all_to_all / all_reduce collectives.
//
// Stage order and what is exchanged after each stage: the table at the top of shard.hip.
#include "job_internal.hpp"

#include <algorithm>
#include <atomic>
#include <pthread.h>
#include <thread>

using namespace cora;

extern "C" {
int cora_hip_shard_begin(cora_hip_job *job, int rank, int world);
int cora_hip_shard_stage(cora_hip_job *job, int stage, int arg, uint64_t *counts);
int cora_hip_shard_slot_info(cora_hip_job *job, int slot, uint64_t *n_elems, uint32_t *elem_bytes);
int cora_hip_shard_slot_ptr(cora_hip_job *job, int slot, void **ptr);
int cora_hip_shard_slot_import(cora_hip_job *job, int slot, const void *src_device, uint64_t n_elems);
int cora_hip_shard_slot_reserve(cora_hip_job *job, int slot, uint64_t n_elems, void **ptr);
int cora_hip_shard_get_first_singleton(cora_hip_job *job, uint64_t *key);
int cora_hip_shard_seed_sketch(cora_hip_job *job, int *on);
int cora_hip_shard_filter(cora_hip_job *job, int set, int *on);
int cora_hip_shard_set_first_singleton(cora_hip_job *job, uint64_t key);
}

struct MultiState {
    std::vector<cora_hip_job *> clones;      // jobs of ranks 1 .. world-1 (rank 0 is the caller's job)
    std::vector<int> devices;                // device of every rank
};

void multi_state_free(cora_hip_job *job)
{
    if (!job->multi) return;
    MultiState *m = job->multi;
    job->multi = nullptr;
    for (cora_hip_job *c : m->clones) {
        if (!c) continue;
        (void)hipSetDevice(c->device);
        cora_hip_job_destroy(c);
    }
    (void)hipSetDevice(job->device);
    delete m;
}

namespace {

constexpr int MAX_WORLD = 64;
constexpr int MAX_SLOTS_PER_EXCHANGE = 4;

struct Published {                            // what a rank offers to the others in one exchange
    void *ptr[MAX_SLOTS_PER_EXCHANGE];
    uint64_t n[MAX_SLOTS_PER_EXCHANGE];       // elements of each slot (gathers)
    uint64_t counts[MAX_WORLD];               // all-to-all: records for every rank
    uint64_t value;                           // a host scalar (first singleton)
    uint64_t stats[8];
};

struct Team {
    int world = 1;
    std::vector<cora_hip_job *> jobs;
    std::vector<int> dev;
    std::vector<Published> pub;
    pthread_barrier_t bar;
    std::atomic<int> failed{0};
    std::vector<int> rc;
    std::vector<std::string> err;
};

struct Rank {
    Team &t;
    int r;
    cora_hip_job *job;
    hipStream_t s;
    Rank(Team &team, int rank) : t(team), r(rank), job(team.jobs[rank]), s(team.jobs[rank]->stream) {}

    // Every step has the shape: work (skipped once someone has failed) -> sync() -> everyone takes the same decision.
    // Two barriers: nobody may raise `failed` again (in its next step) before everybody has read it for this step.
    bool sync()
    {
        pthread_barrier_wait(&t.bar);
        const bool ok = t.failed.load() == 0;
        pthread_barrier_wait(&t.bar);
        return ok;
    }
    void fail_with(int code)
    {
        if (!t.rc[r]) { t.rc[r] = code ? code : 1; t.err[r] = g_err; }
        t.failed.store(1);
    }
#define TRY(x) do { if (!t.failed.load()) { int rc_ = (x); if (rc_) fail_with(rc_); } } while (0)

    int copy_from(int src_rank, void *dst, const void *src, size_t bytes)
    {
        if (!bytes) return 0;
        if (t.dev[src_rank] == t.dev[r]) HIPQ(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
        else HIPQ(hipMemcpyPeerAsync(dst, t.dev[r], src, t.dev[src_rank], bytes, s));
        return 0;
    }

    int publish_slots(const int *send_slots, int n_slots)
    {
        Published &p = t.pub[r];
        for (int k = 0; k < n_slots; k++) {
            uint32_t eb;
            RCQ(cora_hip_shard_slot_info(job, send_slots[k], &p.n[k], &eb));
            RCQ(cora_hip_shard_slot_ptr(job, send_slots[k], &p.ptr[k]));       // (waits for the job's stream: the data is there)
        }
        return 0;
    }

    // all-to-all: counts[q] records of each send slot go to rank q; chunks land in sender (= position) order
    int pull_all_to_all(const int *recv_slots, const uint32_t *elem_bytes, int n_slots)
    {
        uint64_t total = 0;
        for (int q = 0; q < t.world; q++) total += t.pub[q].counts[r];
        for (int k = 0; k < n_slots; k++) {
            void *dst = nullptr;
            RCQ(cora_hip_shard_slot_reserve(job, recv_slots[k], total, &dst));
            uint64_t at = 0;
            for (int q = 0; q < t.world; q++) {
                uint64_t skip = 0;
                for (int z = 0; z < r; z++) skip += t.pub[q].counts[z];
                const uint64_t n = t.pub[q].counts[r];
                RCQ(copy_from(q, (char *)dst + at * elem_bytes[k], (const char *)t.pub[q].ptr[k] + skip * elem_bytes[k], n * elem_bytes[k]));
                at += n;
            }
        }
        HIPQ(hipStreamSynchronize(s));
        return 0;
    }

    bool all_to_all(int stage, int arg, const int *send_slots, const int *recv_slots, const uint32_t *elem_bytes, int n_slots)
    {
        TRY(cora_hip_shard_stage(job, stage, arg, t.pub[r].counts));
        return exchange(send_slots, recv_slots, elem_bytes, n_slots);
    }
    // the all-to-all itself, after a stage has left its per-destination counts in t.pub[r].counts
    bool exchange(const int *send_slots, const int *recv_slots, const uint32_t *elem_bytes, int n_slots)
    {
        TRY(publish_slots(send_slots, n_slots));
        if (!sync()) return false;
        TRY(pull_all_to_all(recv_slots, elem_bytes, n_slots));
        return sync();                                       // the senders' buffers are free again
    }

    // everything of every rank's send slots to rank 0, in rank order
    int pull_gather(const int *recv_slots, const uint32_t *elem_bytes, int n_slots)
    {
        for (int k = 0; k < n_slots; k++) {
            uint64_t total = 0;
            for (int q = 0; q < t.world; q++) total += t.pub[q].n[k];
            void *dst = nullptr;
            RCQ(cora_hip_shard_slot_reserve(job, recv_slots[k], total, &dst));
            uint64_t at = 0;
            for (int q = 0; q < t.world; q++) {
                RCQ(copy_from(q, (char *)dst + at * elem_bytes[k], t.pub[q].ptr[k], t.pub[q].n[k] * elem_bytes[k]));
                at += t.pub[q].n[k];
            }
        }
        HIPQ(hipStreamSynchronize(s));
        return 0;
    }

    bool gather_to_root(const int *send_slots, const int *recv_slots, const uint32_t *elem_bytes, int n_slots)
    {
        TRY(publish_slots(send_slots, n_slots));
        if (!sync()) return false;
        if (r == 0) TRY(pull_gather(recv_slots, elem_bytes, n_slots));
        return sync();
    }

    // element-wise combine (or / sum) of one u32 slot over all ranks; every rank ends up with the result imported
    DBuf<u32> acc, tmp;
    int combine_u32(int slot, bool is_or)
    {
        PoolScope scope(&job->pool);
        const uint64_t n = t.pub[r].n[0];
        for (int q = 0; q < t.world; q++) if (t.pub[q].n[0] != n) return fail(19, "ranks disagree about the size of slot %d", slot);
        RCQ(acc.alloc(n)); RCQ(tmp.alloc(n));
        RCQ(copy_from(r, acc.p, t.pub[r].ptr[0], n * sizeof(u32)));
        for (int q = 0; q < t.world; q++) {
            if (q == r) continue;
            RCQ(copy_from(q, tmp.p, t.pub[q].ptr[0], n * sizeof(u32)));
            if (is_or) launch_or_u32(s, acc.p, tmp.p, n); else launch_add_u32(s, acc.p, tmp.p, n);
        }
        HIPQ(hipStreamSynchronize(s));
        HIPQ(hipGetLastError());
        return 0;
    }
    void release_combine()
    {
        PoolScope scope(&job->pool);
        acc.release(); tmp.release();
    }

    bool all_reduce_u32(int slot, bool is_or)
    {
        const int slots[1] = {slot};
        TRY(publish_slots(slots, 1));
        if (!sync()) return false;
        TRY(combine_u32(slot, is_or));
        // every rank has read every published array before anyone overwrites its own with the result
        if (!sync()) { release_combine(); return false; }
        TRY(cora_hip_shard_slot_import(job, slot, acc.p, t.pub[r].n[0]));
        release_combine();
        return sync();
    }

    void run()
    {
        if (hipSetDevice(t.dev[r]) != hipSuccess) { fail(100, "hipSetDevice(%d) failed", t.dev[r]); fail_with(100); }
        const int world = t.world;
        static const uint32_t eb_rec[2] = {8, 4}, eb_parts[4] = {8, 4, 4, 1}, eb_code[1] = {8};
        static const int send_ac[2] = {CORA_SLOT_SEND_A, CORA_SLOT_SEND_C}, recv_ac[2] = {CORA_SLOT_RECV_A, CORA_SLOT_RECV_C};
        static const int send_a[1] = {CORA_SLOT_SEND_A}, recv_a[1] = {CORA_SLOT_RECV_A};
        static const int x_parts[4] = {CORA_SLOT_X_CLS, CORA_SLOT_X_CNT, CORA_SLOT_X_ENT, CORA_SLOT_X_DIR};
        static const int rx_parts[4] = {CORA_SLOT_RX_CLS, CORA_SLOT_RX_CNT, CORA_SLOT_RX_ENT, CORA_SLOT_RX_DIR};

        // CORA_TRACE: rank 0 prints the wall time of each part of the flow (the other ranks are inside the same barriers)
        const bool trace = r == 0 && getenv("CORA_TRACE");
        auto t_last = std::chrono::steady_clock::now();
        auto mark = [&](const char *what) {
            if (!trace) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[cora] ranks: %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        TRY(cora_hip_shard_begin(job, r, world));
        if (!t.failed.load()) shard_set_host_gather(job, true);
        if (!sync()) return;
        mark("pack");
        // exact: K-mer records to the owner of their hash range -- by all-to-all, or (2 to 4 ranks: shard.hip) every rank scans all
        // positions and keeps its own; the same choice on every rank (same world, same environment)
        int filter = 0;
        TRY(cora_hip_shard_filter(job, -1, &filter));
        if (filter) TRY(cora_hip_shard_stage(job, CORA_ST_KMERS, 0, nullptr));
        else if (!all_to_all(CORA_ST_KMERS, 0, send_ac, recv_ac, eb_rec, 2)) return;
        mark("k-mer records");
        TRY(cora_hip_shard_stage(job, CORA_ST_EXACT, 0, nullptr));
        TRY(cora_hip_shard_get_first_singleton(job, &t.pub[r].value));
        if (!sync()) return;
        {
            uint64_t m = ~0ull;
            for (int q = 0; q < world; q++) m = t.pub[q].value < m ? t.pub[q].value : m;
            TRY(cora_hip_shard_set_first_singleton(job, m));
        }
        mark("exact classes");
        if (!all_reduce_u32(CORA_SLOT_REPMASK, true)) return;              // later class members: disjoint bits
        mark("bitmap all-reduce");
        TRY(cora_hip_shard_stage(job, CORA_ST_PARTS, 0, nullptr));
        if (!gather_to_root(x_parts, rx_parts, eb_parts, 4)) return;
        mark("class parts to rank 0");
        if (r == 0) TRY(cora_hip_shard_stage(job, CORA_ST_MERGE, 0, nullptr));
        mark("exact table (rank 0)");
        // inexact: seed records to the owner of their code hash, verification there
        TRY(cora_hip_shard_stage(job, CORA_ST_REPS, 0, nullptr));
        // (large references: the records' sketches travel with them -- every rank takes the same decision from the reference's size)
        static const int send_ab[2] = {CORA_SLOT_SEND_A, CORA_SLOT_SEND_B}, recv_ab[2] = {CORA_SLOT_RECV_A, CORA_SLOT_RECV_B};
        static const uint32_t eb_code2[2] = {8, 8};
        for (int seed = 1; seed <= job->dp.nseeds; seed++) {
            if (filter) TRY(cora_hip_shard_stage(job, CORA_ST_SEED, seed, nullptr));
            else {
                // (the library decides whether the records carry sketches -- one rule, in shard.hip -- and says so after the stage)
                TRY(cora_hip_shard_stage(job, CORA_ST_SEED, seed, t.pub[r].counts));
                int sketched = 0;
                TRY(cora_hip_shard_seed_sketch(job, &sketched));
                if (!(sketched ? exchange(send_ab, recv_ab, eb_code2, 2) : exchange(send_a, recv_a, eb_code, 1))) return;
            }
            mark("seed records");
            TRY(cora_hip_shard_stage(job, CORA_ST_VERIFY, seed, nullptr));
            mark("seed sort + verification");
        }
        // lists are finished by the owner of their source position; intervals are cut where a position holds no list
        TRY(cora_hip_shard_stage(job, CORA_ST_ITEMS, 0, nullptr));
        if (!sync()) return;
        if (world > 1 && !all_reduce_u32(CORA_SLOT_WCOUNT, false)) return;
        mark("items + window counts");
        if (!all_to_all(CORA_ST_ITEM_CUT, 0, send_ac, recv_ac, eb_rec, 2)) return;
        mark("items to their owners");
        TRY(cora_hip_shard_stage(job, CORA_ST_FINISH, 0, nullptr));
        mark("lists + own table bytes");
        // Every rank's <Inexact> bytes are in ITS host buffer (shard_set_host_gather): the ranks share the address space, so each
        // copies its range straight into rank 0's table, all at once, instead of the bytes going up to the device, over to rank 0's
        // and down again through rank 0 alone (8 ranks sharing a GPU: 123-202 ms of the 0.94 s build).  Counts are summed on the host.
        {
            const cora_hip_stats &ms = job->stats;
            const uint64_t mine[8] = {ms.inexact_nodes, ms.dirty_nodes, ms.big_nodes, ms.inexact_items, ms.hom_pairs, ms.candidate_pairs, 0, 0};
            memcpy(t.pub[r].stats, mine, sizeof mine);
            t.pub[r].ptr[0] = (void *)job->inexact_bytes.p;
            t.pub[r].n[0] = job->inexact_bytes.n;
        }
        if (!sync()) return;
        if (r == 0 && !t.failed.load()) {
            uint64_t total = 0;
            for (int q = 0; q < world; q++) total += t.pub[q].n[0];
            ByteBuf &tb = job->inexact_bytes;
            // (first build of a job only: later ones find the room of the last table)
            if (!tb.grow_keep(total + 16)) { fail(101, "host allocation of %llu bytes for <Inexact> failed", (unsigned long long)total + 16); fail_with(101); }
            t.pub[0].ptr[1] = (void *)tb.p;
        }
        if (!sync()) return;
        if (r > 0 && t.pub[r].n[0]) {
            uint64_t at = 0;
            for (int q = 0; q < r; q++) at += t.pub[q].n[0];
            uint8_t *dst = (uint8_t *)t.pub[0].ptr[1] + at;
            const uint8_t *src = (const uint8_t *)t.pub[r].ptr[0];
            const size_t n = (size_t)t.pub[r].n[0];
            const int nt = std::max(1, std::min(8, job->host_threads / world));
            std::vector<std::thread> cp;
            for (int k = 1; k < nt; k++) cp.emplace_back([=] { memcpy(dst + n * k / nt, src + n * k / nt, n * (k + 1) / nt - n * k / nt); });
            memcpy(dst, src, n / nt);
            for (auto &x : cp) x.join();
        }
        if (!sync()) return;
        if (r == 0) {
            uint64_t sum[8] = {0}, total = 0;
            for (int q = 0; q < world; q++) { total += t.pub[q].n[0]; for (int k = 0; k < 8; k++) sum[k] += t.pub[q].stats[k]; }
            job->inexact_bytes.n = total;
            cora_hip_stats &st = job->stats;
            st.inexact_nodes = sum[0]; st.dirty_nodes = sum[1]; st.big_nodes = sum[2]; st.inexact_items = sum[3]; st.hom_pairs = sum[4];
            st.candidate_pairs = sum[5];
        }
        sync();
        mark("table bytes to rank 0");
    }
#undef TRY
};

}  // namespace

extern "C" int cora_hip_device_count(int *n)
{
    if (!n) return fail(19, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) return fail(102, "no HIP device available (%s)", hipGetErrorString(e));
    *n = c;
    return 0;
}

extern "C" int cora_hip_job_run_multi(cora_hip_job *job, int world, const int *devices)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    if (world < 1 || world > MAX_WORLD) return fail(19, "bad number of ranks %d (1..%d)", world, MAX_WORLD);
    // the arena of an earlier build of this job over another number of ranks (DevPool; e.g. a single-GPU run's, most of the
    // device) goes back: ranks may share this GPU and need its memory, and a rank's arena is sized for its share of the records
    if (job->pool.arena && job->arena_world != world) {
        { PoolScope scope(&job->pool); shard_state_free(job); }
        job->pool.release_arena();
    }
    if (!job->pool.arena) job->arena_tried = false;
    int ndev = 0;
    HIPQ(hipGetDeviceCount(&ndev));
    std::vector<int> dev(world);
    for (int r = 0; r < world; r++) {
        dev[r] = devices ? devices[r] : r % ndev;
        if (dev[r] < 0 || dev[r] >= ndev) return fail(102, "rank %d: device %d out of range (have %d)", r, dev[r], ndev);
    }
    if (dev[0] != job->device) return fail(19, "rank 0 must run on the job's own device %d", job->device);
    // the other ranks' jobs: same parameters, the reference image copied from the caller's job (device to device)
    if (job->multi && (job->multi->devices != dev)) multi_state_free(job);
    if (!job->multi) {
        // built aside and handed to the job only once every clone exists: a job must never keep a half-built team (a later call
        // with the same devices would skip this block and run ranks on null clones -- round-2 advice)
        MultiState *m = new MultiState();
        m->devices = dev;
        m->clones.assign(world, nullptr);
        struct Guard {                                         // every early return below destroys what exists so far
            MultiState *m; int home;
            ~Guard()
            {
                if (!m) return;
                for (cora_hip_job *c : m->clones) if (c) { (void)hipSetDevice(c->device); cora_hip_job_destroy(c); }
                (void)hipSetDevice(home);
                delete m;
            }
        } guard{m, job->device};
        HIPQ(hipStreamSynchronize(job->stream));
        for (int r = 1; r < world; r++) {
            HIPQ(hipSetDevice(dev[r]));
            hipDeviceProp_t prop;
            HIPQ(hipGetDeviceProperties(&prop, dev[r]));
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(102, "device %d is %s; the kernels are built for gfx950 only", dev[r], prop.gcnArchName);
            if (dev[r] != job->device) {
                int can = 0;
                HIPQ(hipDeviceCanAccessPeer(&can, dev[r], job->device));
                if (!can) return fail(102, "device %d cannot access device %d peer to peer", dev[r], job->device);
                for (int q = 0; q < world; q++) {
                    if (dev[q] == dev[r]) continue;
                    const hipError_t e = hipDeviceEnablePeerAccess(dev[q], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(100, "hipDeviceEnablePeerAccess(%d -> %d): %s", dev[r], dev[q], hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            }
            cora_hip_job *c = nullptr;
            int rc = cora_hip_job_create(&job->prm, &c);
            if (!rc) {
                m->clones[r] = c;                              // (the guard owns it from here)
                c->sketch_mode = job->sketch_mode; c->hash_bits = job->hash_bits; c->host_threads = job->host_threads;
                rc = set_reference_impl(c, job->ref_contigs.data(), (int32_t)job->ref_contigs.size(), job->ref_W, job->ref_image_bytes, nullptr, job->raw.p, nullptr);
            }
            // CORA_TEST_FAIL_CLONE=r: tests make the set-up of rank r fail here, as a device without memory would
            if (!rc && getenv("CORA_TEST_FAIL_CLONE") && atoi(getenv("CORA_TEST_FAIL_CLONE")) == r) rc = fail(101, "set-up of rank %d failed (CORA_TEST_FAIL_CLONE)", r);
            if (rc) return rc;
        }
        HIPQ(hipSetDevice(job->device));
        for (int q = 1; q < world; q++) {                     // the caller's device reads its peers as well
            if (dev[q] == job->device) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(dev[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(100, "hipDeviceEnablePeerAccess(%d -> %d): %s", job->device, dev[q], hipGetErrorString(e));
            (void)hipGetLastError();
        }
        guard.m = nullptr;                                    // complete: the job takes it
        job->multi = m;
    }
    for (int r = 0; r < world; r++) {                        // ranks that share a device share its memory (reserve_job_arena)
        int share = 0;
        for (int q = 0; q < world; q++) share += dev[q] == dev[r];
        (r == 0 ? job : job->multi->clones[r])->ranks_on_device = share;
    }
    Team t;
    t.world = world;
    t.dev = dev;
    t.jobs.assign(world, nullptr);
    t.jobs[0] = job;
    for (int r = 1; r < world; r++) t.jobs[r] = job->multi->clones[r];
    t.pub.assign(world, Published());
    t.rc.assign(world, 0);
    t.err.assign(world, std::string());
    if (pthread_barrier_init(&t.bar, nullptr, (unsigned)world) != 0) return fail(100, "pthread_barrier_init failed");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int r = 1; r < world; r++) th.emplace_back([&t, r]() { Rank(t, r).run(); });
    Rank(t, 0).run();
    for (auto &x : th) x.join();
    pthread_barrier_destroy(&t.bar);
    HIPQ(hipSetDevice(job->device));
    for (int r = 0; r < world; r++)
        if (t.rc[r]) return fail(t.rc[r], "rank %d of %d (device %d): %s", r, world, dev[r], t.err[r].c_str());
    // whole-job figures: K-mers and representatives are per-rank counts (every record is owned by one rank)
    cora_hip_stats &st = job->stats;
    for (int r = 1; r < world; r++) {
        const cora_hip_stats &o = t.jobs[r]->stats;
        st.kmers += o.kmers; st.representatives += o.representatives; st.exact_classes += o.exact_classes;
        st.hash_mixed_runs += o.hash_mixed_runs; st.full_key_fallback |= o.full_key_fallback;
    }
    st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
