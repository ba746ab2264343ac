// job.hip -- C-ABI of libcora_hip.so (include/cora_hip.h) and the stage orchestration of one
// table build on one MI355X.  HTS = /root/reference/homTable_setup.cpp.
#include "job_internal.hpp"
#include <mutex>

using namespace cora;

// representatives above which the seed records carry sketches (measured: stages.hip, st_seed_verify)
constexpr u64 SKETCH_MIN_REPS = 1ull << 28;

extern "C" const char *cora_hip_last_error(void) { return g_err.c_str(); }

static int scratch_ptr_of(int device, void **out);

extern "C" int cora_hip_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(102, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(102, "device %d out of range (have %d)", device, n);
    HIPQ(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPQ(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(102, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
    g_device = device;
    void *sp = nullptr;
    RCQ(scratch_ptr_of(device, &sp));          // static scratch of cora_hip_radix_sort_noalloc: its address on this GPU
    return 0;
}

extern "C" int cora_hip_shutdown(void)
{
    g_device = -1;
    return 0;
}

extern "C" int cora_hip_job_create(const cora_hip_params *params, cora_hip_job **out)
{
    if (!params || !out) return fail(19, "null argument");
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (params->K < 33 || params->K > 64) return fail(65, "read-mer length %d outside 33..64", params->K);                 // HTS:2418-2424
    if (params->d < 1 || params->d > 3) return fail(6, "number of mismatches %d outside 1..3", params->d);                 // HTS:2017-2021
    if (params->line_width < 1) return fail(51, "bad FASTA line width %d", params->line_width);
    cora_hip_job *j = new cora_hip_job();
    j->prm = *params;
    if (hipGetDevice(&j->device) != hipSuccess) { delete j; return fail(100, "hipGetDevice failed"); }
    memset(&j->stats, 0, sizeof j->stats);
    HostGenome &hg = j->hg;
    hg.K = params->K; hg.d = params->d; hg.chr_short = params->chr_short;
    hg.n_groups = params->n_groups; hg.n_threads = params->n_threads;
    memcpy(hg.sig2, params->sig2, 16);
    memcpy(hg.sig4, params->sig4, 256);
    make_seed_template(hg.K, hg.d, hg.tmpl, &hg.seed_q);
    if (const char *e = getenv("CORA_SKETCH")) j->sketch_mode = atoi(e) != 0;
    if (const char *e = getenv("CORA_HASH_BITS")) { int v = atoi(e); if (v >= 8 && v <= 56) j->hash_bits = v & ~7; }
    {   // host threads of the table finish (serialising the lists with tied targets, moving the byte stream into place)
        const unsigned hc = std::thread::hardware_concurrency();
        j->host_threads = hc ? (int)(hc < 16u ? hc : 16u) : 8;
    }
    if (const char *e = getenv("CORA_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) j->host_threads = v; }
    if (hipStreamCreate(&j->stream) != hipSuccess) { delete j; return fail(100, "hipStreamCreate failed"); }
    for (auto &e : j->ev) if (hipEventCreate(&e) != hipSuccess) { delete j; return fail(100, "hipEventCreate failed"); }
    if (hipStreamCreateWithFlags(&j->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete j; return fail(100, "hipStreamCreate failed"); }
    for (auto &e : j->ev_copy) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete j; return fail(100, "hipEventCreate failed"); }
    *out = j;
    return 0;
}

int job_side_context(cora_hip_job *job, cora_hip_job **out)
{
    cora_hip_job *c = nullptr;
    RCQ(cora_hip_job_create(&job->prm, &c));
    c->device = job->device;
    c->sketch_mode = job->sketch_mode; c->hash_bits = job->hash_bits; c->host_threads = job->host_threads;
    if (c->ctr.alloc(1)) { cora_hip_job_destroy(c); return fail(101, "device allocation failed"); }
    *out = c;
    return 0;
}

extern "C" void cora_hip_job_destroy(cora_hip_job *job)
{
    if (!job) return;
    join_warm(job);
    join_exact_writer(job);
    multi_state_free(job);
    (void)hipSetDevice(job->device);
    shard_state_free(job);
    job->sort_scratch.release();
    job->raw.release(); job->d_gstart.release(); job->d_off.release(); job->d_len.release(); job->d_window.release();
    job->pk.release(); job->bad.release(); job->ctr.release();
    job->pool.destroy();
    for (auto &e : job->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : job->ev_copy) if (e) (void)hipEventDestroy(e);
    if (job->copy_stream) { (void)hipStreamSynchronize(job->copy_stream); (void)hipStreamDestroy(job->copy_stream); }
    if (job->stream) (void)hipStreamDestroy(job->stream);
    delete job;
}

extern "C" int cora_hip_job_set_reference(cora_hip_job *job, const uint8_t *image, size_t image_bytes,
                                          const cora_hip_contig *contigs, int32_t n_contigs)
{
    if (!job || !image || !contigs || n_contigs <= 0) return fail(51, "bad reference arguments");
    return set_reference_impl(job, contigs, n_contigs, (u32)job->prm.line_width, image_bytes, image, nullptr, nullptr);
}

// The image comes from host memory (copied), from device memory (copied), or as a device buffer that the job takes over.
int set_reference_impl(cora_hip_job *job, const cora_hip_contig *contigs, int32_t n_contigs, u32 W, size_t image_bytes,
                       const uint8_t *host_image, const uint8_t *dev_image, DBuf<u8> *adopt)
{
    if (!job || !contigs || n_contigs <= 0 || W < 1) return fail(51, "bad reference arguments");
    if (!job->prm.chr_short && n_contigs > 127)
        return fail(19, "%d contigs need the 2-byte layout (homTable_setup_L; cora.cpp:25,488-493)", n_contigs);
    if (n_contigs > 65530) return fail(19, "too many contigs (%d)", n_contigs);                                            // HTS:23-27
    join_warm(job);
    if (job->packed_in_flight) { HIPQ(hipStreamSynchronize(job->copy_stream)); job->packed_in_flight = false; }
    HostGenome &hg = job->hg;
    hg.gstart.assign(n_contigs + 1, 0);
    hg.len.assign(n_contigs, 0);
    std::vector<u64> off(n_contigs);
    u64 g = 0, bases = 0;
    for (int c = 0; c < n_contigs; c++) {
        const u64 L = contigs[c].length;
        if (L <= (u64)job->prm.K)
            return fail(54, "contig %d has %llu bases: no findable %d-mer, the reference's behaviour is undefined (HTS:2530)", c + 1,
                        (unsigned long long)L, job->prm.K);
        const u64 last = contigs[c].file_offset + (L - 1) + (L - 1) / W;      // last base of the contig in the image
        if (last >= image_bytes) return fail(51, "contig %d runs past the end of the FASTA image", c + 1);
        hg.gstart[c] = g; hg.len[c] = (u32)L; off[c] = contigs[c].file_offset;
        g = (g + L + 1 + 31) & ~31ull;                                 // at least one pad base, 32-base aligned starts
        bases += L;
    }
    const u64 G = (g + 63) & ~63ull;
    if (G + 64 >= (1ull << 32)) return fail(19, "reference too long for 32-bit global coordinates (%llu padded bases)", (unsigned long long)G);
    hg.gstart[n_contigs] = G;
    plan_windows(hg.d, hg.len, hg.window);
    job->genome_bases = bases;
    job->ref_contigs.assign(contigs, contigs + n_contigs);
    job->ref_W = W; job->ref_image_bytes = image_bytes;
    multi_state_free(job);                                         // replicas of an earlier reference are stale

    if (adopt) {
        job->raw.release();
        job->raw.p = adopt->p; job->raw.n = adopt->n; job->raw.pool = adopt->pool;
        adopt->p = nullptr; adopt->n = 0;
    } else {
        RCQ(job->raw.alloc(image_bytes + 16));
        if (host_image) HIPQ(hipMemcpyAsync(job->raw.p, host_image, image_bytes, hipMemcpyHostToDevice, job->stream));
        else HIPQ(hipMemcpyAsync(job->raw.p, dev_image, image_bytes, hipMemcpyDeviceToDevice, job->stream));
    }
    RCQ(job->d_gstart.alloc(n_contigs + 1));
    RCQ(job->d_off.alloc(n_contigs));
    RCQ(job->d_len.alloc(n_contigs));
    RCQ(job->d_window.alloc(n_contigs));
    HIPQ(hipMemcpyAsync(job->d_gstart.p, hg.gstart.data(), (n_contigs + 1) * sizeof(u64), hipMemcpyHostToDevice, job->stream));
    HIPQ(hipMemcpyAsync(job->d_off.p, off.data(), n_contigs * sizeof(u64), hipMemcpyHostToDevice, job->stream));
    HIPQ(hipMemcpyAsync(job->d_len.p, hg.len.data(), n_contigs * sizeof(u32), hipMemcpyHostToDevice, job->stream));
    HIPQ(hipMemcpyAsync(job->d_window.p, hg.window.data(), n_contigs * sizeof(u32), hipMemcpyHostToDevice, job->stream));

    DevParams &dp = job->dp;
    memset(&dp, 0, sizeof dp);
    dp.K = hg.K; dp.d = hg.d; dp.nseeds = hg.d + 1; dp.seed_q = hg.seed_q;
    dp.line_width = W; dp.n_contigs = (u32)n_contigs; dp.G = G; dp.n_words = G / 32; dp.image_bytes = image_bytes;
    dp.kmask_lo = (hg.K == 64) ? ~0ull : ~((~0ull) >> (2 * (hg.K - 32)));
    for (int i = 0; i < hg.K; i++) {
        const int s = hg.tmpl[i];
        if (!s) continue;
        const u64 bit = 1ull << (62 - 2 * (i & 31));
        if (i < 32) dp.seedmask_hi[s - 1] |= bit; else dp.seedmask_lo[s - 1] |= bit;
    }
    dp.any_windowed = 0;
    for (u32 wl : hg.window) if (wl) dp.any_windowed = 1;
    memcpy(dp.sig2, hg.sig2, 16);
    dp.ctg_gstart = job->d_gstart.p; dp.ctg_len = job->d_len.p; dp.ctg_off = job->d_off.p; dp.ctg_window = job->d_window.p;
    RCQ(job->pk.alloc(dp.n_words + 4));
    RCQ(job->bad.alloc(dp.n_words + 4));
    RCQ(job->valid.alloc(dp.n_words + 12));
    RCQ(job->ctr.alloc(1));
    HIPQ(hipStreamSynchronize(job->stream));
    job->have_ref = true;
    return 0;
}

// A large reference: the stage arrays are cut from one allocation (DevPool: why).  One GPU: 80 B per base hold what a d = 2
// build has in use at once (182 GB at 3.09 Gbp, the seed sorts' chain of five second-word arrays included) with the 1 GB
// alignment of its big arrays; the rest comes from separate blocks as before.  A rank of a sharded build holds 1 / world of the
// records (80 B per base / world) and, when every rank scans all positions and keeps its own (keep_own: 2 to 4 GPUs), the
// records of ALL positions of one stage while its sort's first pass reads them (16 B per base for a seed with sketches).
// CORA_ARENA=0: no arena; CORA_ARENA_GB=n: its size.
void reserve_job_arena(cora_hip_job *job, int world, bool keep_own)
{
    if (job->pool.arena || job->arena_tried || job->genome_bases < (256ull << 20)) return;
    job->arena_tried = true;
    const DevParams &dp = job->dp;
    size_t want = (size_t)dp.G * 80 / (size_t)world + (keep_own ? (size_t)dp.G * 20 : 0) + ((size_t)6 << 30), free_b = 0, total_b = 0;
    // At most 190 GiB (CORA_ARENA_CAP_GB=n: n).  Giving back an arena that reaches into the last third of the device's memory
    // is slow -- 16 ms for 190 GiB, 0.3 s for 210, 1.15 s for 236 (profiles/e2e_arena.py) -- which a job that builds many
    // tables never notices and a process that builds one pays at its end.  Round 2 capped the executables only and let
    // in-process jobs take 236 GiB (seeds stage 407 instead of 414 ms: a chain of five arrays instead of three); now every
    // build has the one layout, so that the benchmark's step and the executable's build are the same build.
    const size_t cap = (size_t)(getenv("CORA_ARENA_CAP_GB") ? atoll(getenv("CORA_ARENA_CAP_GB")) : 190) << 30;
    if (want > cap) want = cap;
    if (getenv("CORA_ARENA_GB")) want = (size_t)atoll(getenv("CORA_ARENA_GB")) << 30;
    if (getenv("CORA_ARENA") && atoi(getenv("CORA_ARENA")) == 0) return;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b <= ((size_t)48 << 30)) return;
    // ranks that share the device (rehearsals) share what is free; each leaves room for the others and for what no arena holds
    const size_t share = (size_t)(job->ranks_on_device > 0 ? job->ranks_on_device : 1);
    const size_t room = share > 1 ? (total_b - ((size_t)24 << 30)) / share - ((size_t)6 << 30) : free_b - ((size_t)16 << 30);
    if (share > 1 && (room > free_b || room < ((size_t)8 << 30))) return;
    if (want > room) want = room;
    const auto a0 = std::chrono::steady_clock::now();
    // (the digit bytes of the sorts, one per record of the largest sort, in the allocation's tail: DevPool::tail)
    const size_t digit_bytes = (size_t)dp.G / (size_t)world + (size_t)dp.G / 8 + ((size_t)4 << 20);
    const bool ok = job->pool.reserve_arena(want, digit_bytes);
    if (ok) job->arena_world = world;
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] arena: hipMalloc took %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count());
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] arena of %.1f GB (%d rank(s)): %s\n", want / 1073741824.0, world, ok ? "reserved" : "not available, separate blocks");
}

extern "C" int cora_hip_job_run(cora_hip_job *job)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    PoolScope pool_scope(&job->pool);
    shard_state_free(job);                                 // (what a sharded run of this job left, e.g. one that failed half-way)
    join_exact_writer(job);
    job->exact_written = false;
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    cora_hip_stats &st = job->stats;
    memset(&st, 0, sizeof st);
    st.genome_bases = job->genome_bases;
    job->sort_timing = SortTiming();
    job->sort_timing.enabled = true;
    const auto t_begin = std::chrono::steady_clock::now();
    double host_ms = 0;
    reserve_job_arena(job, 1, false);
    HIPQ(hipEventRecord(job->ev[0], s));

    // Which end of the arena the big arrays come from: the K-mer records and their sort take the high end, everything after them
    // the low end.  Measured on two boxes, alternating (profiles/arena_end_ab.sh): exact stage 133.3 ms from the high end against
    // 141.0 from the low end, seeds stage 424.3 against 422.0 -- pieces of device memory differ in how fast they take scattered
    // writes ("Where the arrays lie"), and the 66 GB of the first sort fit the better part, the 95 GB of a seed's do not.
    // CORA_ARENA_HIGH=0 / 1: the low / the high end throughout.
    static const int arena_end = getenv("CORA_ARENA_HIGH") ? atoi(getenv("CORA_ARENA_HIGH")) : -1;
    job->pool.ranges.big_from_high = arena_end != 0;
    // ---- stage 1: pack, hashed K-mer records (HTS:2499-2565) ----
    join_warm(job);                                                                                                 // (a run that failed half-way)
    if (job->packed_in_flight) { HIPQ(hipStreamSynchronize(job->copy_stream)); job->packed_in_flight = false; }
    RCQ(st_pack(job));
    const size_t packed_words = dp.n_words + 4;
    if (job->h_packed.cap < packed_words && !job->h_stream.cap && !job->inexact_bytes.p && job->genome_bases >= (256ull << 20)) {
        // The first build of a job on a large reference pays for its host buffers: pinning the landing areas of the packed
        // reference and of the byte stream, the first CPU touch of every pinned page (the copy threads of assemble_inexact
        // moved 14 GB/s out of a fresh h_stream against 67 GB/s later) and the page faults of a fresh <Inexact> buffer --
        // 100 of the 130 ms such a build took longer than the following ones.  None of them is needed before the lists are
        // finished 0.6 s from now, so a helper thread prepares them beside the kernels; the byte stream's size is a guess
        // (0.4 B per base; 0.27 on the GRCh38-like workload: a wrong guess costs memory or, too small, what it cost before).
        // The packed reference then travels when st_items_finish asks for it (15 ms) instead of right now.
        const size_t est = (size_t)(job->genome_bases / 5 * 2);
        const int dev = job->device;
        HIPQ(hipEventRecord(job->ev_copy[EV_PACKED_READY], s));          // (k_pack has been launched: the packed reference may leave behind it)
        // the landing area and the host buffer of <Exact> in a thread of their own: the exact stage is through 0.15 s from now, and the
        // device waited 31 ms for these two in a first build (0.02 B per base on the GRCh38-like workload; 1/32 is reserved)
        job->warm_exact = std::thread([job, dev] {
            if (hipSetDevice(dev) != hipSuccess) return;
            const size_t est_exact = (size_t)(job->genome_bases / 32) + (1u << 20);
            const auto w0 = std::chrono::steady_clock::now();
            if (!job->h_exact.reserve(est_exact)) { uint8_t x = 0; for (size_t o = 0; o < est_exact; o += 4096) x ^= ((const volatile uint8_t *)job->h_exact.p)[o]; (void)x; }
            if (job->exact_bytes.reserve_uninit(est_exact))
                for (size_t o = 0; o < est_exact; o += 4096) job->exact_bytes.p[o] = 0;
            if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] warm: <Exact> buffers ready after %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
        });
        job->warm = std::thread([job, est, dev, packed_words] {
            if (hipSetDevice(dev) != hipSuccess) return;
            auto touch = [](const volatile uint8_t *p, size_t n) { uint8_t x = 0; for (size_t o = 0; o < n; o += 4096) x ^= p[o]; return x; };
            // (the table's pages in a thread of their own: when the buffer is the table's file -- cora_hip_job_set_table_files -- their
            //  first touch takes 90 ms per 0.9 GB, in ONE thread: several threads touching one file's pages are slower, profiles/io_probe.cpp)
            static const bool trace = getenv("CORA_TRACE") != nullptr;
            const auto w0 = std::chrono::steady_clock::now();
            auto since = [w0] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
            std::thread table([job, est, since] {
                if (job->inexact_bytes.reserve_uninit(est))
                    for (size_t o = 0; o < est; o += 4096) job->inexact_bytes.p[o] = 0;
                if (trace) fprintf(stderr, "[cora] warm: table pages touched after %.1f ms\n", since());
            });
            // (the byte stream's landing area, the largest, in a thread of its own)
            std::thread stream_area([job, est, touch, since] {
                if (!job->h_stream.reserve(est)) (void)touch(job->h_stream.p, est);
                if (trace) fprintf(stderr, "[cora] warm: byte-stream landing area pinned after %.1f ms\n", since());
            });
            if (!job->h_packed.reserve(packed_words)) {
                (void)touch((const uint8_t *)job->h_packed.p, packed_words * sizeof(u64));
                // ... and the packed reference leaves for it at once, on the copy stream (nobody else uses that stream before
                // st_items_finish has joined this thread): asked for by st_items_finish it took 15 ms of the items stage
                if (hipStreamWaitEvent(job->copy_stream, job->ev_copy[EV_PACKED_READY], 0) == hipSuccess &&
                    hipMemcpyAsync(job->h_packed.p, job->pk.p, packed_words * sizeof(u64), hipMemcpyDeviceToHost, job->copy_stream) == hipSuccess &&
                    hipEventRecord(job->ev_copy[EV_PACKED_ARRIVED], job->copy_stream) == hipSuccess)
                    job->packed_in_flight = true;          // (read after join_warm)
                else (void)hipGetLastError();
            }
            if (trace) fprintf(stderr, "[cora] warm: packed-reference landing area pinned after %.1f ms\n", since());
            // the landing areas of the lists the host finishes (tied targets): one list per 2048 bases and one item per 512 are
            // four times what the GRCh38-like workload needs; pinned here, st_items_finish finds them ready (2-3 ms of hipHostMalloc each)
            const size_t lists = (size_t)(job->genome_bases / 2048) + 1024, items = (size_t)(job->genome_bases / 512) + 4096;
            (void)job->h_hnodes.reserve(lists); (void)job->h_node_start.reserve(lists + 1);
            (void)job->h_key.reserve(items); (void)job->h_meta.reserve(items); (void)job->h_blen.reserve(items);
            if (trace) fprintf(stderr, "[cora] warm: host-list landing areas pinned after %.1f ms\n", since());
            stream_area.join();
            table.join();
        });
        if (getenv("CORA_WARM_SYNC")) join_warm(job);          // (diagnostic: the buffers first, then the kernels)
    } else {
        // the packed reference starts its way to the host now (0.25 B per base over PCIe beside 0.8 s of kernels): lists
        // with tied targets are finished there and read their K-mers from it (st_items_finish)
        RCQ(job->h_packed.reserve(packed_words));
        HIPQ(hipEventRecord(job->ev_copy[EV_PACKED_READY], s));
        HIPQ(hipStreamWaitEvent(job->copy_stream, job->ev_copy[EV_PACKED_READY], 0));
        HIPQ(hipMemcpyAsync(job->h_packed.p, job->pk.p, packed_words * sizeof(u64), hipMemcpyDeviceToHost, job->copy_stream));
        HIPQ(hipEventRecord(job->ev_copy[EV_PACKED_ARRIVED], job->copy_stream));
        job->packed_in_flight = true;
    }
    // (the digit bytes of every sort of this run, lent to the sort scratch from the pool: no sort here holds more than G records)
    NextDigitLease digit_bytes(job->sort_scratch, dp.G);
    ExactLocal x;
    RCQ(st_kmers(job, 0, (u32)((dp.G + 255) / 256), x.rec, x.gp, x.N, &x.first_digit_ready));
    st.kmers = x.N;
    HIPQ(hipEventRecord(job->ev[1], s));
    // every contig must own at least one window (else the reference reads before its buffer, HTS:2530)
    {
        launch_contig_check(s, dp, x.gp.p, x.N, job->ctr.p);
        DevCounters h;
        RCQ(read_counters(job, h));
        FORMAT_ERROR_CHECK(h);
        if (h.empty_contigs)
            return fail(54, "%llu contig(s) hold no findable %d-mer: the reference's behaviour is undefined there (HTS:2530)",
                        (unsigned long long)h.empty_contigs, dp.K);
    }

    // ---- stage 2: sort, classes, representatives, exact table (HTS:2516-2578, 249-635) ----
    FinishStats fst;
    DBuf<u32> reps;
    u64 NR = 0;
    {
        RCQ(st_exact_classes(job, x));
        st.representatives = x.NR; st.exact_classes = x.NNS;
        launch_rep_mask_words(s, job->valid.p, x.nonrep.p, x.nonrep.p, dp.n_words);   // in place: now the representative bitmap
        RCQ(st_reps(job, x.nonrep.p, 0, dp.n_words, reps, NR));
        ExactParts parts;
        RCQ(st_exact_parts(job, x, parts));
        x.release_all();
        HIPQ(hipEventRecord(job->ev[2], s));
        RCQ(st_exact_finish(job, parts, fst, host_ms));
        if (!job->exact_path.empty()) {                    // <Exact> goes to its file now, beside the inexact stages (9 ms of fwrite for 68 MB)
            job->exact_write_failed = false;
            job->exact_writer = std::thread([job] {
                FILE *f = fopen(job->exact_path.c_str(), "wb");
                const size_t n = job->exact_bytes.size();
                const bool ok = f && (n == 0 || fwrite(job->exact_bytes.data(), 1, n, f) == n);
                if (!(f && fclose(f) == 0 && ok)) job->exact_write_failed = true;
            });
            job->exact_written = true;
        }
    }

    job->pool.ranges.big_from_high = arena_end == 1;
    // ---- stage 3: per seed: codes, sort, equal runs, verification (HTS:2671-2800, 2138-2265) ----
    PairStore ps;
    {
        // every seed's codes in one pass over the representatives, each with its first-digit tile histogram
        DBuf<u64> codes[CORA_HIP_MAX_SEEDS], skb[CORA_HIP_MAX_SEEDS];
        DBuf<u32> hists[CORA_HIP_MAX_SEEDS], skc[CORA_HIP_MAX_SEEDS];
        u64 *cp[4] = {nullptr, nullptr, nullptr, nullptr};
        u32 *hp[4] = {nullptr, nullptr, nullptr, nullptr};
        SeedSketches sk;
        memset(&sk, 0, sizeof sk);
        // Chance collisions of 32-bit seed codes grow with NR^2 / 2^32: from a few hundred million representatives on they
        // dominate verification, and carrying the other seeds' codes through the sort (4 d bytes per record) pays for
        // itself: the candidate kernels then dismiss partners from the sorted records alone, without touching the genome.
        const bool use_sketch = job->sketch_mode >= 0 ? job->sketch_mode != 0 : NR > SKETCH_MIN_REPS;
        sk.on = use_sketch;
        const u32 tile_recs = radix_sort_tile_records(use_sketch && dp.nseeds >= 3, use_sketch && dp.nseeds != 3);
        const u64 tiles = (NR + tile_recs - 1) / tile_recs;
        // With sketches (large references) the records of one seed are made, sorted and verified before the next seed's are
        // made: all seeds at once took 16 (d + 1) bytes per representative on top of the sort's spare buffers -- 242 GB
        // in use at once on a GRCh38-sized reference.  Without them every seed's codes come from one pass.
        const bool per_seed = use_sketch;
        // The sketches' second word climbs through a chain of arrays during its sort (radix_sort.hip, "the second word
        // climbs": seeds stage 424 -> 414 ms with 3 arrays, 407 ms with 5 at 2.6e9 representatives) when the arena has the
        // room: 5 arrays want 70 B of arena per base, 3 want 56.  CORA_B_CHAIN=n: that many (1: a ping-pong, as without arena).
        static const int b_chain_env = getenv("CORA_B_CHAIN") ? atoi(getenv("CORA_B_CHAIN")) : 0;
        int b_chain = 1;
        if (per_seed && job->pool.arena) {
            b_chain = job->pool.arena_bytes >= (size_t)dp.G * 70 ? 5 : job->pool.arena_bytes >= (size_t)dp.G * 56 ? 3 : 1;
            if (b_chain_env >= 1 && b_chain_env <= 5) b_chain = b_chain_env;
        }
        const bool lds = seed_lds_finish(NR);
        auto make = [&](int q) -> int {
            RCQ(codes[q].alloc(NR)); RCQ(hists[q].alloc(tiles * 256));
            cp[q] = codes[q].p; hp[q] = hists[q].p;
            if (use_sketch && dp.nseeds >= 3) {
                // (no room for the chain: a shorter one, at last a ping-pong, from here on)
                while (b_chain > 1 && skb[q].alloc(NR * (u64)b_chain)) { b_chain = b_chain > 3 ? 3 : 1; (void)hipGetLastError(); }
                if (b_chain == 1) RCQ(skb[q].alloc(NR));
                sk.b[q] = skb[q].p;
            }
            if (use_sketch && dp.nseeds != 3) { RCQ(skc[q].alloc(NR)); sk.c[q] = skc[q].p; }
            return 0;
        };
        if (!per_seed) {
            for (int q = 0; q < dp.nseeds; q++) RCQ(make(q));
            launch_seed_codes_all(s, dp, job->pk.p, reps.p, NR, cp, hp, sk, tile_recs, -1, seed_first_digit_shift(lds));
        }
        for (int seed = 1; seed <= dp.nseeds && NR; seed++) {
            if (per_seed) {
                RCQ(make(seed - 1));
                launch_seed_codes_all(s, dp, job->pk.p, reps.p, NR, cp, hp, sk, tile_recs, seed - 1, seed_first_digit_shift(lds));
            }
            RCQ(st_seed_verify(job, codes[seed - 1], NR, seed, ps, hists[seed - 1].p, sk.b[seed - 1], sk.c[seed - 1], nullptr, b_chain, lds));
            codes[seed - 1].release(); hists[seed - 1].release(); skb[seed - 1].release(); skc[seed - 1].release();
            cp[seed - 1] = nullptr; hp[seed - 1] = nullptr; sk.b[seed - 1] = nullptr; sk.c[seed - 1] = nullptr;
        }
        st.hom_pairs = ps.n;
    }
    reps.release();
    HIPQ(hipGetLastError());
    HIPQ(hipEventRecord(job->ev[3], s));

    // ---- stage 4 + 5: directed items, lists, block merge, final order and bytes (HTS:1022-1432) ----
    {
        const u64 M = 2 * ps.n;
        if (M >= (1ull << 32)) return fail(19, "too many directed items (%llu)", (unsigned long long)M);
        DBuf<u64> ik;
        DBuf<u32> im;
        RCQ(ik.alloc(M)); RCQ(im.alloc(M));
        launch_items_expand(s, dp, ps.pairs.p, ps.n, ik.p, im.p);
        HIPQ(hipStreamSynchronize(s));
        ps.pairs.release();
        RCQ(st_items_finish(job, ik, im, M, fst, host_ms));
    }
    if (job->packed_in_flight) { HIPQ(hipStreamSynchronize(job->copy_stream)); job->packed_in_flight = false; }
    st.exact_nodes = fst.exact_nodes; st.inexact_nodes = fst.inexact_nodes;
    st.dirty_nodes = fst.dirty_nodes; st.big_nodes = fst.big_nodes;
    (void)hipEventElapsedTime(&st.ms_pack, job->ev[0], job->ev[1]);
    (void)hipEventElapsedTime(&st.ms_exact, job->ev[1], job->ev[2]);
    (void)hipEventElapsedTime(&st.ms_seeds, job->ev[2], job->ev[3]);
    (void)hipEventElapsedTime(&st.ms_items, job->ev[3], job->ev[4]);
    st.ms_host_finish = (float)host_ms;
    st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    st.ms_sort_scatter = job->sort_timing.scatter_ms;
    st.n_sort_scatter = job->sort_timing.scatter_launches;
    st.sort_scatter_bytes = job->sort_timing.scatter_bytes;
    st.pool_peak_bytes = job->pool.peak_live;
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] stages: pack+records %.3f exact %.3f seeds %.3f items %.3f ms (device), host %.3f ms\n", st.ms_pack, st.ms_exact, st.ms_seeds,
                                      st.ms_items, st.ms_host_finish);
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] run %.3f ms; device pool: %zu blocks, %zu hipMalloc calls so far (%.1f MB), most in use at once %.1f MB\n",
                                      st.ms_total, job->pool.blocks.size(), job->pool.n_malloc, job->pool.malloc_bytes / 1e6, job->pool.peak_live / 1e6);
    return 0;
}

extern "C" int cora_hip_job_tables(const cora_hip_job *job, const uint8_t **exact, size_t *exact_bytes, uint64_t *exact_nodes,
                                   const uint8_t **inexact, size_t *inexact_bytes, uint64_t *inexact_nodes)
{
    if (!job) return fail(19, "null job");
    if (exact) *exact = job->exact_bytes.data();
    if (exact_bytes) *exact_bytes = job->exact_bytes.size();
    if (exact_nodes) *exact_nodes = job->stats.exact_nodes;
    if (inexact) *inexact = job->inexact_bytes.data();
    if (inexact_bytes) *inexact_bytes = job->inexact_bytes.size();
    if (inexact_nodes) *inexact_nodes = job->stats.inexact_nodes;
    return 0;
}

extern "C" int cora_hip_job_set_table_files(cora_hip_job *job, const char *exact_path, const char *inexact_path)
{
    if (!job || !exact_path || !inexact_path) return fail(19, "null argument");
    join_warm(job);
    join_exact_writer(job);
    job->exact_path = exact_path;
    if (!job->inexact_bytes.open_file(inexact_path)) return fail(68, "Can't write to file: %s", inexact_path);                 // HTS:1451-1456
    return 0;
}

extern "C" int cora_hip_job_finish_table_files(cora_hip_job *job)
{
    if (!job) return fail(19, "null job");
    join_warm(job);
    join_exact_writer(job);
    bool ok = true;
    if (job->inexact_bytes.fd >= 0) ok = job->inexact_bytes.close_file();
    if (!job->exact_path.empty()) {
        if (job->exact_written) ok = ok && !job->exact_write_failed;
        else {                                             // (a build over several ranks: the exact table came from rank 0's helper context)
            FILE *f = fopen(job->exact_path.c_str(), "wb");
            const size_t n = job->exact_bytes.size();
            const bool w = f && (n == 0 || fwrite(job->exact_bytes.data(), 1, n, f) == n);
            ok = ok && f && fclose(f) == 0 && w;
        }
    }
    const std::string ep = job->exact_path;
    job->exact_path.clear();
    job->exact_written = false;
    if (!ok) return fail(68, "Can't write to file: %s (or the <Inexact> file beside it)", ep.c_str());                           // HTS:2485-2489
    return 0;
}

extern "C" int cora_hip_job_stats(const cora_hip_job *job, cora_hip_stats *out)
{
    if (!job || !out) return fail(19, "null argument");
    *out = job->stats;
    return 0;
}

// ------------------------------------------------------------------ stage-level entry points
extern "C" int cora_hip_radix_sort(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val, uint64_t *tmp_hi, uint64_t *tmp_lo,
                                   uint32_t *tmp_val, uint64_t n, int bit_lo, int bit_hi)
{
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (!key_hi || !tmp_hi || (key_lo && !tmp_lo) || (val && !tmp_val)) return fail(19, "bad sort arguments");
    SortArrays keys, tmp;
    keys.a = key_hi; keys.b = key_lo; keys.c = val;
    tmp.a = tmp_hi; tmp.b = tmp_lo; tmp.c = tmp_val;
    SortScratch scratch;
    int rc = radix_sort(nullptr, keys, tmp, n, bit_lo, bit_hi, scratch, nullptr);
    if (!rc && keys.a != key_hi) {       // odd number of passes: bring the result home
        (void)hipMemcpyAsync(key_hi, keys.a, n * sizeof(u64), hipMemcpyDeviceToDevice, nullptr);
        if (key_lo) (void)hipMemcpyAsync(key_lo, keys.b, n * sizeof(u64), hipMemcpyDeviceToDevice, nullptr);
        if (val) (void)hipMemcpyAsync(val, keys.c, n * sizeof(u32), hipMemcpyDeviceToDevice, nullptr);
    }
    hipError_t e = hipStreamSynchronize(nullptr);
    scratch.release();
    if (rc || e != hipSuccess) return fail(100, "radix sort failed (%d, %s)", rc, hipGetErrorString(e));
    return 0;
}

// ---- the same sort for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc; not inside a stream
// capture: it runs on the null stream and synchronises).  No counterpart in the reference.  The per-tile digit histograms and the chunk sums/offsets, which
// cora_hip_radix_sort keeps in hipMalloc'ed SortScratch memory, live in a file-scope device array; every GPU holds its
// own copy at its own address (looked up once per device).  Every call on a GPU shares that GPU's array: calls that
// overlap on one GPU race.  Nothing in it needs to start at zero -- every pass writes each word before reading it.
constexpr u64 NOALLOC_MAX_RECORDS = 1ull << 24;
constexpr size_t NOALLOC_TILES = NOALLOC_MAX_RECORDS / RS_TILE_MIN;                   // 4096 tiles of the smallest shape
constexpr size_t NOALLOC_CHUNKS = (NOALLOC_TILES + RS_CHUNK_TILES - 1) / RS_CHUNK_TILES;
constexpr size_t NOALLOC_HIST_BYTES = NOALLOC_TILES * 256 * sizeof(u32);              // 4 MiB
constexpr size_t NOALLOC_CHUNK_BYTES = 2 * NOALLOC_CHUNKS * 256 * sizeof(u32);        // 128 KiB
static __device__ __align__(256) unsigned char scratch[NOALLOC_HIST_BYTES + NOALLOC_CHUNK_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static void *g_scratch_ptr[NOALLOC_MAX_DEVICES];       // device address of `scratch` on each GPU (the array's name is a host-side stand-in)
static std::mutex g_scratch_mutex;

// device address of this GPU's copy of `scratch`; `device` must be the current device
static int scratch_ptr_of(int device, void **out)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(102, "device %d out of range for the static sort scratch", device);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(100, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

extern "C" uint64_t cora_hip_radix_sort_noalloc_max(void) { return NOALLOC_MAX_RECORDS; }

extern "C" int cora_hip_radix_sort_noalloc(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val, uint64_t *tmp_hi, uint64_t *tmp_lo,
                                           uint32_t *tmp_val, uint64_t n, int bit_lo, int bit_hi)
{
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (!key_hi || !tmp_hi || (key_lo && !tmp_lo) || (val && !tmp_val)) return fail(19, "bad sort arguments");
    if (n > NOALLOC_MAX_RECORDS)
        return fail(19, "cora_hip_radix_sort_noalloc sorts at most %llu records (static scratch of %zu bytes); got %llu",
                    (unsigned long long)NOALLOC_MAX_RECORDS, sizeof scratch, (unsigned long long)n);
    int dev = -1;
    HIPQ(hipGetDevice(&dev));
    void *sp = nullptr;
    RCQ(scratch_ptr_of(dev, &sp));
    SortScratch sc;                                        // borrows the static array: reserve() never allocates, release() never frees
    sc.bind_external((u32 *)sp, NOALLOC_TILES, (u32 *)((unsigned char *)sp + NOALLOC_HIST_BYTES), NOALLOC_CHUNKS);
    SortArrays keys, tmp;
    keys.a = key_hi; keys.b = key_lo; keys.c = val;
    tmp.a = tmp_hi; tmp.b = tmp_lo; tmp.c = tmp_val;
    int rc = radix_sort(nullptr, keys, tmp, n, bit_lo, bit_hi, sc, nullptr);      // untimed: no events are created
    if (!rc && n && keys.a != key_hi) {       // odd number of passes: bring the result home
        (void)hipMemcpyAsync(key_hi, keys.a, n * sizeof(u64), hipMemcpyDeviceToDevice, nullptr);
        if (key_lo) (void)hipMemcpyAsync(key_lo, keys.b, n * sizeof(u64), hipMemcpyDeviceToDevice, nullptr);
        if (val) (void)hipMemcpyAsync(val, keys.c, n * sizeof(u32), hipMemcpyDeviceToDevice, nullptr);
    }
    hipError_t e = hipStreamSynchronize(nullptr);
    if (rc || e != hipSuccess) return fail(100, "radix sort failed (%d, %s)", rc, hipGetErrorString(e));
    return 0;
}

extern "C" int cora_hip_test_list_sort(uint32_t *elems, uint64_t n, int forced_depth)
{
    g_err.clear();
    if (g_device < 0) return fail(2, "cora_hip_init has not been called");
    if (!n) return 0;
    HIPQ(hipSetDevice(g_device));
    PoolScope no_pool(nullptr);
    DBuf<u32> d;
    RCQ(d.alloc(n));
    HIPQ(hipMemcpy(d.p, elems, n * sizeof(u32), hipMemcpyHostToDevice));
    launch_test_list_sort(nullptr, d.p, n, forced_depth);
    HIPQ(hipGetLastError());
    HIPQ(hipDeviceSynchronize());
    HIPQ(hipMemcpy(elems, d.p, n * sizeof(u32), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int cora_hip_test_sort_host(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val, uint64_t n, int bit_lo, int bit_hi)
{
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    DBuf<u64> a, b, ta, tb;
    DBuf<u32> c, tc;
    RCQ(a.alloc(n)); RCQ(ta.alloc(n));
    HIPQ(hipMemcpy(a.p, key_hi, n * sizeof(u64), hipMemcpyHostToDevice));
    if (key_lo) { RCQ(b.alloc(n)); RCQ(tb.alloc(n)); HIPQ(hipMemcpy(b.p, key_lo, n * sizeof(u64), hipMemcpyHostToDevice)); }
    if (val) { RCQ(c.alloc(n)); RCQ(tc.alloc(n)); HIPQ(hipMemcpy(c.p, val, n * sizeof(u32), hipMemcpyHostToDevice)); }
    RCQ(cora_hip_radix_sort(a.p, key_lo ? b.p : nullptr, val ? c.p : nullptr, ta.p, key_lo ? tb.p : nullptr, val ? tc.p : nullptr, n, bit_lo, bit_hi));
    HIPQ(hipMemcpy(key_hi, a.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    if (key_lo) HIPQ(hipMemcpy(key_lo, b.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    if (val) HIPQ(hipMemcpy(val, c.p, n * sizeof(u32), hipMemcpyDeviceToHost));
    return 0;
}

// The two-word sort with the second word climbing through a chain of `chain` arrays (SortArrays::b_chain_n, what the seed
// sorts of a large build use), on host buffers.
extern "C" int cora_hip_test_sort_chain_host(uint64_t *key_hi, uint64_t *key_lo, uint64_t n, int bit_lo, int bit_hi, int chain)
{
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (!key_hi || !key_lo || chain < 1 || chain > 8) return fail(19, "bad argument");
    if (!n) return 0;
    DBuf<u64> a, ta, b;
    RCQ(a.alloc(n)); RCQ(ta.alloc(n)); RCQ(b.alloc(n * (u64)(chain > 1 ? chain : 2)));
    HIPQ(hipMemcpy(a.p, key_hi, n * sizeof(u64), hipMemcpyHostToDevice));
    HIPQ(hipMemcpy(b.p, key_lo, n * sizeof(u64), hipMemcpyHostToDevice));
    SortArrays keys, tmp;
    keys.a = a.p; keys.b = b.p; tmp.a = ta.p; tmp.b = b.p + n;
    if (chain > 1) { keys.b_chain_n = chain; keys.b_stride = (size_t)n; }
    SortScratch scratch;
    const int rc = radix_sort(nullptr, keys, tmp, n, bit_lo, bit_hi, scratch, nullptr);
    if (!rc) (void)hipDeviceSynchronize();
    scratch.release();
    if (rc) return fail(100, "radix sort failed (%d)", rc);
    HIPQ(hipMemcpy(key_hi, keys.a, n * sizeof(u64), hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(key_lo, keys.b, n * sizeof(u64), hipMemcpyDeviceToHost));
    return 0;
}

// The sort whose first pass keeps one rank's records (radix_sort_count_kept + SortSource: the multi-GPU flow on few GPUs),
// on host buffers: all n records in, the kept ones out, sorted.
extern "C" int cora_hip_test_sort_kept_host(const uint64_t *key_hi, const uint64_t *key_lo, const uint32_t *val, uint64_t n, int bit_lo, int bit_hi,
                                            int owner_mode, uint32_t world, uint32_t me, uint64_t *out_hi, uint64_t *out_lo, uint32_t *out_val,
                                            uint64_t cap, uint64_t *n_kept)
{
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (!key_hi || !out_hi || !n_kept || (key_lo && !out_lo) || (val && !out_val)) return fail(19, "null argument");
    hipStream_t s = nullptr;
    DBuf<u64> a, b, ka, kb, ta, tb;
    DBuf<u32> c, kc, tc;
    RCQ(a.alloc(n));
    HIPQ(hipMemcpy(a.p, key_hi, n * sizeof(u64), hipMemcpyHostToDevice));
    if (key_lo) { RCQ(b.alloc(n)); HIPQ(hipMemcpy(b.p, key_lo, n * sizeof(u64), hipMemcpyHostToDevice)); }
    if (val) { RCQ(c.alloc(n)); HIPQ(hipMemcpy(c.p, val, n * sizeof(u32), hipMemcpyHostToDevice)); }
    SortSource from;
    from.src.a = a.p; from.src.b = key_lo ? b.p : nullptr; from.src.c = val ? c.p : nullptr;
    from.n = n; from.fmode = owner_mode; from.world = world; from.me = me;
    SortScratch scratch;
    u64 kept = 0;
    int rc = radix_sort_count_kept(s, from, bit_lo, bit_hi, scratch, &kept);
    if (rc) { scratch.release(); return fail(19, "radix_sort_count_kept refused the arguments (%d)", rc); }
    *n_kept = kept;
    if (kept > cap) { scratch.release(); return fail(19, "%llu records kept, room for %llu", (unsigned long long)kept, (unsigned long long)cap); }
    RCQ(ka.alloc(kept)); RCQ(ta.alloc(kept));
    if (key_lo) { RCQ(kb.alloc(kept)); RCQ(tb.alloc(kept)); }
    if (val) { RCQ(kc.alloc(kept)); RCQ(tc.alloc(kept)); }
    SortArrays keys, tmp;
    keys.a = ka.p; keys.b = key_lo ? kb.p : nullptr; keys.c = val ? kc.p : nullptr;
    tmp.a = ta.p; tmp.b = key_lo ? tb.p : nullptr; tmp.c = val ? tc.p : nullptr;
    rc = radix_sort(s, keys, tmp, kept, bit_lo, bit_hi, scratch, nullptr, nullptr, &from);
    if (!rc) rc = (int)hipStreamSynchronize(s);
    scratch.release();
    if (rc) return fail(100, "radix sort failed (%d)", rc);
    if (kept) {
        HIPQ(hipMemcpy(out_hi, keys.a, kept * sizeof(u64), hipMemcpyDeviceToHost));
        if (key_lo) HIPQ(hipMemcpy(out_lo, keys.b, kept * sizeof(u64), hipMemcpyDeviceToHost));
        if (val) HIPQ(hipMemcpy(out_val, keys.c, kept * sizeof(u32), hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int cora_hip_test_pack(cora_hip_job *job, uint64_t *packed, uint32_t *bad, uint64_t *contig_start, uint64_t *padded_len)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    const DevParams &dp = job->dp;
    DevCounters zero;
    memset(&zero, 0, sizeof zero);
    RCQ(write_counters(job, zero));
    launch_pack(job->stream, dp, job->raw.p, job->pk.p, job->bad.p, job->ctr.p);
    HIPQ(hipGetLastError());
    if (packed) HIPQ(hipMemcpyAsync(packed, job->pk.p, dp.n_words * sizeof(u64), hipMemcpyDeviceToHost, job->stream));
    if (bad) HIPQ(hipMemcpyAsync(bad, job->bad.p, dp.n_words * sizeof(u32), hipMemcpyDeviceToHost, job->stream));
    HIPQ(hipStreamSynchronize(job->stream));
    if (contig_start) memcpy(contig_start, job->hg.gstart.data(), dp.n_contigs * sizeof(u64));
    if (padded_len) *padded_len = dp.G;
    DevCounters h;
    RCQ(read_counters(job, h));
    if (h.format_errors) return fail(51, "FASTA image does not have the fixed line width (%llu misplaced line ends)", (unsigned long long)h.format_errors);
    return 0;
}

extern "C" int cora_hip_test_window_bits(cora_hip_job *job, uint32_t *fast, uint32_t *slow, uint64_t cap_words, uint64_t *n_words)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    PoolScope pool_scope(&job->pool);
    const DevParams &dp = job->dp;
    if (n_words) *n_words = dp.n_words;
    if (dp.n_words > cap_words) return fail(19, "buffer too small: %llu words", (unsigned long long)dp.n_words);
    RCQ(st_pack(job));
    DBuf<u32> f, sl;
    RCQ(f.alloc(dp.n_words)); RCQ(sl.alloc(dp.n_words));
    launch_test_window_bits(job->stream, dp, job->bad.p, f.p, sl.p);
    HIPQ(hipGetLastError());
    HIPQ(hipMemcpyAsync(fast, f.p, dp.n_words * sizeof(u32), hipMemcpyDeviceToHost, job->stream));
    HIPQ(hipMemcpyAsync(slow, sl.p, dp.n_words * sizeof(u32), hipMemcpyDeviceToHost, job->stream));
    HIPQ(hipStreamSynchronize(job->stream));
    return 0;
}

extern "C" int cora_hip_test_kmers(cora_hip_job *job, uint64_t *key_hi, uint64_t *key_lo, uint32_t *gpos, uint64_t cap, uint64_t *n_out)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    PoolScope pool_scope(&job->pool);
    DBuf<u64> hi, lo;
    DBuf<u32> gp;
    u64 n = 0;
    RCQ(st_pack(job));
    {
        const DevParams &dp = job->dp;
        const u32 n_blocks = (u32)((dp.G + 255) / 256);
        DBuf<u32> bc;
        DBuf<u64> boff, scr, tot;
        RCQ(bc.alloc(n_blocks)); RCQ(boff.alloc(n_blocks)); RCQ(scr.alloc(scan_scratch_elems(n_blocks))); RCQ(tot.alloc(1));
        launch_kmer_count(job->stream, dp, job->bad.p, bc.p, n_blocks);
        if (exclusive_scan_u32_to_u64(job->stream, bc.p, boff.p, n_blocks, scr.p, tot.p)) return fail(100, "scan launch failed");
        HIPQ(hipMemcpyAsync(&n, tot.p, sizeof n, hipMemcpyDeviceToHost, job->stream));
        HIPQ(hipStreamSynchronize(job->stream));
        RCQ(hi.alloc(n)); RCQ(lo.alloc(n)); RCQ(gp.alloc(n));
        launch_kmer_emit(job->stream, dp, job->pk.p, job->bad.p, boff.p, n_blocks, hi.p, lo.p, gp.p);
        HIPQ(hipStreamSynchronize(job->stream));
    }
    if (n_out) *n_out = n;
    if (n > cap) return fail(19, "buffer too small: %llu windows", (unsigned long long)n);
    HIPQ(hipMemcpy(key_hi, hi.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(key_lo, lo.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(gpos, gp.p, n * sizeof(u32), hipMemcpyDeviceToHost));
    return 0;
}
