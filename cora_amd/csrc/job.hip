// job.hip -- C-ABI of libcora_hip.so (include/cora_hip.h) and the stage orchestration of one
// table build on one MI355X.  HTS = /root/reference/homTable_setup.cpp.
#include "job_internal.hpp"

using namespace cora;

extern "C" const char *cora_hip_last_error(void) { return g_err.c_str(); }

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
    memset(&j->stats, 0, sizeof j->stats);
    HostGenome &hg = j->hg;
    hg.K = params->K; hg.d = params->d; hg.chr_short = params->chr_short;
    hg.n_groups = params->n_groups; hg.n_threads = params->n_threads;
    memcpy(hg.sig2, params->sig2, 16);
    memcpy(hg.sig4, params->sig4, 256);
    make_seed_template(hg.K, hg.d, hg.tmpl, &hg.seed_q);
    if (const char *e = getenv("CORA_HASH_BITS")) { int v = atoi(e); if (v >= 8 && v <= 56) j->hash_bits = v & ~7; }
    if (const char *e = getenv("CORA_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) j->host_threads = v; }
    if (hipStreamCreate(&j->stream) != hipSuccess) { delete j; return fail(100, "hipStreamCreate failed"); }
    for (auto &e : j->ev) if (hipEventCreate(&e) != hipSuccess) { delete j; return fail(100, "hipEventCreate failed"); }
    *out = j;
    return 0;
}

extern "C" void cora_hip_job_destroy(cora_hip_job *job)
{
    if (!job) return;
    job->sort_scratch.release();
    job->raw.release(); job->d_gstart.release(); job->d_off.release(); job->d_len.release(); job->d_window.release();
    job->pk.release(); job->bad.release(); job->ctr.release();
    job->pool.destroy();
    for (auto &e : job->ev) if (e) (void)hipEventDestroy(e);
    if (job->stream) (void)hipStreamDestroy(job->stream);
    delete job;
}

extern "C" int cora_hip_job_set_reference(cora_hip_job *job, const uint8_t *image, size_t image_bytes,
                                          const cora_hip_contig *contigs, int32_t n_contigs)
{
    if (!job || !image || !contigs || n_contigs <= 0) return fail(51, "bad reference arguments");
    if (!job->prm.chr_short && n_contigs > 127)
        return fail(19, "%d contigs need the 2-byte layout (homTable_setup_L; cora.cpp:25,488-493)", n_contigs);
    if (n_contigs > 65530) return fail(19, "too many contigs (%d)", n_contigs);                                            // HTS:23-27
    HostGenome &hg = job->hg;
    hg.gstart.assign(n_contigs + 1, 0);
    hg.len.assign(n_contigs, 0);
    std::vector<u64> off(n_contigs);
    u64 g = 0, bases = 0;
    const u32 W = (u32)job->prm.line_width;
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

    RCQ(job->raw.alloc(image_bytes + 16));
    HIPQ(hipMemcpyAsync(job->raw.p, image, image_bytes, hipMemcpyHostToDevice, job->stream));
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
    RCQ(job->ctr.alloc(1));
    HIPQ(hipStreamSynchronize(job->stream));
    job->have_ref = true;
    return 0;
}

// pack + valid windows in position order; hashed: 12-byte records (hash|flags, gpos), else full 128-bit keys
static int stage_kmers(cora_hip_job *job, bool hashed, DBuf<u64> &a, DBuf<u64> &b, DBuf<u32> &gp, u64 &n_kmers)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    DevCounters zero;
    memset(&zero, 0, sizeof zero);
    zero.first_singleton_key = ~0ull;
    RCQ(write_counters(job, zero));
    launch_pack(s, dp, job->raw.p, job->pk.p, job->bad.p, job->ctr.p);
    const u32 n_blocks = (u32)((dp.G + 255) / 256);
    DBuf<u32> bc;
    DBuf<u64> boff, scr, tot;
    RCQ(bc.alloc(n_blocks)); RCQ(boff.alloc(n_blocks)); RCQ(scr.alloc(scan_scratch_elems(n_blocks))); RCQ(tot.alloc(1));
    launch_kmer_count(s, dp, job->bad.p, bc.p, n_blocks);
    if (exclusive_scan_u32_to_u64(s, bc.p, boff.p, n_blocks, scr.p, tot.p)) return fail(100, "scan launch failed");
    u64 n = 0;
    HIPQ(hipMemcpyAsync(&n, tot.p, sizeof n, hipMemcpyDeviceToHost, s));
    DevCounters h;
    RCQ(read_counters(job, h));
    if (h.format_errors)
        return fail(51, "FASTA image does not have the fixed line width the reference's loader assumes (%llu misplaced line ends)",
                    (unsigned long long)h.format_errors);
    n_kmers = n;
    RCQ(a.alloc(n)); RCQ(gp.alloc(n));
    if (hashed) launch_kmer_emit_hashed(s, dp, job->pk.p, job->bad.p, boff.p, n_blocks, a.p, gp.p);
    else { RCQ(b.alloc(n)); launch_kmer_emit(s, dp, job->pk.p, job->bad.p, boff.p, n_blocks, a.p, b.p, gp.p); }
    HIPQ(hipStreamSynchronize(s));
    return 0;
}

extern "C" int cora_hip_job_run(cora_hip_job *job)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    PoolScope pool_scope(&job->pool);
    const DevParams &dp = job->dp;
    const HostGenome &hg = job->hg;
    hipStream_t s = job->stream;
    cora_hip_stats &st = job->stats;
    memset(&st, 0, sizeof st);
    st.genome_bases = job->genome_bases;
    job->sort_timing = SortTiming();
    job->sort_timing.enabled = true;
    const auto t_begin = std::chrono::steady_clock::now();
    double host_ms = 0;
    HIPQ(hipEventRecord(job->ev[0], s));

    // ---- stage 1: pack, hashed K-mer records (HTS:2499-2565) ----
    DBuf<u64> rec, klo;
    DBuf<u32> gp;
    u64 N = 0;
    RCQ(stage_kmers(job, true, rec, klo, gp, N));
    st.kmers = N;
    HIPQ(hipEventRecord(job->ev[1], s));
    // every contig must own at least one window (else the reference reads before its buffer, HTS:2530)
    {
        launch_contig_check(s, dp, gp.p, N, job->ctr.p);
        DevCounters h;
        RCQ(read_counters(job, h));
        if (h.empty_contigs)
            return fail(54, "%llu contig(s) hold no findable %d-mer: the reference's behaviour is undefined there (HTS:2530)",
                        (unsigned long long)h.empty_contigs, dp.K);
    }

    // ---- stage 2: sort, classes, representatives (HTS:2516-2578, 249-497) ----
    DBuf<u32> reps;
    u64 NR = 0;
    FinishStats fst;
    if (N >= (1ull << 32) - 1) return fail(19, "too many K-mers for 32-bit indices");
    {
        DBuf<u64> trec, thi, tlo;
        DBuf<u32> tgp, head_flag, mixed_list;
        const u32 mixed_cap = 4096;
        RCQ(trec.alloc(N)); RCQ(tgp.alloc(N)); RCQ(head_flag.alloc(N)); RCQ(mixed_list.alloc(mixed_cap));
        // Equal canonical keys have equal hashes: sort 12-byte (hash, position) records on the top hb hash bits
        // (6-7 passes) instead of 20-byte full keys on 2K bits (13 passes); collisions are found and repaired below.
        int hb = job->hash_bits ? job->hash_bits : (N <= (1ull << 27) ? 48 : 56);
        SortArrays keys, tmp;
        keys.a = rec.p; keys.c = gp.p; tmp.a = trec.p; tmp.c = tgp.p;
        if (radix_sort(s, keys, tmp, N, 64 - hb, 64, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        launch_same_detect(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_list.p, mixed_cap, job->ctr.p);
        DevCounters h;
        RCQ(read_counters(job, h));
        bool fallback = false;
        st.hash_mixed_runs = h.n_mixed;
        if (h.n_mixed > mixed_cap) fallback = true;
        else if (h.n_mixed) {
            launch_mixed_fix(s, dp, job->pk.p, keys.a, keys.c, N, hb, mixed_list.p, (u32)h.n_mixed, job->ctr.p);
            DevCounters z = h;
            z.n_mixed = 0;
            RCQ(read_counters(job, h));
            z.fallback = h.fallback;
            RCQ(write_counters(job, z));
            launch_same_detect(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_list.p, mixed_cap, job->ctr.p);
            RCQ(read_counters(job, h));
            if (h.n_mixed || h.fallback) fallback = true;
        }
        const u64 *rec_sorted = keys.a;
        const u32 *gp_sorted = keys.c;
        DBuf<u64> fhi, flo;
        DBuf<u32> fgp;
        if (fallback) {                 // a long run of colliding keys: redo with the full 128-bit keys (always exact)
            st.full_key_fallback = 1;
            u64 N2 = 0;
            RCQ(stage_kmers(job, false, fhi, flo, fgp, N2));
            if (N2 != N) return fail(100, "internal: window count changed");
            RCQ(thi.alloc(N)); RCQ(tlo.alloc(N));
            SortArrays fk, ft;
            fk.a = fhi.p; fk.b = flo.p; fk.c = fgp.p; ft.a = thi.p; ft.b = tlo.p; ft.c = tgp.p;
            if (radix_sort(s, fk, ft, N, 128 - 2 * dp.K, 128, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
            launch_heads_from_keys(s, dp, job->pk.p, fk.a, fk.b, fk.c, N, head_flag.p, rec.p);
            if (fk.c != fgp.p) HIPQ(hipMemcpyAsync(fgp.p, fk.c, N * sizeof(u32), hipMemcpyDeviceToDevice, s));
            rec_sorted = rec.p; gp_sorted = fgp.p;
        }
        // class index of every record, class starts
        DBuf<u64> cid, scr0, tot0;
        RCQ(cid.alloc(N)); RCQ(scr0.alloc(scan_scratch_elems(N))); RCQ(tot0.alloc(1));
        if (exclusive_scan_u32_to_u64(s, head_flag.p, cid.p, N, scr0.p, tot0.p)) return fail(100, "scan launch failed");
        HIPQ(hipMemcpyAsync(&NR, tot0.p, sizeof NR, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        DBuf<u32> class_start;
        RCQ(class_start.alloc(NR + 1));
        launch_node_starts(s, head_flag.p, cid.p, N, class_start.p);
        const u32 n32 = (u32)N;
        HIPQ(hipMemcpyAsync(class_start.p + NR, &n32, sizeof(u32), hipMemcpyHostToDevice, s));
        cid.release(); scr0.release();

        const u32 kb = (u32)((NR + 255) / 256);
        DBuf<u32> blk_nons;
        DBuf<u64> blk_min, nons_off, scr2, tots;
        RCQ(blk_nons.alloc(kb)); RCQ(blk_min.alloc(kb)); RCQ(nons_off.alloc(kb));
        RCQ(scr2.alloc(scan_scratch_elems(kb))); RCQ(tots.alloc(1));
        launch_class_count(s, rec_sorted, gp_sorted, class_start.p, (u32)NR, blk_nons.p, (unsigned long long *)blk_min.p, job->ctr.p);
        if (exclusive_scan_u32_to_u64(s, blk_nons.p, nons_off.p, kb, scr2.p, tots.p)) return fail(100, "scan launch failed");
        u64 NNS = 0;
        HIPQ(hipMemcpyAsync(&NNS, tots.p, sizeof NNS, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        st.representatives = NR; st.exact_classes = NNS;
        DBuf<u32> repmask;
        DBuf<u64> cls, cls_tmp;
        RCQ(repmask.alloc(dp.n_words)); RCQ(cls.alloc(NNS)); RCQ(cls_tmp.alloc(NNS));
        HIPQ(hipMemsetAsync(repmask.p, 0, dp.n_words * sizeof(u32), s));
        launch_class_emit(s, rec_sorted, gp_sorted, class_start.p, (u32)NR, repmask.p, cls.p, nons_off.p);
        // representatives in position order
        {
            const u32 nb = (u32)((dp.n_words + 255) / 256);
            DBuf<u32> bc; DBuf<u64> boff, scr;
            RCQ(bc.alloc(nb)); RCQ(boff.alloc(nb)); RCQ(scr.alloc(scan_scratch_elems(nb)));
            launch_bitmap_count(s, repmask.p, dp.n_words, bc.p, nb);
            if (exclusive_scan_u32_to_u64(s, bc.p, boff.p, nb, scr.p, nullptr)) return fail(100, "scan launch failed");
            RCQ(reps.alloc(NR));
            launch_bitmap_emit(s, repmask.p, dp.n_words, boff.p, nb, reps.p);
            HIPQ(hipStreamSynchronize(s));
        }
        // classes by representative, entries, block merge flags
        ExactDeviceResult er;
        if (NNS) {
            SortArrays ck, ct;
            ck.a = cls.p; ct.a = cls_tmp.p;
            if (radix_sort(s, ck, ct, NNS, 32, 32 + bits_for(dp.G), job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
            DBuf<u32> cf, ce;
            DBuf<u64> eoff, scr, tot;
            RCQ(cf.alloc(NNS)); RCQ(ce.alloc(NNS)); RCQ(eoff.alloc(NNS)); RCQ(scr.alloc(scan_scratch_elems(NNS))); RCQ(tot.alloc(1));
            launch_class_info(s, rec_sorted, class_start.p, ck.a, (u32)NNS, job->ctr.p, cf.p, ce.p);
            if (exclusive_scan_u32_to_u64(s, ce.p, eoff.p, NNS, scr.p, tot.p)) return fail(100, "scan launch failed");
            u64 NE = 0;
            HIPQ(hipMemcpyAsync(&NE, tot.p, sizeof NE, hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            DBuf<u32> ent; DBuf<u8> edir, mp;
            RCQ(ent.alloc(NE)); RCQ(edir.alloc(NE)); RCQ(mp.alloc(NNS));
            launch_class_entries(s, rec_sorted, gp_sorted, class_start.p, ck.a, (u32)NNS, cf.p, eoff.p, ent.p, edir.p);
            launch_class_merge(s, ck.a, (u32)NNS, ce.p, eoff.p, ent.p, edir.p, mp.p);
            RCQ(d2h(job, job->h_cls, er.cls, (const u64 *)ck.a, NNS));
            RCQ(d2h(job, job->h_cls_entries, er.cls_entries, (const u32 *)ce.p, NNS));
            RCQ(d2h(job, job->h_ent_off, er.ent_off, (const u64 *)eoff.p, NNS));
            RCQ(d2h(job, job->h_ent, er.ent, (const u32 *)ent.p, NE));
            RCQ(d2h(job, job->h_ent_dir, er.ent_dir, (const u8 *)edir.p, NE));
            RCQ(d2h(job, job->h_merge_prev, er.merge_prev, (const u8 *)mp.p, NNS));
            HIPQ(hipStreamSynchronize(s));
        }
        HIPQ(hipGetLastError());
        HIPQ(hipEventRecord(job->ev[2], s));
        HIPQ(hipStreamSynchronize(s));
        const auto h0 = std::chrono::steady_clock::now();
        serialize_exact(hg, er, job->exact_bytes, fst);
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    }
    rec.release(); klo.release(); gp.release();

    // ---- stage 3: per seed: codes, sort, equal runs, verification (HTS:2671-2800, 2138-2265) ----
    DBuf<PairRec> pairs;
    u64 pair_cap = NR / 8 + (1u << 20);
    u64 n_pairs = 0;
    {
        RCQ(pairs.alloc(pair_cap));
        DBuf<u64> codes, codes_tmp, blk_off, scr, tots;
        DBuf<u32> blk_active;
        const u32 vb = (u32)((NR + 255) / 256);
        RCQ(codes.alloc(NR)); RCQ(codes_tmp.alloc(NR));
        RCQ(blk_active.alloc(vb)); RCQ(blk_off.alloc(vb));
        RCQ(scr.alloc(scan_scratch_elems(NR / 4 + vb + 4096))); RCQ(tots.alloc(2));
        for (int seed = 1; seed <= dp.nseeds && NR; seed++) {
            SortArrays ck, ct;
            ck.a = codes.p; ct.a = codes_tmp.p;
            launch_seed_codes(s, dp, job->pk.p, reps.p, NR, seed, ck.a);
            if (radix_sort(s, ck, ct, NR, 32, 64, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
            // records with an earlier run member (or a palindromic code), compacted
            launch_active_count(s, ck.a, NR, blk_active.p);
            if (exclusive_scan_u32_to_u64(s, blk_active.p, blk_off.p, vb, scr.p, tots.p)) return fail(100, "scan launch failed");
            u64 n_active = 0;
            HIPQ(hipMemcpyAsync(&n_active, tots.p, sizeof n_active, hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            if (n_active == 0) continue;
            if (n_active >= (1ull << 32)) return fail(19, "too many run members");
            DBuf<u32> active, cnt_recs, cnt_cand;
            DBuf<u64> rec_off, cand_off, scr_a;
            RCQ(active.alloc(n_active)); RCQ(cnt_recs.alloc(n_active)); RCQ(cnt_cand.alloc(n_active));
            RCQ(rec_off.alloc(n_active)); RCQ(cand_off.alloc(n_active)); RCQ(scr_a.alloc(scan_scratch_elems(n_active)));
            launch_active_emit(s, ck.a, NR, blk_off.p, active.p);
            launch_verify_count(s, dp, job->pk.p, ck.a, active.p, (u32)n_active, seed, cnt_recs.p, cnt_cand.p);
            if (exclusive_scan_u32_to_u64(s, cnt_recs.p, rec_off.p, n_active, scr_a.p, tots.p)) return fail(100, "scan launch failed");
            if (exclusive_scan_u32_to_u64(s, cnt_cand.p, cand_off.p, n_active, scr_a.p, tots.p + 1)) return fail(100, "scan launch failed");
            u64 totals[2] = {0, 0};
            HIPQ(hipMemcpyAsync(totals, tots.p, sizeof totals, hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            st.candidate_pairs += totals[1];
            if (n_pairs + totals[0] > pair_cap) {                 // grow, keeping the earlier seeds' records
                const u64 want = (n_pairs + totals[0]) * 2;
                DBuf<PairRec> bigger;
                RCQ(bigger.alloc(want));
                if (n_pairs) HIPQ(hipMemcpyAsync(bigger.p, pairs.p, n_pairs * sizeof(PairRec), hipMemcpyDeviceToDevice, s));
                HIPQ(hipStreamSynchronize(s));
                std::swap(pairs.p, bigger.p); std::swap(pairs.n, bigger.n);
                pair_cap = want;
            }
            if (totals[0]) launch_verify_write(s, dp, job->pk.p, ck.a, active.p, (u32)n_active, seed, rec_off.p, pairs.p + n_pairs);
            n_pairs += totals[0];
        }
        st.hom_pairs = n_pairs;
    }
    reps.release();
    HIPQ(hipGetLastError());
    HIPQ(hipEventRecord(job->ev[3], s));

    // ---- stage 4: directed items, (source,target) sort, lists, diagonal block merge (HTS:1022-1270) ----
    InexactDeviceResult ir;
    View<u64> packed_host;
    static const u32 zero_node = 0;
    const u64 M = 2 * n_pairs;
    if (M >= (1ull << 32)) return fail(19, "too many directed items (%llu)", (unsigned long long)M);
    st.inexact_items = M;
    if (M) {
        DBuf<u64> ik, ik_tmp, nexcl, scr, tot;
        DBuf<u32> im, im_tmp, hf, nstart;
        DBuf<int16_t> blen;
        RCQ(ik.alloc(M)); RCQ(ik_tmp.alloc(M)); RCQ(im.alloc(M)); RCQ(im_tmp.alloc(M));
        launch_items_expand(s, dp, pairs.p, n_pairs, ik.p, im.p);
        SortArrays kk, kt;
        kk.a = ik.p; kk.c = im.p; kt.a = ik_tmp.p; kt.c = im_tmp.p;
        const int gb = bits_for(dp.G);
        if (radix_sort(s, kk, kt, M, 0, gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        if (radix_sort(s, kk, kt, M, 32, 32 + gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        pairs.release();
        RCQ(hf.alloc(M)); RCQ(nexcl.alloc(M)); RCQ(scr.alloc(scan_scratch_elems(M))); RCQ(tot.alloc(1)); RCQ(blen.alloc(M));
        launch_node_flags(s, kk.a, M, hf.p, job->ctr.p);
        if (exclusive_scan_u32_to_u64(s, hf.p, nexcl.p, M, scr.p, tot.p)) return fail(100, "scan launch failed");
        u64 n_nodes = 0;
        HIPQ(hipMemcpyAsync(&n_nodes, tot.p, sizeof n_nodes, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        RCQ(nstart.alloc(n_nodes + 1));
        launch_node_starts(s, hf.p, nexcl.p, M, nstart.p);
        const u32 m32 = (u32)M;
        HIPQ(hipMemcpyAsync(nstart.p + n_nodes, &m32, sizeof(u32), hipMemcpyHostToDevice, s));
        launch_chain(s, dp, kk.a, kk.c, hf.p, nexcl.p, nstart.p, (u32)n_nodes, M, blen.p);
        HIPQ(hipGetLastError());
        RCQ(d2h(job, job->h_key, ir.key, (const u64 *)kk.a, M));
        RCQ(d2h(job, job->h_meta, ir.meta, (const u32 *)kk.c, M));
        RCQ(d2h(job, job->h_blen, ir.blen, (const int16_t *)blen.p, M));
        RCQ(d2h(job, job->h_node_start, ir.node_start, (const u32 *)nstart.p, n_nodes + 1));
        DevCounters h;
        RCQ(read_counters(job, h));
        if (h.n_dirty) RCQ(d2h(job, job->h_packed, packed_host, (const u64 *)job->pk.p, dp.n_words + 4));
        HIPQ(hipStreamSynchronize(s));
    } else {
        ir.node_start.p = &zero_node; ir.node_start.n = 1;
    }
    HIPQ(hipEventRecord(job->ev[4], s));
    HIPQ(hipStreamSynchronize(s));

    // ---- stage 5: final ordering and serialisation on the host (HTS:1353-1432) ----
    {
        const auto h0 = std::chrono::steady_clock::now();
        serialize_inexact(hg, ir, packed_host, job->inexact_bytes, fst, job->host_threads);
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    }
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

extern "C" int cora_hip_test_kmers(cora_hip_job *job, uint64_t *key_hi, uint64_t *key_lo, uint32_t *gpos, uint64_t cap, uint64_t *n_out)
{
    if (!job || !job->have_ref) return fail(51, "no reference set");
    PoolScope pool_scope(&job->pool);
    DBuf<u64> hi, lo;
    DBuf<u32> gp;
    u64 n = 0;
    RCQ(stage_kmers(job, false, hi, lo, gp, n));
    if (n_out) *n_out = n;
    if (n > cap) return fail(19, "buffer too small: %llu windows", (unsigned long long)n);
    HIPQ(hipMemcpy(key_hi, hi.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(key_lo, lo.p, n * sizeof(u64), hipMemcpyDeviceToHost));
    HIPQ(hipMemcpy(gpos, gp.p, n * sizeof(u32), hipMemcpyDeviceToHost));
    return 0;
}
