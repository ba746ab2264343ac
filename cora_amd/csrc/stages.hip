// stages.hip -- the stages of a table build as functions over device arrays, shared by the single-GPU
// run (job.hip) and the multi-GPU flow (shard.hip).  HTS = /root/reference/homTable_setup.cpp.
#include "job_internal.hpp"

using namespace cora;

// pack the whole reference (every rank of a sharded run keeps a full replica: 0.375 B per base)
int st_pack(cora_hip_job *job)
{
    DevCounters zero;
    memset(&zero, 0, sizeof zero);
    zero.first_singleton_key = ~0ull;
    RCQ(write_counters(job, zero));
    launch_pack(job->stream, job->dp, job->raw.p, job->pk.p, job->bad.p, job->ctr.p);
    return 0;                       // format_errors is looked at with the caller's next read of the counters (FORMAT_ERROR_CHECK)
}

// hashed records (hash|flags, gpos) of the valid windows of blocks [block_lo, block_lo + n_blocks) of 256 positions
// first_digit_ready (optional): the records' first sort digit (exact_sort_bits) has been left in job->sort_scratch.next_digit for the
// sort that follows at once (st_exact_classes without `from`): its first histogram then reads bytes (radix_sort.hip).
int st_kmers(cora_hip_job *job, u32 block_lo, u32 n_blocks, DBuf<u64> &rec, DBuf<u32> &gp, u64 &n_out, bool *first_digit_ready)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    DBuf<u32> bc;
    DBuf<u64> boff, scr, tot;
    RCQ(bc.alloc(n_blocks)); RCQ(boff.alloc(n_blocks)); RCQ(scr.alloc(scan_scratch_elems(n_blocks))); RCQ(tot.alloc(1));
    launch_kmer_count_range(s, dp, job->bad.p, bc.p, block_lo, n_blocks, job->valid.p);
    if (exclusive_scan_u32_to_u64(s, bc.p, boff.p, n_blocks, scr.p, tot.p)) return fail(100, "scan launch failed");
    u64 n = 0;
    HIPQ(hipMemcpyAsync(&n, tot.p, sizeof n, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    n_out = n;
    RCQ(rec.alloc(n)); RCQ(gp.alloc(n));
    unsigned char *fd = nullptr;
    const int fd_shift = (64 - exact_sort_bits(job)) & ~7;
    if (first_digit_ready) {
        *first_digit_ready = false;
        if (radix_sort_next_digit_wanted(job->sort_scratch, n) && job->sort_scratch.reserve_next_digit(n) == 0) { fd = job->sort_scratch.next_digit; *first_digit_ready = true; }
    }
    launch_kmer_emit_hashed_range(s, dp, job->pk.p, job->valid.p, boff.p, block_lo, n_blocks, rec.p, gp.p, fd, fd_shift);
    HIPQ(hipStreamSynchronize(s));
    return 0;
}

// Sort the records, group equal canonical K-mers exactly, mark the non-representatives, list the classes with >= 2
// entries and find the first singleton (HTS:2516-2578).  x.rec / x.gp hold x.N records in which equal K-mers
// appear in increasing position.
// from (multi-GPU flow on few GPUs): the sort's first pass reads ALL records of the reference from there and keeps the
// x.N this rank owns (radix_sort_count_kept has run on the same bit range, exact_sort_bits); x.rec / x.gp are then only sized.
int st_exact_classes(cora_hip_job *job, ExactLocal &x, const SortSource *from)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    cora_hip_stats &st = job->stats;
    const u64 N = x.N;
    if (N >= (1ull << 32) - 1) return fail(19, "too many K-mers for 32-bit indices");
    DBuf<u64> head_flag;                        // one bit per record: opens a class
    DBuf<u32> mixed_list;
    const u32 mixed_cap = (u32)std::max<u64>(65536, N / 128), long_cap = 65536;    // 40 hash bits at 3e9 records: 2e6 listed places
    RCQ(x.trec.alloc(N)); RCQ(x.tgp.alloc(N)); RCQ(head_flag.alloc(((N + 255) / 256) * 4)); RCQ(mixed_list.alloc(mixed_cap));
    // Equal canonical keys have equal hashes: sort 12-byte (hash, position) records on the top hb hash bits
    // (6-7 passes) instead of 20-byte full keys on 2K bits (13 passes); collisions are found and repaired below.
    // About N^2 / 2^(hb+1) pairs of different K-mers collide: a thousand at 64 M records, 2e6 at 3e9 -- repaired in 3 ms,
    // where a sixth pass over 3e9 records costs 20 (48 bits: CORA_HASH_BITS=48).
    const int hb = exact_sort_bits(job);
    NextDigitLease digit_bytes(job->sort_scratch, N);         // (does nothing inside the single-GPU run's own lease)
    SortArrays keys, tmp;
    keys.a = x.rec.p; keys.c = x.gp.p; tmp.a = x.trec.p; tmp.c = x.tgp.p;
    if (radix_sort(s, keys, tmp, N, 64 - hb, 64, job->sort_scratch, &job->sort_timing, nullptr, from, !from && x.first_digit_ready)) return fail(100, "radix sort failed");
    x.first_digit_ready = false;
    DevCounters h;
    ZERO_COUNTERS(job, n_mixed, 2);                              // n_mixed, fallback
    {
        // heads and the places where an equal-hash run holds its keys out of order; the latter as a list, in record order
        const u64 n_words = (N + 63) / 64, nb = (n_words + 255) / 256;
        DBuf<u64> mixed_mask, boff, scr0, tot0;
        DBuf<u32> bcnt;
        RCQ(mixed_mask.alloc(((N + 255) / 256) * 4)); RCQ(bcnt.alloc(nb)); RCQ(boff.alloc(nb)); RCQ(scr0.alloc(scan_scratch_elems(nb))); RCQ(tot0.alloc(1));
        launch_same_detect(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_mask.p);
        launch_mask_count(s, mixed_mask.p, n_words, bcnt.p);
        if (exclusive_scan_u32_to_u64(s, bcnt.p, boff.p, nb, scr0.p, tot0.p)) return fail(100, "scan launch failed");
        u64 n_mixed = 0;
        HIPQ(hipMemcpyAsync(&n_mixed, tot0.p, sizeof n_mixed, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        h.n_mixed = n_mixed;
        if (n_mixed && n_mixed <= mixed_cap) launch_mask_emit(s, mixed_mask.p, n_words, boff.p, mixed_list.p);
    }
    bool fallback = false;
    st.hash_mixed_runs += h.n_mixed;
    if (h.n_mixed > mixed_cap) fallback = true;
    else if (h.n_mixed) {
        // the repair re-orders each such run by key and rewrites its class heads; short runs in registers, long ones
        // (a many-copy K-mer that collided) by a workgroup each, using the sort's spare buffers as the copy
        DBuf<u32> claim;
        DBuf<u64> long_runs;
        RCQ(claim.alloc(N / 32 + 1)); RCQ(long_runs.alloc(2 * (size_t)long_cap));
        HIPQ(hipMemsetAsync(claim.p, 0, (N / 32 + 1) * sizeof(u32), s));
        ZERO_COUNTERS(job, n_long, 1);
        launch_mixed_fix(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_list.p, (u32)h.n_mixed, claim.p, long_runs.p, long_cap, job->ctr.p);
        RCQ(read_counters(job, h));
        if (h.fallback) fallback = true;
        else if (h.n_long) {
            launch_mixed_fix_long(s, dp, job->pk.p, keys.a, keys.c, tmp.a, tmp.c, long_runs.p, (u32)h.n_long, head_flag.p, job->ctr.p);
            RCQ(read_counters(job, h));
            if (h.fallback) fallback = true;
        }
    }
    x.rec_s = keys.a; x.gp_s = keys.c;
    if (fallback) {                 // a long run of colliding keys: redo on the full 128-bit keys (always exact).
        st.full_key_fallback = 1;   // Equal K-mers are already adjacent and in position order, and the sort is stable.
        DBuf<u64> fhi, flo, thi, tlo;
        RCQ(fhi.alloc(N)); RCQ(flo.alloc(N)); RCQ(thi.alloc(N)); RCQ(tlo.alloc(N));
        launch_keys_from_gpos(s, dp, job->pk.p, keys.c, N, fhi.p, flo.p);
        SortArrays fk, ft;
        fk.a = fhi.p; fk.b = flo.p; fk.c = (u32 *)keys.c; ft.a = thi.p; ft.b = tlo.p; ft.c = (u32 *)tmp.c;
        if (radix_sort(s, fk, ft, N, 128 - 2 * dp.K, 128, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        u64 *rec_out = (u64 *)tmp.a;                         // the hashed records are no longer needed: reuse as flag words
        launch_heads_from_keys(s, dp, job->pk.p, fk.a, fk.b, fk.c, N, head_flag.p, rec_out);
        HIPQ(hipStreamSynchronize(s));
        x.rec_s = rec_out; x.gp_s = fk.c;
    }
    // number of classes = set bits of the head mask
    {
        const u64 n_words = (N + 63) / 64, nb = (n_words + 255) / 256;
        DBuf<u32> bcnt;
        DBuf<u64> boff, scr0, tot0;
        RCQ(bcnt.alloc(nb)); RCQ(boff.alloc(nb)); RCQ(scr0.alloc(scan_scratch_elems(nb))); RCQ(tot0.alloc(1));
        launch_mask_count(s, head_flag.p, n_words, bcnt.p);
        if (exclusive_scan_u32_to_u64(s, bcnt.p, boff.p, nb, scr0.p, tot0.p)) return fail(100, "scan launch failed");
        HIPQ(hipMemcpyAsync(&x.NR, tot0.p, sizeof x.NR, hipMemcpyDeviceToHost, s));
    }
    // one pass over the sorted records: classes with >= 2 entries, the first singleton, and the later members of classes
    // (= the positions that are NOT representatives)
    RCQ(x.nonrep.alloc(dp.n_words));
    HIPQ(hipMemsetAsync(x.nonrep.p, 0, dp.n_words * sizeof(u32), s));
    const u32 kb = (u32)((N + 255) / 256);
    DBuf<u32> blk_nons;
    DBuf<u64> blk_min, nons_off, scr2, tots, nons_mask;
    RCQ(blk_nons.alloc(kb)); RCQ(blk_min.alloc(kb)); RCQ(nons_off.alloc(kb)); RCQ(nons_mask.alloc((u64)kb * 4));
    RCQ(scr2.alloc(scan_scratch_elems(kb))); RCQ(tots.alloc(1));
    launch_class_scan(s, x.rec_s, x.gp_s, head_flag.p, N, blk_nons.p, (unsigned long long *)blk_min.p, nons_mask.p, x.nonrep.p, job->ctr.p);
    if (exclusive_scan_u32_to_u64(s, blk_nons.p, nons_off.p, kb, scr2.p, tots.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&x.NNS, tots.p, sizeof x.NNS, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(x.cls.alloc(x.NNS)); RCQ(x.cls_m.alloc(x.NNS));
    launch_class_emit(s, x.gp_s, head_flag.p, N, nons_mask.p, x.cls.p, x.cls_m.p, nons_off.p);
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    return 0;
}

// Entry lists of this rank's classes (HTS:356-433).  The counters block must hold the GLOBAL first-singleton key.
int st_exact_parts(cora_hip_job *job, ExactLocal &x, ExactParts &p)
{
    hipStream_t s = job->stream;
    p.NNS = x.NNS; p.NE = 0;
    if (!x.NNS) return 0;
    DBuf<u32> cf;
    DBuf<u64> eoff, scr, tot;
    RCQ(cf.alloc(x.NNS)); RCQ(p.ce.alloc(x.NNS)); RCQ(eoff.alloc(x.NNS)); RCQ(scr.alloc(scan_scratch_elems(x.NNS))); RCQ(tot.alloc(1));
    RCQ(p.cls.alloc(x.NNS));
    HIPQ(hipMemcpyAsync(p.cls.p, x.cls.p, x.NNS * sizeof(u64), hipMemcpyDeviceToDevice, s));
    launch_class_info(s, x.rec_s, x.cls_m.p, x.cls.p, (u32)x.NNS, job->ctr.p, cf.p, p.ce.p);
    if (exclusive_scan_u32_to_u64(s, p.ce.p, eoff.p, x.NNS, scr.p, tot.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&p.NE, tot.p, sizeof p.NE, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(p.ent.alloc(p.NE)); RCQ(p.edir.alloc(p.NE));
    launch_class_entries(s, x.rec_s, x.gp_s, x.cls_m.p, x.cls.p, (u32)x.NNS, cf.p, eoff.p, p.ent.p, p.edir.p, job->ctr.p);
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    return 0;
}

// All classes (of all ranks, any order; ent laid out in the order of cls): sort by representative, block-merge
// flags (HTS:455-497), copy back, serialise F1 (HTS:572-635).
int st_exact_finish(cora_hip_job *job, ExactParts &p, FinishStats &fst, double &host_ms)
{
    hipStream_t s = job->stream;
    const DevParams &dp = job->dp;
    const u64 n = p.NNS;
    if (n >= (1ull << 32)) return fail(19, "too many classes");
    if (n) {
        DBuf<u64> key, key_tmp, eoff, eoff_s, scr;
        DBuf<u32> ce_s;
        DBuf<u8> mp;
        RCQ(key.alloc(n)); RCQ(key_tmp.alloc(n)); RCQ(eoff.alloc(n)); RCQ(eoff_s.alloc(n)); RCQ(scr.alloc(scan_scratch_elems(n)));
        RCQ(ce_s.alloc(n)); RCQ(mp.alloc(n));
        if (exclusive_scan_u32_to_u64(s, p.ce.p, eoff.p, n, scr.p, nullptr)) return fail(100, "scan launch failed");
        launch_rekey(s, p.cls.p, (u32)n, key.p);
        SortArrays ck, ct;
        ck.a = key.p; ct.a = key_tmp.p;
        if (radix_sort(s, ck, ct, n, 32, 32 + bits_for(dp.G), job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        launch_gather_parts(s, ck.a, (u32)n, p.ce.p, eoff.p, ce_s.p, eoff_s.p);
        launch_class_merge(s, ck.a, (u32)n, ce_s.p, eoff_s.p, p.ent.p, p.edir.p, mp.p);
        // F1 bytes on the device: sizes, offsets, bytes; only the bytes cross to the host
        const int cw = job->prm.chr_short ? 2 : 1;
        DBuf<u32> csize;
        DBuf<u64> coff, tot;
        RCQ(csize.alloc(n)); RCQ(coff.alloc(n)); RCQ(tot.alloc(1));
        DevCounters h;
        ZERO_COUNTERS(job, emitted_exact, 1);
        launch_exact_size(s, cw, ce_s.p, mp.p, (u32)n, csize.p, job->ctr.p);
        if (exclusive_scan_u32_to_u64(s, csize.p, coff.p, n, scr.p, tot.p)) return fail(100, "scan launch failed");
        u64 total = 0;
        HIPQ(hipMemcpyAsync(&total, tot.p, sizeof total, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        DBuf<u8> stream;
        RCQ(stream.alloc(total));
        launch_exact_write(s, dp, cw, ce_s.p, eoff_s.p, p.ent.p, p.edir.p, mp.p, (u32)n, csize.p, coff.p, stream.p, job->ctr.p);
        View<u8> bytes;
        join_warm_exact(job);                                                  // (a first build prepares these buffers beside the kernels, job.hip)
        RCQ(d2h(job, job->h_exact, bytes, (const u8 *)stream.p, total));      // (its own landing area: h_stream may still be in preparation, job.hip)
        RCQ(read_counters(job, h));
        HIPQ(hipGetLastError());
        const auto h0 = std::chrono::steady_clock::now();
        if (!job->exact_bytes.reserve_uninit(total + 16)) return fail(101, "host allocation of %llu bytes for <Exact> failed", (unsigned long long)total + 16);
        if (total) memcpy(job->exact_bytes.p, bytes.p, total);
        job->exact_bytes.n = total;
        fst.exact_nodes = h.emitted_exact;
        host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    } else {
        join_warm_exact(job);
        if (!job->exact_bytes.reserve_uninit(16)) return fail(101, "host allocation failed");
        job->exact_bytes.n = 0;
        fst.exact_nodes = 0;
    }
    return 0;
}

// representatives (set bits of the position bitmap) inside words [word_lo, word_hi), in position order
int st_reps(cora_hip_job *job, const u32 *repmask, u64 word_lo, u64 word_hi, DBuf<u32> &reps, u64 &n_out)
{
    hipStream_t s = job->stream;
    const u64 nw = word_hi - word_lo;
    const u32 nb = (u32)((nw + 255) / 256);
    n_out = 0;
    if (!nw) { RCQ(reps.alloc(1)); return 0; }
    DBuf<u32> bc; DBuf<u64> boff, scr, tot;
    RCQ(bc.alloc(nb)); RCQ(boff.alloc(nb)); RCQ(scr.alloc(scan_scratch_elems(nb))); RCQ(tot.alloc(1));
    launch_bitmap_count(s, repmask + word_lo, nw, bc.p, nb);
    if (exclusive_scan_u32_to_u64(s, bc.p, boff.p, nb, scr.p, tot.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&n_out, tot.p, sizeof n_out, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(reps.alloc(n_out));
    launch_bitmap_emit(s, repmask + word_lo, nw, boff.p, nb, reps.p, (u32)(word_lo * 32));
    HIPQ(hipStreamSynchronize(s));
    return 0;
}

// One seed: sort the (code, position) records, compact the run members, verify, append the pair records
// (HTS:2671-2800, 2138-2265).  codes holds n records in which equal codes appear in increasing position.
// sk_b / sk_c: the records' sketches (k_seed_codes_all), sorted along with them; either or both may be null.
// from: as in st_exact_classes (the bit range is seed_sort_lo / _hi); codes / sk_b are then only sized for the n kept records.
int st_seed_verify(cora_hip_job *job, DBuf<u64> &codes, u64 n, int seed, PairStore &ps, u32 *first_hist, u64 *sk_b, u32 *sk_c, const SortSource *from,
                   int b_chain_n, bool lds)
{
    DevParams dp = job->dp;
    // (the candidate kernels' copy: which sketch words hold an earlier seed's code; CORA_SK_EARLY=0: none is looked at, the A/B switch)
    dp.sk_early = (getenv("CORA_SK_EARLY") && atoi(getenv("CORA_SK_EARLY")) == 0) ? 0u : sketch_early_words(seed - 1, dp.nseeds);
    hipStream_t s = job->stream;
    if (!n) return 0;
    DBuf<u64> codes_tmp, tots;
    RCQ(codes_tmp.alloc(n)); RCQ(tots.alloc(2));
    DBuf<u64> skb_tmp;
    DBuf<u32> skc_tmp;
    SortArrays ck, ct;
    ck.a = codes.p; ct.a = codes_tmp.p;
    if (sk_b && b_chain_n > 1) { ck.b = sk_b; ck.b_chain_n = b_chain_n; ck.b_stride = (size_t)n; ct.b = sk_b + (size_t)n; }      // (ct.b: a place holder, the chain is used)
    else if (sk_b) { RCQ(skb_tmp.alloc(n)); ck.b = sk_b; ct.b = skb_tmp.p; }               // the sketches travel with their records
    if (sk_c) { RCQ(skc_tmp.alloc(n)); ck.c = sk_c; ct.c = skc_tmp.p; }
    // the records are a (code << 32 | position); with a b array the sort's logical key is a:b, i.e. a's bits sit 64 higher
    const int key_lo = seed_sort_lo(sk_b != nullptr, lds), key_hi = seed_sort_hi(sk_b != nullptr);
    NextDigitLease digit_bytes(job->sort_scratch, n);
    if (radix_sort(s, ck, ct, n, key_lo, key_hi, job->sort_scratch, &job->sort_timing, first_hist, from)) return fail(100, "radix sort failed");
    // lds: the records are now sorted on the top 24 code bits; the candidate kernels order the rest in LDS
    const u64 ab = lds ? verify_blocks_ps(n) : verify_blocks(n);  // workgroups (tiles) of the candidate kernels
    const u64 vt = lds ? 2 * ab : ab;                             // per-tile entries (PS: the balanced kernel's, then the walk's)
    u64 totals[2] = {0, 0};                                      // slots, comparisons
    DBuf<PairRec> slots;
    {
        DBuf<u32> cnt_slots, blk_slots;
        DBuf<u64> blk_cand, slot_off, scr_a, staged;
        DBuf<u8> blk_staged;
        RCQ(cnt_slots.alloc(vt * 256)); RCQ(blk_slots.alloc(vt)); RCQ(blk_cand.alloc(vt));
        RCQ(slot_off.alloc(vt)); RCQ(scr_a.alloc(scan_scratch_elems(vt)));
        RCQ(staged.alloc(lds ? verify_staging_elems_ps(n) : verify_staging_elems(n))); RCQ(blk_staged.alloc(vt));
        DBuf<u8> blk_mode;
        static const bool pairs_kernel = !(getenv("CORA_PAIRS") && atoi(getenv("CORA_PAIRS")) == 0);      // (0: the walk does every tile; four-pass mode only)
        if (pairs_kernel || lds) RCQ(blk_mode.alloc(ab));
        DBuf<u32> tile_list[3], list_n;
        u32 n_listed[3] = {0, 0, 0};
        if (lds) {
            // (the walk writes its half of the per-tile arrays for the tiles it visits only)
            HIPQ(hipMemsetAsync(blk_slots.p + ab, 0, ab * sizeof(u32), s));
            HIPQ(hipMemsetAsync(blk_cand.p + ab, 0, ab * sizeof(u64), s));
            HIPQ(hipMemsetAsync(blk_staged.p + ab, 0, ab, s));
            launch_candidates_count_ps(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, blk_cand.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p);
            for (auto &l : tile_list) RCQ(l.alloc(ab));
            RCQ(list_n.alloc(3));
            HIPQ(hipMemsetAsync(list_n.p, 0, 3 * sizeof(u32), s));
            launch_tile_list_ps(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 0, tile_list[0].p, list_n.p);
            HIPQ(hipMemcpyAsync(n_listed, list_n.p, sizeof(u32), hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            launch_candidates_walk_count_ps(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, blk_cand.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p,
                                            tile_list[0].p, n_listed[0]);
            launch_tile_list_ps(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 1, tile_list[1].p, list_n.p + 1);
            launch_tile_list_ps(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 2, tile_list[2].p, list_n.p + 2);
            HIPQ(hipMemcpyAsync(n_listed, list_n.p, sizeof n_listed, hipMemcpyDeviceToHost, s));
        } else {
            launch_candidates_count(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, blk_cand.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p);
            // the passes that have work in few tiles only take those tiles as lists: [0] the tiles left to the walk, [1] / [2] the
            // tiles whose slots the write pass of the balanced kernel / of the walk must produce
            if (pairs_kernel) {
                for (auto &l : tile_list) RCQ(l.alloc(ab));
                RCQ(list_n.alloc(3));
                HIPQ(hipMemsetAsync(list_n.p, 0, 3 * sizeof(u32), s));
                launch_tile_list(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 0, tile_list[0].p, list_n.p);
                HIPQ(hipMemcpyAsync(n_listed, list_n.p, sizeof(u32), hipMemcpyDeviceToHost, s));
                HIPQ(hipStreamSynchronize(s));
                launch_candidates_walk_count(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, blk_cand.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p,
                                             tile_list[0].p, n_listed[0]);
                launch_tile_list(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 1, tile_list[1].p, list_n.p + 1);
                launch_tile_list(s, ab, blk_mode.p, blk_slots.p, blk_staged.p, 2, tile_list[2].p, list_n.p + 2);
                HIPQ(hipMemcpyAsync(n_listed, list_n.p, sizeof n_listed, hipMemcpyDeviceToHost, s));
            }
        }
        if (exclusive_scan_u32_to_u64(s, blk_slots.p, slot_off.p, vt, scr_a.p, tots.p)) return fail(100, "scan launch failed");
        HIPQ(hipMemsetAsync(tots.p + 1, 0, sizeof(u64), s));
        launch_sum_u64(s, blk_cand.p, vt, tots.p + 1);
        HIPQ(hipMemcpyAsync(totals, tots.p, sizeof totals, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        job->stats.candidate_pairs += totals[1];
        job->stats.verify_slots += totals[0];
        if (sk_b || sk_c) job->stats.seed_sketch = 1;
        if (totals[0] == 0) { HIPQ(hipGetLastError()); return 0; }
        RCQ(slots.alloc(totals[0]));
        if (lds) launch_candidates_write_ps(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, slot_off.p, slots.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p,
                                           tile_list[1].p, n_listed[1], tile_list[2].p, n_listed[2]);
        else launch_candidates_write(s, dp, ck.a, n, cnt_slots.p, blk_slots.p, slot_off.p, slots.p, ck.b, ck.c, staged.p, blk_staged.p, blk_mode.p,
                                     tile_list[1].p, n_listed[1], tile_list[2].p, n_listed[2]);
    }
    const u64 n_slots = totals[0], jb = (n_slots + 255) / 256;
    DBuf<u32> blk_valid;
    DBuf<u64> pair_off, scr_j;
    RCQ(blk_valid.alloc(jb)); RCQ(pair_off.alloc(jb)); RCQ(scr_j.alloc(scan_scratch_elems(jb)));
    launch_judge(s, dp, job->pk.p, slots.p, n_slots, seed, blk_valid.p);
    if (exclusive_scan_u32_to_u64(s, blk_valid.p, pair_off.p, jb, scr_j.p, tots.p)) return fail(100, "scan launch failed");
    u64 n_new = 0;
    HIPQ(hipMemcpyAsync(&n_new, tots.p, sizeof n_new, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    if (ps.n + n_new > ps.cap || !ps.pairs.p) {                  // grow, keeping the earlier seeds' records
        const u64 want = (ps.n + n_new) * 2 + (1u << 20);
        DBuf<PairRec> bigger;
        RCQ(bigger.alloc(want));
        if (ps.n) HIPQ(hipMemcpyAsync(bigger.p, ps.pairs.p, ps.n * sizeof(PairRec), hipMemcpyDeviceToDevice, s));
        HIPQ(hipStreamSynchronize(s));
        std::swap(ps.pairs.p, bigger.p); std::swap(ps.pairs.n, bigger.n); std::swap(ps.pairs.pool, bigger.pool);
        ps.cap = want;
    }
    if (n_new) launch_slot_compact(s, slots.p, n_slots, pair_off.p, ps.pairs.p + ps.n);
    ps.n += n_new;
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    return 0;
}

// Directed items -> (source,target) order -> lists -> diagonal block merge (HTS:1022-1270) -> host: final order and
// F2 bytes (HTS:1353-1432).  ik/im hold M items in any order.
int st_items_finish(cora_hip_job *job, DBuf<u64> &ik, DBuf<u32> &im, u64 M, FinishStats &fst, double &host_ms)
{
    {
        const auto w0 = std::chrono::steady_clock::now();
        join_warm(job);                              // (the host buffers this stage fills may still be in preparation)
        if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] items: waited %.3f ms for the host buffers\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    }
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    InexactDeviceResult ir;                 // the items of the lists the host finishes
    View<u64> packed_host;
    View<u8> stream_host;
    View<HostNode> hnodes_host;
    u64 device_nodes = 0, big_nodes = 0;
    StreamArrival arrival;
    bool have_arrival = false;
    static const u32 zero_node = 0;
    if (M >= (1ull << 32)) return fail(19, "too many directed items (%llu)", (unsigned long long)M);
    job->stats.inexact_items = M;
    if (M) {
        DBuf<u64> ik_tmp, nexcl, scr, tot;
        DBuf<u32> im_tmp, hf, nstart;
        DBuf<int16_t> blen;
        trace_mark(job, "items: enter", 0);
        RCQ(ik_tmp.alloc(M)); RCQ(im_tmp.alloc(M));
        trace_mark(job, "items: sort buffers", 1);
        SortArrays kk, kt;
        kk.a = ik.p; kk.c = im.p; kt.a = ik_tmp.p; kt.c = im_tmp.p;
        const int gb = bits_for(dp.G);
        NextDigitLease digit_bytes(job->sort_scratch, M);
        if (gb == 32) {                                     // (target then source = all 64 bits: one sort, no histogram of key words in between)
            if (radix_sort(s, kk, kt, M, 0, 64, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        } else {
            if (radix_sort(s, kk, kt, M, 0, gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
            if (radix_sort(s, kk, kt, M, 32, 32 + gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        }
        trace_mark(job, "items: two sorts", 2);
        const u64 item_words = (M + 63) / 64;               // list heads as a mask: one word per 64 items
        DBuf<u64> head_mask;
        RCQ(head_mask.alloc(item_words)); RCQ(hf.alloc(item_words)); RCQ(nexcl.alloc(item_words));
        RCQ(scr.alloc(scan_scratch_elems(M))); RCQ(tot.alloc(1)); RCQ(blen.alloc(M));
        trace_mark(job, "items: flag buffers", 3);
        DevCounters h;
        ZERO_COUNTERS(job, n_dirty, 1);
        launch_node_flags(s, kk.a, M, head_mask.p, hf.p, job->ctr.p);                    // hf: heads per word, nexcl: lists before each word
        if (exclusive_scan_u32_to_u64(s, hf.p, nexcl.p, item_words, scr.p, tot.p)) return fail(100, "scan launch failed");
        u64 n_nodes = 0;
        HIPQ(hipMemcpyAsync(&n_nodes, tot.p, sizeof n_nodes, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        RCQ(nstart.alloc(n_nodes + 1));
        launch_node_starts(s, head_mask.p, nexcl.p, M, nstart.p);
        const u32 m32 = (u32)M;
        HIPQ(hipMemcpyAsync(nstart.p + n_nodes, &m32, sizeof(u32), hipMemcpyHostToDevice, s));
        DBuf<u32> succ;
        RCQ(succ.alloc(M));
        launch_chain_link(s, dp, kk.a, kk.c, head_mask.p, nexcl.p, nstart.p, (u32)n_nodes, M, succ.p, blen.p);
        {
            // the surviving items, listed; each walks its chain (head_mask and hf / nexcl have served: reused for the list's mask and counts)
            const u64 nb = (item_words + 255) / 256;
            DBuf<u64> aoff;
            DBuf<u32> acnt, alive;
            RCQ(acnt.alloc(nb)); RCQ(aoff.alloc(nb));
            launch_alive_mask(s, blen.p, M, head_mask.p);
            launch_mask_count(s, head_mask.p, item_words, acnt.p);
            if (exclusive_scan_u32_to_u64(s, acnt.p, aoff.p, nb, scr.p, tot.p)) return fail(100, "scan launch failed");
            u64 n_alive = 0;
            HIPQ(hipMemcpyAsync(&n_alive, tot.p, sizeof n_alive, hipMemcpyDeviceToHost, s));
            HIPQ(hipStreamSynchronize(s));
            RCQ(alive.alloc(n_alive));
            launch_mask_emit(s, head_mask.p, item_words, aoff.p, alive.p);
            launch_chain_walk(s, succ.p, alive.p, n_alive, blen.p);
            HIPQ(hipStreamSynchronize(s));              // (alive lives on this frame)
        }
        HIPQ(hipGetLastError());
        trace_mark(job, "items: lists + chains", 4);
        // ---- bytes: lists the device can order itself are written here; the others go to the host (kernels.hip) ----
        const int cw = job->prm.chr_short ? 2 : 1;
        const u32 nn = (u32)n_nodes;
        DBuf<u32> run_head, node_tie, run_dirty, nsize, hflag, hcnt;
        DBuf<u64> run_excl, noff, hidx, hoff, tots;
        RCQ(run_head.alloc(nn)); RCQ(node_tie.alloc(nn)); RCQ(run_dirty.alloc(nn)); RCQ(nsize.alloc(nn)); RCQ(hflag.alloc(nn)); RCQ(hcnt.alloc(nn));
        RCQ(run_excl.alloc(nn)); RCQ(noff.alloc(nn)); RCQ(hidx.alloc(nn)); RCQ(hoff.alloc(nn)); RCQ(tots.alloc(4));
        ZERO_COUNTERS(job, emitted_nodes, 1);
        ZERO_COUNTERS(job, big_nodes, 1);
        launch_node_class(s, dp, kk.a, nstart.p, nn, run_head.p, node_tie.p);
        if (exclusive_scan_u32_to_u64(s, run_head.p, run_excl.p, nn, scr.p, nullptr)) return fail(100, "scan launch failed");
        HIPQ(hipMemsetAsync(run_dirty.p, 0, (size_t)nn * sizeof(u32), s));
        launch_run_dirty(s, node_tie.p, run_excl.p, run_head.p, nn, run_dirty.p);
        DBuf<u32> size_counts;
        RCQ(size_counts.alloc(2 * (((size_t)nn + 255) / 256)));
        launch_node_size(s, dp, cw, nstart.p, blen.p, run_excl.p, run_head.p, run_dirty.p, nn, nsize.p, hflag.p, hcnt.p, size_counts.p, job->ctr.p);
        if (exclusive_scan_u32_to_u64(s, nsize.p, noff.p, nn, scr.p, tots.p)) return fail(100, "scan launch failed");
        if (exclusive_scan_u32_to_u64(s, hflag.p, hidx.p, nn, scr.p, tots.p + 1)) return fail(100, "scan launch failed");
        if (exclusive_scan_u32_to_u64(s, hcnt.p, hoff.p, nn, scr.p, tots.p + 2)) return fail(100, "scan launch failed");
        // the lists the device writes, listed: most lists have lost all their items to the block merge (kernels.hip, k_node_emit_list)
        const u64 size_blocks = ((u64)nn + 255) / 256;
        DBuf<u64> eoff;
        DBuf<u32> elist;
        RCQ(eoff.alloc(size_blocks + 1));
        if (exclusive_scan_u32_to_u64(s, size_counts.p, eoff.p, size_blocks, scr.p, tots.p + 3)) return fail(100, "scan launch failed");
        HIPQ(hipMemcpyAsync(eoff.p + size_blocks, tots.p + 3, sizeof(u64), hipMemcpyDeviceToDevice, s));
        u64 totals[4] = {0, 0, 0, 0};                       // stream bytes, host lists, their items, device-written lists
        HIPQ(hipMemcpyAsync(totals, tots.p, sizeof totals, hipMemcpyDeviceToHost, s));
        // the byte stream is written and sent in up to STREAM_PIECES pieces of consecutive lists: where each piece starts.  (What the
        // host still has to do when the device is done is the last piece's copy and its move into place: 7 ms with four pieces.)
        constexpr int PIECES = STREAM_PIECES;
        const int pieces = nn >= (1u << 20) ? PIECES : 1;
        u32 piece_node[PIECES + 1];
        u64 piece_off[PIECES + 1], piece_list[PIECES + 1];      // ... in the byte stream, in the list of device-written lists
        for (int c = 0; c <= pieces; c++) piece_node[c] = (u32)((u64)nn * c / pieces) & ~255u;
        piece_node[pieces] = nn;
        for (int c = 1; c < pieces; c++) {
            HIPQ(hipMemcpyAsync(&piece_off[c], noff.p + piece_node[c], sizeof(u64), hipMemcpyDeviceToHost, s));
            HIPQ(hipMemcpyAsync(&piece_list[c], eoff.p + piece_node[c] / 256, sizeof(u64), hipMemcpyDeviceToHost, s));
        }
        HIPQ(hipStreamSynchronize(s));
        piece_off[0] = 0; piece_off[pieces] = totals[0];
        piece_list[0] = 0; piece_list[pieces] = totals[3];
        RCQ(elist.alloc(totals[3]));
        launch_node_emit_list(s, nsize.p, nn, eoff.p, elist.p);
        trace_mark(job, "items: sizes + scans", 5);
        DBuf<u8> stream;
        DBuf<HostNode> hnodes;
        DBuf<u64> hk;
        DBuf<u32> hm;
        DBuf<int16_t> hbl;
        DBuf<u32> ord;
        RCQ(ord.alloc(M));
        RCQ(stream.alloc(totals[0])); RCQ(hnodes.alloc(totals[1])); RCQ(hk.alloc(totals[2])); RCQ(hm.alloc(totals[2])); RCQ(hbl.alloc(totals[2]));
        RCQ(job->h_stream.reserve(totals[0]));
        stream_host.p = job->h_stream.p; stream_host.n = totals[0];
        // 1. the lists the host finishes: their items leave first (a few MB), so that the host can work on them while
        //    the device writes the others
        launch_node_write(s, dp, cw, kk.a, kk.c, blen.p, nstart.p, 0, nn, nsize.p, noff.p, hflag.p, hidx.p, hoff.p, node_tie.p, stream.p,
                          hnodes.p, hk.p, hm.p, hbl.p, ord.p, 1);
        // (wait for the kernel BEFORE queueing copies to the host on this stream: queued behind pending kernels such copies
        //  made the second run of every job 5.5 ms slower -- once, a runtime-internal set-up; bisected with CORA_SYNC_MASK)
        HIPQ(hipStreamSynchronize(s));
        RCQ(d2h(job, job->h_hnodes, hnodes_host, (const HostNode *)hnodes.p, totals[1]));
        RCQ(d2h(job, job->h_key, ir.key, (const u64 *)hk.p, totals[2]));
        RCQ(d2h(job, job->h_meta, ir.meta, (const u32 *)hm.p, totals[2]));
        RCQ(d2h(job, job->h_blen, ir.blen, (const int16_t *)hbl.p, totals[2]));
        RCQ(read_counters(job, h));
        device_nodes = h.emitted_nodes;
        big_nodes = h.big_nodes;
        if (h.n_dirty) {
            if (job->packed_in_flight) { packed_host.p = job->h_packed.p; packed_host.n = dp.n_words + 4; }     // sent right after k_pack (job.hip)
            else RCQ(d2h(job, job->h_packed, packed_host, (const u64 *)job->pk.p, dp.n_words + 4));
        }
        HIPQ(hipStreamSynchronize(s));
        // 2. the byte stream of all other lists, written and sent in pieces: a piece leaves on the copy stream as soon as it
        //    is written, while the next one is being written (PCIe at 55 GB/s takes longer over the whole stream than
        //    k_node_write does); the host moves each piece to its place when it has arrived (assemble_inexact below)
        arrival.upto.clear();
        for (int c = 0; c < pieces; c++) {
            launch_node_write(s, dp, cw, kk.a, kk.c, blen.p, nstart.p, (u32)piece_list[c], (u32)piece_list[c + 1], nsize.p, noff.p, hflag.p, hidx.p, hoff.p,
                              node_tie.p, stream.p, hnodes.p, hk.p, hm.p, hbl.p, ord.p, 2, elist.p);
            HIPQ(hipEventRecord(job->ev_copy[EV_WRITTEN + c], s));
            HIPQ(hipStreamWaitEvent(job->copy_stream, job->ev_copy[EV_WRITTEN + c], 0));
            if (piece_off[c + 1] > piece_off[c])
                HIPQ(hipMemcpyAsync(job->h_stream.p + piece_off[c], stream.p + piece_off[c], piece_off[c + 1] - piece_off[c], hipMemcpyDeviceToHost,
                                    job->copy_stream));
            HIPQ(hipEventRecord(job->ev_copy[EV_ARRIVED + c], job->copy_stream));
            arrival.upto.push_back((size_t)piece_off[c + 1]);
        }
        HIPQ(hipGetLastError());
        arrival.wait = [job](size_t c) {
            static const bool trace = getenv("CORA_TRACE") != nullptr;
            const auto w0 = std::chrono::steady_clock::now();
            const bool ok = hipEventSynchronize(job->ev_copy[EV_ARRIVED + c]) == hipSuccess;
            if (trace) fprintf(stderr, "[cora]   stream piece %zu: waited %.3f ms for it\n", c, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
            return ok;
        };
        have_arrival = true;
        if (packed_host.n && job->packed_in_flight) HIPQ(hipEventSynchronize(job->ev_copy[EV_PACKED_ARRIVED]));      // (sent 0.7 s ago)
        trace_mark(job, "items: copies queued", 7);
        // list starts of the compacted host items
        RCQ(job->h_node_start.reserve(totals[1] + 1));
        for (u64 i = 0; i < totals[1]; i++) job->h_node_start.p[i] = (u32)hnodes_host[i].item_off;
        job->h_node_start.p[totals[1]] = (u32)totals[2];
        ir.node_start.p = job->h_node_start.p; ir.node_start.n = totals[1] + 1;
    } else {
        ir.node_start.p = &zero_node; ir.node_start.n = 1;
    }
    HIPQ(hipEventRecord(job->ev[4], s));
    const auto h0 = std::chrono::steady_clock::now();
    const bool assembled = assemble_inexact(job->hg, stream_host, hnodes_host, ir, packed_host, device_nodes, job->inexact_bytes, fst,
                                            job->host_threads, have_arrival ? &arrival : nullptr);
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipStreamSynchronize(job->copy_stream));             // (also when the assembly gave up half-way: nothing may stay in flight)
    if (!assembled) return fail(101, "host allocation for the %zu bytes of <Inexact> failed, or a copy from the device did", stream_host.size());
    fst.big_nodes = big_nodes;                           // lists above 16 items, counted on the device (most never reach the host)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    host_ms += ms;
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] inexact host finish %.3f ms (%zu stream bytes, %zu host lists, %zu host items)\n", ms,
                                      stream_host.size(), hnodes_host.size(), ir.key.size());
    return 0;
}
