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
    DevCounters h;
    RCQ(read_counters(job, h));
    if (h.format_errors)
        return fail(51, "FASTA image does not have the fixed line width the reference's loader assumes (%llu misplaced line ends)",
                    (unsigned long long)h.format_errors);
    return 0;
}

// hashed records (hash|flags, gpos) of the valid windows of blocks [block_lo, block_lo + n_blocks) of 256 positions
int st_kmers(cora_hip_job *job, u32 block_lo, u32 n_blocks, DBuf<u64> &rec, DBuf<u32> &gp, u64 &n_out)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    DBuf<u32> bc;
    DBuf<u64> boff, scr, tot;
    RCQ(bc.alloc(n_blocks)); RCQ(boff.alloc(n_blocks)); RCQ(scr.alloc(scan_scratch_elems(n_blocks))); RCQ(tot.alloc(1));
    launch_kmer_count_range(s, dp, job->bad.p, bc.p, block_lo, n_blocks);
    if (exclusive_scan_u32_to_u64(s, bc.p, boff.p, n_blocks, scr.p, tot.p)) return fail(100, "scan launch failed");
    u64 n = 0;
    HIPQ(hipMemcpyAsync(&n, tot.p, sizeof n, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    n_out = n;
    RCQ(rec.alloc(n)); RCQ(gp.alloc(n));
    launch_kmer_emit_hashed_range(s, dp, job->pk.p, job->bad.p, boff.p, block_lo, n_blocks, rec.p, gp.p);
    HIPQ(hipStreamSynchronize(s));
    return 0;
}

// Sort the records, group equal canonical K-mers exactly, mark the non-representatives, list the classes with >= 2
// entries and find the first singleton (HTS:2516-2578).  x.rec / x.gp hold x.N records in which equal K-mers
// appear in increasing position.
int st_exact_classes(cora_hip_job *job, ExactLocal &x)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    cora_hip_stats &st = job->stats;
    const u64 N = x.N;
    if (N >= (1ull << 32) - 1) return fail(19, "too many K-mers for 32-bit indices");
    DBuf<u32> head_flag, mixed_list;
    const u32 mixed_cap = 4096;
    RCQ(x.trec.alloc(N)); RCQ(x.tgp.alloc(N)); RCQ(head_flag.alloc(N)); RCQ(mixed_list.alloc(mixed_cap));
    // Equal canonical keys have equal hashes: sort 12-byte (hash, position) records on the top hb hash bits
    // (6-7 passes) instead of 20-byte full keys on 2K bits (13 passes); collisions are found and repaired below.
    const int hb = job->hash_bits ? job->hash_bits : (N <= (1ull << 27) ? 48 : 56);
    SortArrays keys, tmp;
    keys.a = x.rec.p; keys.c = x.gp.p; tmp.a = x.trec.p; tmp.c = x.tgp.p;
    if (radix_sort(s, keys, tmp, N, 64 - hb, 64, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
    DevCounters h;
    RCQ(read_counters(job, h));
    h.n_mixed = 0; h.fallback = 0;
    RCQ(write_counters(job, h));
    launch_same_detect(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_list.p, mixed_cap, job->ctr.p);
    RCQ(read_counters(job, h));
    bool fallback = false;
    st.hash_mixed_runs += h.n_mixed;
    if (h.n_mixed > mixed_cap) fallback = true;
    else if (h.n_mixed) {
        launch_mixed_fix(s, dp, job->pk.p, keys.a, keys.c, N, hb, mixed_list.p, (u32)h.n_mixed, job->ctr.p);
        RCQ(read_counters(job, h));
        h.n_mixed = 0;
        RCQ(write_counters(job, h));
        launch_same_detect(s, dp, job->pk.p, keys.a, keys.c, N, hb, head_flag.p, mixed_list.p, mixed_cap, job->ctr.p);
        RCQ(read_counters(job, h));
        if (h.n_mixed || h.fallback) fallback = true;
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
    // later members of classes = the positions that are NOT representatives
    RCQ(x.nonrep.alloc(dp.n_words));
    HIPQ(hipMemsetAsync(x.nonrep.p, 0, dp.n_words * sizeof(u32), s));
    launch_nonrep_mark(s, head_flag.p, x.gp_s, N, x.nonrep.p);
    // class index of every record, class starts
    DBuf<u64> cid, scr0, tot0;
    RCQ(cid.alloc(N)); RCQ(scr0.alloc(scan_scratch_elems(N))); RCQ(tot0.alloc(1));
    if (exclusive_scan_u32_to_u64(s, head_flag.p, cid.p, N, scr0.p, tot0.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&x.NR, tot0.p, sizeof x.NR, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(x.class_start.alloc(x.NR + 1));
    launch_node_starts(s, head_flag.p, cid.p, N, x.class_start.p);
    const u32 n32 = (u32)N;
    HIPQ(hipMemcpyAsync(x.class_start.p + x.NR, &n32, sizeof(u32), hipMemcpyHostToDevice, s));
    cid.release(); scr0.release(); head_flag.release();

    const u32 kb = (u32)((x.NR + 255) / 256);
    DBuf<u32> blk_nons;
    DBuf<u64> blk_min, nons_off, scr2, tots;
    RCQ(blk_nons.alloc(kb)); RCQ(blk_min.alloc(kb)); RCQ(nons_off.alloc(kb));
    RCQ(scr2.alloc(scan_scratch_elems(kb))); RCQ(tots.alloc(1));
    launch_class_count(s, x.rec_s, x.gp_s, x.class_start.p, (u32)x.NR, blk_nons.p, (unsigned long long *)blk_min.p, job->ctr.p);
    if (exclusive_scan_u32_to_u64(s, blk_nons.p, nons_off.p, kb, scr2.p, tots.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&x.NNS, tots.p, sizeof x.NNS, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(x.cls.alloc(x.NNS));
    launch_class_emit(s, x.rec_s, x.gp_s, x.class_start.p, (u32)x.NR, x.cls.p, nons_off.p);
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
    launch_class_info(s, x.rec_s, x.class_start.p, x.cls.p, (u32)x.NNS, job->ctr.p, cf.p, p.ce.p);
    if (exclusive_scan_u32_to_u64(s, p.ce.p, eoff.p, x.NNS, scr.p, tot.p)) return fail(100, "scan launch failed");
    HIPQ(hipMemcpyAsync(&p.NE, tot.p, sizeof p.NE, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    RCQ(p.ent.alloc(p.NE)); RCQ(p.edir.alloc(p.NE));
    launch_class_entries(s, x.rec_s, x.gp_s, x.class_start.p, x.cls.p, (u32)x.NNS, cf.p, eoff.p, p.ent.p, p.edir.p);
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
    ExactDeviceResult er;
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
        RCQ(d2h(job, job->h_cls, er.cls, (const u64 *)ck.a, n));
        RCQ(d2h(job, job->h_cls_entries, er.cls_entries, (const u32 *)ce_s.p, n));
        RCQ(d2h(job, job->h_ent_off, er.ent_off, (const u64 *)eoff_s.p, n));
        RCQ(d2h(job, job->h_ent, er.ent, (const u32 *)p.ent.p, p.NE));
        RCQ(d2h(job, job->h_ent_dir, er.ent_dir, (const u8 *)p.edir.p, p.NE));
        RCQ(d2h(job, job->h_merge_prev, er.merge_prev, (const u8 *)mp.p, n));
        HIPQ(hipStreamSynchronize(s));
        HIPQ(hipGetLastError());
    }
    const auto h0 = std::chrono::steady_clock::now();
    serialize_exact(job->hg, er, job->exact_bytes, fst);
    host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
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
int st_seed_verify(cora_hip_job *job, DBuf<u64> &codes, u64 n, int seed, PairStore &ps)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    if (!n) return 0;
    DBuf<u64> codes_tmp, blk_off, scr, tots;
    DBuf<u32> blk_active;
    const u32 vb = (u32)((n + 255) / 256);
    RCQ(codes_tmp.alloc(n)); RCQ(blk_active.alloc(vb)); RCQ(blk_off.alloc(vb));
    RCQ(scr.alloc(scan_scratch_elems(vb))); RCQ(tots.alloc(2));
    SortArrays ck, ct;
    ck.a = codes.p; ct.a = codes_tmp.p;
    if (radix_sort(s, ck, ct, n, 32, 64, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
    launch_active_count(s, ck.a, n, blk_active.p);
    if (exclusive_scan_u32_to_u64(s, blk_active.p, blk_off.p, vb, scr.p, tots.p)) return fail(100, "scan launch failed");
    u64 n_active = 0;
    HIPQ(hipMemcpyAsync(&n_active, tots.p, sizeof n_active, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    if (n_active == 0) return 0;
    if (n_active >= (1ull << 32)) return fail(19, "too many run members");
    DBuf<u32> active, cnt_recs, cnt_cand;
    DBuf<u64> rec_off, cand_off, scr_a;
    RCQ(active.alloc(n_active)); RCQ(cnt_recs.alloc(n_active)); RCQ(cnt_cand.alloc(n_active));
    RCQ(rec_off.alloc(n_active)); RCQ(cand_off.alloc(n_active)); RCQ(scr_a.alloc(scan_scratch_elems(n_active)));
    launch_active_emit(s, ck.a, n, blk_off.p, active.p);
    launch_verify_count(s, dp, job->pk.p, ck.a, active.p, (u32)n_active, seed, cnt_recs.p, cnt_cand.p);
    if (exclusive_scan_u32_to_u64(s, cnt_recs.p, rec_off.p, n_active, scr_a.p, tots.p)) return fail(100, "scan launch failed");
    if (exclusive_scan_u32_to_u64(s, cnt_cand.p, cand_off.p, n_active, scr_a.p, tots.p + 1)) return fail(100, "scan launch failed");
    u64 totals[2] = {0, 0};
    HIPQ(hipMemcpyAsync(totals, tots.p, sizeof totals, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    job->stats.candidate_pairs += totals[1];
    if (ps.n + totals[0] > ps.cap || !ps.pairs.p) {              // grow, keeping the earlier seeds' records
        const u64 want = (ps.n + totals[0]) * 2 + (1u << 20);
        DBuf<PairRec> bigger;
        RCQ(bigger.alloc(want));
        if (ps.n) HIPQ(hipMemcpyAsync(bigger.p, ps.pairs.p, ps.n * sizeof(PairRec), hipMemcpyDeviceToDevice, s));
        HIPQ(hipStreamSynchronize(s));
        std::swap(ps.pairs.p, bigger.p); std::swap(ps.pairs.n, bigger.n); std::swap(ps.pairs.pool, bigger.pool);
        ps.cap = want;
    }
    if (totals[0]) launch_verify_write(s, dp, job->pk.p, ck.a, active.p, (u32)n_active, seed, rec_off.p, ps.pairs.p + ps.n);
    ps.n += totals[0];
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    return 0;
}

// Directed items -> (source,target) order -> lists -> diagonal block merge (HTS:1022-1270) -> host: final order and
// F2 bytes (HTS:1353-1432).  ik/im hold M items in any order.
int st_items_finish(cora_hip_job *job, DBuf<u64> &ik, DBuf<u32> &im, u64 M, FinishStats &fst, double &host_ms)
{
    const DevParams &dp = job->dp;
    hipStream_t s = job->stream;
    InexactDeviceResult ir;
    View<u64> packed_host;
    static const u32 zero_node = 0;
    if (M >= (1ull << 32)) return fail(19, "too many directed items (%llu)", (unsigned long long)M);
    job->stats.inexact_items = M;
    if (M) {
        DBuf<u64> ik_tmp, nexcl, scr, tot;
        DBuf<u32> im_tmp, hf, nstart;
        DBuf<int16_t> blen;
        RCQ(ik_tmp.alloc(M)); RCQ(im_tmp.alloc(M));
        SortArrays kk, kt;
        kk.a = ik.p; kk.c = im.p; kt.a = ik_tmp.p; kt.c = im_tmp.p;
        const int gb = bits_for(dp.G);
        if (radix_sort(s, kk, kt, M, 0, gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        if (radix_sort(s, kk, kt, M, 32, 32 + gb, job->sort_scratch, &job->sort_timing)) return fail(100, "radix sort failed");
        RCQ(hf.alloc(M)); RCQ(nexcl.alloc(M)); RCQ(scr.alloc(scan_scratch_elems(M))); RCQ(tot.alloc(1)); RCQ(blen.alloc(M));
        DevCounters h;
        RCQ(read_counters(job, h));
        h.n_dirty = 0;
        RCQ(write_counters(job, h));
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
        RCQ(read_counters(job, h));
        if (h.n_dirty) RCQ(d2h(job, job->h_packed, packed_host, (const u64 *)job->pk.p, dp.n_words + 4));
        HIPQ(hipStreamSynchronize(s));
    } else {
        ir.node_start.p = &zero_node; ir.node_start.n = 1;
    }
    HIPQ(hipEventRecord(job->ev[4], s));
    HIPQ(hipStreamSynchronize(s));
    const auto h0 = std::chrono::steady_clock::now();
    serialize_inexact(job->hg, ir, packed_host, job->inexact_bytes, fst, job->host_threads);
    host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    return 0;
}
