// kernels.hpp -- device-side parameters and launch wrappers of the homology-table
// pipeline (kernels.hip).  HTS = /root/reference/homTable_setup.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "finish.hpp"

namespace cora {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// Everything a kernel needs to know about the run; passed by value.
struct DevParams {
    int K, d, nseeds;
    int seed_q;               // bases per seed half (8, or K/(2(d+1)) for short read-mers, HTS:2056)
    u32 line_width;           // FASTA bases per line
    u32 n_contigs;
    u64 G;                    // padded length of the global coordinate (multiple of 64)
    u64 n_words;              // G / 32
    u64 image_bytes;
    u64 kmask_lo;             // keeps the top 2*(K-32) bits of the low window word
    u64 seedmask_hi[4], seedmask_lo[4];   // one bit per base (bit 62-2i of word i/32) for seed s = index+1
    u64 pos_base;             // first position of the block range a window kernel works on (0 = whole genome)
    u32 any_windowed;         // d == 3 and some contig is compacted in 50 Mbp windows (HTS:2852-2856)
    u32 sk_early;             // candidate kernels: sketch_early_words() of the seed at hand (set per launch; 0: no word is looked at)
    u8 sig2[16];              // physical group of (first,last) base pair
    const u64 *ctg_gstart;    // [n_contigs+1] start of each contig in the global coordinate (last = G)
    const u32 *ctg_len;       // [n_contigs]
    const u64 *ctg_off;       // [n_contigs] file offset of the first base
    const u32 *ctg_window;    // [n_contigs] 0 or the window length
};

// Which words of a seed record's sketch hold the code of an EARLIER seed.  The sketch of seed s (0-based) holds the codes of seeds
// (s + 1) % NS, (s + 2) % NS, ... in this order: NS = 2: one word, c; NS = 3: the high and the low word of b; NS = 4: those and c
// (k_seed_codes_all).  Bit 0: the high word of b, bit 1: its low word, bit 2: c.  A pair that agrees on an earlier seed is reported
// by that seed and by no later one (HTS:2222-2232): the candidate kernels give it no slot.
inline u32 sketch_early_words(int s, int ns)
{
    if (ns == 2) return s == 1 ? 4u : 0u;
    u32 m = 0;
    if ((s + 1) % ns < s) m |= 1u;
    if ((s + 2) % ns < s) m |= 2u;
    if (ns == 4 && (s + 3) % ns < s) m |= 4u;
    return m;
}

// pair record meta / item meta bit layout
//   bit 0      dir
//   bits 1-7   off1, 8-14 off2, 15-21 off3   (1-based mismatch offsets, 0 = none)
//   bits 22-23 seed-1, bit 24 palindrome pass, bit 25 duplicate index (Q1)
constexpr u32 META_DIR = 1u;
__host__ __device__ inline u32 meta_pack(u32 dir, u32 o1, u32 o2, u32 o3, u32 seed0, u32 pass, u32 dup)
{
    return dir | (o1 << 1) | (o2 << 8) | (o3 << 15) | (seed0 << 22) | (pass << 24) | (dup << 25);
}
__host__ __device__ inline u32 meta_o1(u32 m) { return (m >> 1) & 127u; }
__host__ __device__ inline u32 meta_o2(u32 m) { return (m >> 8) & 127u; }
__host__ __device__ inline u32 meta_o3(u32 m) { return (m >> 15) & 127u; }
__host__ __device__ inline u32 meta_seed0(u32 m) { return (m >> 22) & 3u; }
__host__ __device__ inline u32 meta_pass(u32 m) { return (m >> 24) & 1u; }
__host__ __device__ inline u32 meta_dup(u32 m) { return (m >> 25) & 1u; }
constexpr u32 META_MATCH_MASK = (1u << 22) - 1u;   // dir + offsets: what block merging compares

struct PairRec { u32 p1, p2, meta; };

// counters block kept on the device for one run
struct DevCounters {
    unsigned long long n_mixed, fallback, first_singleton_key, n_pairs, n_candidates, n_dirty, format_errors;
    unsigned long long empty_contigs, emitted_nodes, emitted_exact, big_nodes, n_long;
    struct Huge { u32 n; u32 id[31]; } huge_entries, huge_write;   // classes too long for one workgroup (kernels.hip: HUGE_MIN)
    unsigned long long other_letters;   // reference bytes that are neither A C G T nor N (after upper-casing): IUPAC codes and the like
};


// ---- launch wrappers (all asynchronous on stream s) ----
void launch_pack(hipStream_t s, const DevParams &p, const u8 *raw, u64 *pk, u32 *bad, DevCounters *ctr);
void launch_kmer_count(hipStream_t s, const DevParams &p, const u32 *bad, u32 *block_counts, u32 n_blocks);
void launch_test_window_bits(hipStream_t s, const DevParams &p, const u32 *bad, u32 *fast /* window_bits */, u32 *slow /* window_ok */);
void launch_kmer_emit(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *bad, const u64 *block_off,
                      u32 n_blocks, u64 *key_hi, u64 *key_lo, u32 *gpos);
void launch_contig_check(hipStream_t s, const DevParams &p, const u32 *gpos_by_position, u64 n, DevCounters *ctr);
void launch_contig_scan(hipStream_t s, const DevParams &p, const u32 *bad, DevCounters *ctr);
void launch_same_detect(hipStream_t s, const DevParams &p, const u64 *pk, const u64 *rec, const u32 *gpos, u64 n, int hb,
                        u64 *head_mask /* [ceil(n/256)*4], bit l of word w = record 64w+l */, u64 *mixed_mask /* same shape: places for k_mixed_fix */);
void launch_mixed_fix(hipStream_t s, const DevParams &p, const u64 *pk, u64 *rec, u32 *gpos, u64 n, int hb, u64 *head_mask,
                      const u32 *mixed_list, u32 n_mixed, u32 *claim /* zeroed bitmap over the records */,
                      u64 *long_runs /* [2 * long_cap] (start, end) of runs above 64 records */, u32 long_cap, DevCounters *ctr);
void launch_mixed_fix_long(hipStream_t s, const DevParams &p, const u64 *pk, u64 *rec, u32 *gpos, u64 *rec_copy, u32 *gpos_copy,
                           const u64 *long_runs, u32 n_long, u64 *head_mask, DevCounters *ctr);
void launch_heads_from_keys(hipStream_t s, const DevParams &p, const u64 *pk, const u64 *hi, const u64 *lo, const u32 *gpos,
                            u64 n, u64 *head_mask, u64 *rec);
// per sorted record (head_mask bit i = record i opens a class): classes with >= 2 entries per block of 256 records and as a
// mask, the first singleton, and the later members of classes marked in the (zeroed) position bitmap nonrep
void launch_class_scan(hipStream_t s, const u64 *rec, const u32 *gpos, const u64 *head_mask, u64 n, u32 *blk_nons /* [ceil(n/256)] */,
                       unsigned long long *blk_min, u64 *nons_mask /* [ceil(n/256)*4] */, u32 *nonrep, DevCounters *ctr);
void launch_class_emit(hipStream_t s, const u32 *gpos, const u64 *head_mask, u64 n, const u64 *nons_mask,
                       u64 *cls /* rep<<32 | first record */, u32 *cls_m /* records of the class */, const u64 *nons_off);
void launch_rep_mask_words(hipStream_t s, const u32 *valid, const u32 *nonrep, u32 *rep, u64 n_words);
void launch_rep_mask(hipStream_t s, const DevParams &p, const u32 *bad, const u32 *nonrep, u32 *rep /* may alias nonrep */);
void launch_mask_count(hipStream_t s, const u64 *mask, u64 n_words, u32 *blk_count /* [ceil(n_words/256)] */);
void launch_mask_emit(hipStream_t s, const u64 *mask, u64 n_words, const u64 *blk_off, u32 *pos);
void launch_bitmap_count(hipStream_t s, const u32 *bits, u64 n_words, u32 *block_counts, u32 n_blocks);
void launch_bitmap_emit(hipStream_t s, const u32 *bits, u64 n_words, const u64 *block_off, u32 n_blocks, u32 *out, u32 base);
void launch_class_info(hipStream_t s, const u64 *rec, const u32 *cls_m, const u64 *cls, u32 n_cls, const DevCounters *ctr,
                       u32 *cls_flags, u32 *cls_entries);
void launch_class_entries(hipStream_t s, const u64 *rec, const u32 *gpos, const u32 *cls_m, const u64 *cls, u32 n_cls,
                          const u32 *cls_flags, const u64 *ent_off, u32 *ent /* gpos */, u8 *ent_dir, DevCounters *ctr);
void launch_class_merge(hipStream_t s, const u64 *cls, u32 n_cls, const u32 *cls_entries, const u64 *ent_off,
                        const u32 *ent, const u8 *ent_dir, u8 *merge_prev);
void launch_seed_codes(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *reps, u64 n_reps, int seed, u64 *out);
// every seed's (code, position) records at once + the first-digit tile histogram of each (one block per tile of the sort)
// per-seed sketch arrays (k_seed_codes_all): b = two other seeds' codes (nseeds >= 3), c = one more code (nseeds 2 and 4)
struct SeedSketches { int on; u64 *b[4]; u32 *c[4]; };
void launch_seed_codes_all(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *reps, u64 n_reps, u64 *const out[4], u32 *const hist[4],
                           const SeedSketches &sk, u32 tile_records /* radix_sort_tile_records() of the record shape */,
                           int only = -1 /* >= 0: write this (0-based) seed's arrays only */,
                           int hist_shift = 0 /* the digit the tile histograms count: code bits hist_shift .. + 7 (8 when the sort leaves the last byte to LDS) */);
// Pair verification over the n code-sorted records of one seed (kernels.hip: k_candidates, k_judge, k_slot_compact).
// candidates: one workgroup per tile of records; cnt_slots[verify_blocks(n) * 256] per thread, blk_slots / blk_cand per
// workgroup; sk_*: the sketches, sorted with the codes, or null.  judge: slots -> pair records or none, in place, and
// the reported pairs per block of 256 slots.  compact: those pairs at blk_off (scan of blk_valid) + rank in the block.
u64 verify_blocks(u64 n);
// staged[verify_staging_elems(n)], blk_staged[verify_blocks(n)]: the slots of the tiles the counting pass could keep whole
u64 verify_staging_elems(u64 n);
void launch_candidates_count(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                             const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged,
                             u8 *blk_mode /* [blocks] which kernel did the tile; null: the walk (k_candidates) does them all */);
// the walk's counting pass over the listed tiles (the ones the balanced kernel left: launch_tile_list, which 0)
void launch_candidates_walk_count(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                  const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode, const u32 *tiles, u32 n_listed);
// tiles_pairs / tiles_walk: launch_tile_list which 1 / 2 (ignored when blk_mode is null: the walk then looks at every tile)
void launch_candidates_write(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots,
                             const u64 *blk_off /* scan of blk_slots */, PairRec *slots, const u64 *sk_b, const u32 *sk_c,
                             u64 *staged, u8 *blk_staged, u8 *blk_mode, const u32 *tiles_pairs, u32 n_pairs, const u32 *tiles_walk, u32 n_walk);
// list[0 .. *count) = the tiles a candidate pass has to look at (which: 0 left to the walk, 1 / 2 write pass of the balanced kernel /
// of the walk); *count (device) must be zero
void launch_tile_list(hipStream_t s, u64 n_tiles, const u8 *blk_mode, const u32 *blk_slots, const u8 *blk_staged, int which, u32 *list, u32 *count);
// The same three passes over records sorted on the TOP 24 BITS of the code only (kernels.hip, "the last sort digit in LDS"): tiles
// cut at bucket heads, every bucket ordered in LDS first.  All per-tile arrays hold 2 x verify_blocks_ps(n) entries (the balanced
// kernel's, then the walk's); cnt_slots 2 x verify_blocks_ps(n) x 256; blk_slots / blk_cand / blk_staged must be zero on entry.
u64 verify_blocks_ps(u64 n);
u64 verify_staging_elems_ps(u64 n);
void launch_candidates_count_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode);
void launch_candidates_walk_count_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                     const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, const u8 *blk_mode, const u32 *tiles, u32 n_listed);
void launch_candidates_write_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, const u64 *blk_off,
                                PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode,
                                const u32 *tiles_pairs, u32 n_pairs, const u32 *tiles_walk, u32 n_walk);
void launch_tile_list_ps(hipStream_t s, u64 n_tiles, const u8 *blk_mode, const u32 *blk_slots, const u8 *blk_staged, int which, u32 *list, u32 *count);
void launch_judge(hipStream_t s, const DevParams &p, const u64 *pk, PairRec *slots, u64 n_slots, int seed, u32 *blk_valid /* [ceil(n_slots / 256)] */);
void launch_slot_compact(hipStream_t s, const PairRec *slots, u64 n_slots, const u64 *blk_off, PairRec *pairs);
void launch_sum_u64(hipStream_t s, const u64 *in, u64 n, u64 *out /* device, accumulated into */);
void launch_items_expand(hipStream_t s, const DevParams &p, const PairRec *pairs, u64 n_pairs, u64 *item_key, u32 *item_meta);
// heads of the lists in the sorted items as a mask (bit l of word w = item 64 w + l) + heads per word; word_first = its exclusive scan
void launch_node_flags(hipStream_t s, const u64 *item_key, u64 m, u64 *head_mask /* [ceil(m/64)] */, u32 *word_heads /* [ceil(m/64)] */, DevCounters *ctr);
void launch_node_starts(hipStream_t s, const u64 *head_mask, const u64 *word_first, u64 m, u32 *node_start);
// diagonal block merge as links: succ[i] = the item that continues item i's diagonal in the next position's list (absorbed: blen -1);
// then the surviving items (launch_alive_mask + mask count / emit) walk their chains: blen = chain length
void launch_chain_link(hipStream_t s, const DevParams &p, const u64 *item_key, const u32 *item_meta, const u64 *head_mask,
                       const u64 *word_first, const u32 *node_start, u32 n_nodes, u64 m, u32 *succ, int16_t *blen);
void launch_alive_mask(hipStream_t s, const int16_t *blen, u64 m, u64 *mask /* [ceil(m/64)] */);
void launch_chain_walk(hipStream_t s, const u32 *succ, const u32 *alive, u64 n_alive, int16_t *blen);

void launch_exact_size(hipStream_t s, int cw, const u32 *cls_entries, const u8 *merge_prev, u32 n_cls, u32 *size, DevCounters *ctr);
void launch_exact_write(hipStream_t s, const DevParams &p, int cw, const u32 *cls_entries, const u64 *ent_off, const u32 *ent, const u8 *ent_dir,
                        const u8 *merge_prev, u32 n_cls, const u32 *size, const u64 *off, u8 *stream, DevCounters *ctr);
void launch_node_class(hipStream_t s, const DevParams &p, const u64 *item_key, const u32 *node_start, u32 n_nodes, u32 *run_head, u32 *node_tie);
void launch_run_dirty(hipStream_t s, const u32 *node_tie, const u64 *run_excl, const u32 *run_head, u32 n_nodes, u32 *run_dirty);
void launch_node_size(hipStream_t s, const DevParams &p, int cw, const u32 *node_start, const int16_t *blen, const u64 *run_excl,
                      const u32 *run_head, const u32 *run_dirty, u32 n_nodes, u32 *size, u32 *hflag, u32 *hcnt, u32 *blk_counts /* [2 * ceil(n_nodes / 256)]: device-written lists per block of 256 lists, then the lists above 16 items */, DevCounters *ctr);
void launch_node_write(hipStream_t s, const DevParams &p, int cw, const u64 *item_key, const u32 *item_meta, const int16_t *blen,
                       const u32 *node_start, u32 node_lo, u32 node_hi /* lists node_lo .. node_hi-1 */, const u32 *size, const u64 *off,
                       const u32 *hflag, const u64 *hidx, const u64 *hoff, const u32 *node_tie, u8 *stream, HostNode *hnodes, u64 *h_key, u32 *h_meta, int16_t *h_blen,
                       u32 *ord /* [items] scratch of the lists above 16 items */,
                       int which /* 1: only the lists the host finishes (hflag), 2: only the others, 3: all */,
                       const u32 *elist = nullptr /* launch_node_emit_list's list: node_lo / node_hi are then places in it */);
// the lists with size != 0, in list order; blk_off: the exclusive scan of launch_node_size's first ceil(n_nodes / 256) block counts
void launch_node_emit_list(hipStream_t s, const u32 *size, u32 n_nodes, const u64 *blk_off, u32 *list);
void launch_test_list_sort(hipStream_t s, u32 *e, u64 n, long forced_depth);

}  // namespace cora

namespace cora {
void launch_keys_from_gpos(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *gpos, u64 n, u64 *hi, u64 *lo);
void launch_rekey(hipStream_t s, const u64 *cls, u32 n, u64 *key);
void launch_gather_parts(hipStream_t s, const u64 *key_sorted, u32 n, const u32 *ce, const u64 *eoff, u32 *ce_s, u64 *eoff_s);
void launch_window_counts(hipStream_t s, const u64 *item_key, u64 m, const u32 *bnd, u32 n_bnd, u32 win, u32 *wc);
void launch_dest_by_cuts(hipStream_t s, const u64 *item_key, u64 m, const u32 *cuts, u32 n_cuts, u64 *dest);
void launch_count_dest(hipStream_t s, const u64 *dest, u64 n, u32 world, unsigned long long *counts);
// emit hashed records / count windows of a block range [block_lo, block_hi) of 256 positions each
void launch_kmer_count_range(hipStream_t s, const DevParams &p, const u32 *bad, u32 *block_counts, u32 block_lo, u32 n_blocks,
                             u32 *valid /* window-validity bitmap, written for the blocks of the range */);
void launch_kmer_emit_hashed_range(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *valid, const u64 *block_off,
                                   u32 block_lo, u32 n_blocks, u64 *rec, u32 *gpos,
                                   unsigned char *first_digit = nullptr /* [records] byte fd_shift / 8 of every record (the first histogram of its sort reads it) */,
                                   int fd_shift = 0);
}  // namespace cora

namespace cora {
void launch_or_u32(hipStream_t s, u32 *acc, const u32 *other, u64 n);
void launch_add_u32(hipStream_t s, u32 *acc, const u32 *other, u64 n);
}  // namespace cora
