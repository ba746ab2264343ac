// radix_sort.hip -- stable LSD radix sort (8-bit digits) and prefix scans, written for
// CDNA4 (gfx950): 64-lane wavefronts, ballot-based match-any ranking, records staged
// through LDS so that every global store of a pass is a run of consecutive addresses.
//
// One pass = three small kernels over the [tiles][256] digit histogram plus the scatter:
//   rs_hist      per-tile digit histogram                      (reads the digit word; skipped when the producer counted)
//   rs_colsum / rs_chunkscan / rs_colscan   digit-major exclusive scan, row-coalesced, loads batched
//   rs_scatter   rank in LDS, stage, write                     (reads + writes every array)
// HBM traffic per pass and record: 8 B (hist) + 2 x record bytes (scatter).
// Tiles: 256 threads x 16 records for key-only records, 512 x 16 for records with a payload; the scatter hands runs of
// consecutive tiles to one XCD (see rs_scatter).  A single-pass variant with decoupled look-back was measured slower on
// this chip; tools/sortbench.cpp compares with copy / read ceilings and with rocPRIM.
#include "radix_sort.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace cora {

#define RS_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32 *lds_waves /*[4]*/, u32 *total)
{
    // threads 0..255 hold one value each (later threads of a bigger block pass 0 and ignore the result);
    // returns the exclusive prefix of v over threadIdx.x.  Every thread of the block must call it.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63 && wv < 4) lds_waves[wv] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { u32 s = lds_waves[w]; if (w < wv) base += s; tot += s; }
    if (total) *total = tot;
    __syncthreads();
    return base + x - v;
}

// The digit of a record: 8 key bits, or -- for the one-pass partition of records by the rank that owns them (multi-GPU
// flow, shard.hip) -- the owner of the key's top 32 bits, taken as they are (dmode 1: hash ranges) or multiplied by the
// golden ratio first (dmode 2: seed codes), scaled to `world` ranks.
__device__ __forceinline__ u32 rs_digit(u64 word, int shift, u32 mask, u32 dmode, u32 world)
{
    if (dmode == 0) return (u32)(word >> shift) & mask;
    u32 top = (u32)(word >> 32);
    if (dmode == 2) top *= 0x9E3779B1u;
    return (u32)(((u64)top * world) >> 32);
}

// fmode != 0: only the records whose owner (rs_digit with dmode = fmode) is `fme` are counted -- the first pass of a sort
// that reads ALL records of a multi-GPU build and keeps this rank's (radix_sort's SortSource).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void rs_hist(const u64 *__restrict__ w, u64 n, int shift, u32 mask,
                                                    u32 *__restrict__ tile_hist, u32 dmode, u32 world, u32 fmode, u32 fme)
{
    constexpr int RS_THREADS = THREADS, RS_TILE = THREADS * RS_IPT;
    __shared__ u32 h[256];
    const int tid = threadIdx.x;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RS_TILE;
    if (base + RS_TILE <= n && ((size_t)w & 15) == 0) { // whole tile: 16-byte loads, two records each (order is irrelevant here)
        const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w + base);
        ulonglong2 v[RS_IPT / 2];
#pragma unroll
        for (int k = 0; k < RS_IPT / 2; k++) { v[k].x = __builtin_nontemporal_load(&w2[k * RS_THREADS + tid].x); v[k].y = __builtin_nontemporal_load(&w2[k * RS_THREADS + tid].y); }
#pragma unroll
        for (int k = 0; k < RS_IPT / 2; k++) {
            if (!fmode || rs_digit(v[k].x, 0, 0, fmode, world) == fme) atomicAdd(&h[rs_digit(v[k].x, shift, mask, dmode, world)], 1u);
            if (!fmode || rs_digit(v[k].y, 0, 0, fmode, world) == fme) atomicAdd(&h[rs_digit(v[k].y, shift, mask, dmode, world)], 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < RS_IPT; k++) {
            u64 idx = base + (u64)k * RS_THREADS + tid;
            if (idx < n) {
                const u64 v = w[idx];
                if (!fmode || rs_digit(v, 0, 0, fmode, world) == fme) atomicAdd(&h[rs_digit(v, shift, mask, dmode, world)], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 256) tile_hist[(size_t)blockIdx.x * 256 + tid] = h[tid];
}

// The same histogram from the digit bytes the previous pass's scatter left in output order (SortScratch::next_digit).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void rs_hist_bytes(const unsigned char *__restrict__ nd, u64 n, u32 *__restrict__ tile_hist)
{
    constexpr int RS_TILE = THREADS * RS_IPT;
    static_assert(RS_IPT == 16, "one 16-byte load per thread");
    __shared__ u32 h[256];
    const int tid = threadIdx.x;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RS_TILE;
    if (base + RS_TILE <= n && ((size_t)nd & 15) == 0) {
        const uint4 *p = reinterpret_cast<const uint4 *>(nd + base) + tid;
        u32 w[4];
        w[0] = __builtin_nontemporal_load(&p->x); w[1] = __builtin_nontemporal_load(&p->y);
        w[2] = __builtin_nontemporal_load(&p->z); w[3] = __builtin_nontemporal_load(&p->w);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&h[w[k] & 255u], 1u); atomicAdd(&h[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&h[(w[k] >> 16) & 255u], 1u); atomicAdd(&h[w[k] >> 24], 1u);
        }
    } else {
        for (u64 i = base + tid; i < n && i < base + RS_TILE; i += THREADS) atomicAdd(&h[nd[i]], 1u);
    }
    __syncthreads();
    if (tid < 256) tile_hist[(size_t)blockIdx.x * 256 + tid] = h[tid];
}

// Offsets: the run of digit d of tile t starts at  sum of all counts of digits < d  +  counts of d in tiles < t.
//   rs_colsum    per chunk of RS_CHUNK_TILES tiles: count of every digit               [chunks][256]
//   rs_chunkscan one workgroup: per digit, exclusive scan over the chunks + the digit's base
//   rs_colscan   per chunk: walk its tiles, turning counts into offsets
// Loads are issued in batches of independent loads before anything is stored: these kernels are latency-bound.
constexpr int RS_BATCH = 16;

__global__ __launch_bounds__(256) void rs_colsum(const u32 *__restrict__ tile_hist, u32 tiles, u32 *__restrict__ chunk_sums,
                                                  unsigned long long *__restrict__ grand_total /* may be null: + all counts of this chunk */)
{
    const u32 t0 = blockIdx.x * RS_CHUNK_TILES;
    const u32 t1 = min(tiles, t0 + RS_CHUNK_TILES);
    u32 s = 0;
    for (u32 t = t0; t < t1; t += RS_BATCH) {
        u32 v[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) v[i] = (t + i < t1) ? tile_hist[(size_t)(t + i) * 256 + threadIdx.x] : 0;
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) s += v[i];
    }
    chunk_sums[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (grand_total) {                               // (one atomic per chunk of 64 tiles)
        __shared__ u32 ws[4];
        u32 x = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(grand_total, (unsigned long long)ws[0] + ws[1] + ws[2] + ws[3]);
    }
}

// 1024 threads: thread (q, d) scans quarter q of the chunks for digit d; chunk_off[c][d] = digit base + counts in chunks < c
__global__ __launch_bounds__(1024) void rs_chunkscan(const u32 *__restrict__ chunk_sums, u32 *__restrict__ chunk_off, u32 chunks)
{
    __shared__ u32 qsum[4][256];
    __shared__ u32 wsum[4];
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    const u32 per = (chunks + 3) / 4;
    const u32 c0 = min(chunks, q * per), c1 = min(chunks, c0 + per);
    u32 run = 0;
    for (u32 c = c0; c < c1; c += RS_BATCH) {
        u32 v[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) v[i] = (c + i < c1) ? chunk_sums[(size_t)(c + i) * 256 + d] : 0;
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) run += v[i];
    }
    qsum[q][d] = run;
    __syncthreads();
    u32 base = 0;
    if (q == 0) {                                    // (threads 0..255: exactly the shape block_exclusive_scan_256 expects)
        const u32 tot = qsum[0][d] + qsum[1][d] + qsum[2][d] + qsum[3][d];
        const int lane = d & 63, wv = d >> 6;
        u32 x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u32 y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wv] = x;
        base = x - tot;
    }
    __syncthreads();
    if (q == 0) {
        for (int w = 0; w < (d >> 6); w++) base += wsum[w];
        const u32 q0 = qsum[0][d], q1 = qsum[1][d], q2 = qsum[2][d];
        qsum[0][d] = base; qsum[1][d] = base + q0; qsum[2][d] = base + q0 + q1; qsum[3][d] = base + q0 + q1 + q2;
    }
    __syncthreads();
    run = qsum[q][d];
    for (u32 c = c0; c < c1; c += RS_BATCH) {
        u32 v[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) v[i] = (c + i < c1) ? chunk_sums[(size_t)(c + i) * 256 + d] : 0;
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) {
            if (c + i < c1) chunk_off[(size_t)(c + i) * 256 + d] = run;
            run += v[i];
        }
    }
}

__global__ __launch_bounds__(256) void rs_colscan(u32 *tile_hist, u32 tiles, const u32 *__restrict__ chunk_off)
{
    const u32 t0 = blockIdx.x * RS_CHUNK_TILES;
    const u32 t1 = min(tiles, t0 + RS_CHUNK_TILES);
    u32 run = chunk_off[(size_t)blockIdx.x * 256 + threadIdx.x];
    for (u32 t = t0; t < t1; t += RS_BATCH) {
        u32 v[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) v[i] = (t + i < t1) ? tile_hist[(size_t)(t + i) * 256 + threadIdx.x] : 0;
#pragma unroll
        for (int i = 0; i < RS_BATCH; i++) {
            if (t + i < t1) tile_hist[(size_t)(t + i) * 256 + threadIdx.x] = run;
            run += v[i];
        }
    }
}

// The scatter.  Thread (wave w, lane l) owns records base + w*64*IPT + r*64 + l, r = 0..IPT-1,
// so that (wave, round, lane) order is the input order: ranks assigned in that order keep
// the sort stable.
template <bool HAS_B, bool HAS_C, bool DIG_B, int THREADS, bool FILTER>
__global__ __launch_bounds__(THREADS) void rs_scatter(const u64 *__restrict__ a_in, const u64 *__restrict__ b_in,
                                                          const u32 *__restrict__ c_in, u64 *__restrict__ a_out,
                                                          u64 *__restrict__ b_out, u32 *__restrict__ c_out, u64 n,
                                                          int shift, u32 mask, const u32 *__restrict__ tile_off, u32 xg, u32 dmode, u32 world,
                                                          u32 fmode, u32 fme, unsigned char *__restrict__ nd_out /* null, or: the next pass's digit of every
                                                          record, from the digit word, at the record's place */, int nd_shift, u32 nd_mask)
{
    // FILTER: the tile is read whole, but only the records whose owner (rs_digit with dmode = fmode) is `fme` are ranked,
    // staged and written -- the first pass of a sort over ALL records of a multi-GPU build that keeps this rank's
    // (SortSource); tile_off then holds the offsets of the kept records only.
    constexpr int RS_THREADS = THREADS, RS_TILE = THREADS * RS_IPT, RS_WAVES = THREADS / 64;
    __shared__ u64 stage[RS_TILE];          // 8 B per record of the tile
    __shared__ u32 wcnt[RS_WAVES][256];            // per-wave digit counters, then per-wave digit bases
    __shared__ u32 delta[256];              // global offset of the digit's run minus its local start
    __shared__ u32 wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // The 8 XCDs have separate L2s and workgroups are dealt round-robin over them (b % 8).  The output runs of
    // neighbouring tiles meet inside cache lines; when the neighbours run on different XCDs both L2s write their
    // part of such a line back as a partial write (measured: 20-30 % of the scatter bandwidth).  So every XCD
    // gets runs of xg consecutive tiles.  Placement is a speed matter only: any mapping is a permutation.
    u32 tile = blockIdx.x;
    if (xg > 1) {
        const u32 W = 8 * xg, w = tile / W, r = tile % W;
        if ((w + 1) * W <= gridDim.x) tile = w * W + (r % 8) * xg + r / 8;
    }
    const u64 base = (u64)tile * RS_TILE;
    const u32 count = (u32)min((u64)RS_TILE, n - base);
    const u64 *dig_in = DIG_B ? b_in : a_in;
    const u64 *oth_in = DIG_B ? a_in : b_in;
    u64 *dig_out = DIG_B ? b_out : a_out;
    u64 *oth_out = DIG_B ? a_out : b_out;

    u64 kw[RS_IPT];
    const u32 l0 = (u32)wv * 64 * RS_IPT + lane;      // local index of round 0
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        u32 li = l0 + r * 64;
        kw[r] = (li < count) ? __builtin_nontemporal_load(dig_in + base + li) : 0;
    }
    u32 cv[HAS_C ? RS_IPT : 1];             // the payload is fetched now, beside the keys, and staged after them
    if (HAS_C) {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) {
            u32 li = l0 + r * 64;
            cv[r] = (li < count) ? __builtin_nontemporal_load(c_in + base + li) : 0;
        }
    }
    for (int i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();

    u32 lpos[RS_IPT];
    u32 vmask = 0;                          // FILTER: which of this thread's records are kept
    const u64 lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        bool valid = (l0 + r * 64) < count;
        if (FILTER) {
            valid = valid && rs_digit(kw[r], 0, 0, fmode, world) == fme;
            vmask |= valid ? (1u << r) : 0u;
        }
        const u32 d = rs_digit(kw[r], shift, mask, dmode, world);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1;
            const u64 bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const u32 cnt = (u32)__popcll(peers);
        const u32 below = (u32)__popcll(peers & lt_mask);
        const int leader = peers ? (__ffsll((long long)peers) - 1) : lane;
        u32 prev = 0;
        if (valid && lane == leader) { prev = wcnt[wv][d]; wcnt[wv][d] = prev + cnt; }
        prev = __shfl(prev, leader, 64);
        lpos[r] = prev + below;               // rank within (wave, digit) so far
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    u32 staged = count;                     // records of this tile that are written (FILTER: the kept ones)
    {
        // threads 0..255 own one digit each: per-wave counts -> per-wave starts inside the tile
        const int dg = tid & 255;
        u32 c[RS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { c[w] = wcnt[w][dg]; tot += c[w]; }
        u32 kept = 0;
        const u32 start = block_exclusive_scan_256(tid < 256 ? tot : 0u, wsum, FILTER ? &kept : nullptr);
        if (FILTER) staged = kept;
        if (tid < 256) {
            u32 run = start;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { wcnt[w][dg] = run; run += c[w]; }
            delta[tid] = tile_off[(size_t)tile * 256 + tid] - start;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; r++) {
        if (FILTER ? ((vmask >> r) & 1u) != 0 : (l0 + r * 64) < count) {
            const u32 d = rs_digit(kw[r], shift, mask, dmode, world);
            lpos[r] += wcnt[wv][d];
            stage[lpos[r]] = kw[r];
        }
    }
    // The other key word is fetched now, not beside the digit word: two-word records go through 512-thread tiles, where 32
    // more registers held from the start would leave one workgroup per CU (30.3 against 27.1 ms per pass at 2.9e9 records).
    // Here the key registers are free again, and the loads are in flight while the first word's runs are written (fetched
    // only after that, every tile waited a whole memory latency there).
    u64 lv[HAS_B ? RS_IPT : 1];
    if (HAS_B) {
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) {
            u32 li = l0 + r * 64;
            lv[r] = (li < count) ? __builtin_nontemporal_load(oth_in + base + li) : 0;
        }
    }
    __syncthreads();
    u32 dst[RS_IPT];
#pragma unroll
    for (int k = 0; k < RS_IPT; k++) {
        const u32 i = k * RS_THREADS + tid;
        if (i < staged) {
            const u64 v = stage[i];
            dst[k] = delta[rs_digit(v, shift, mask, dmode, world)] + i;
            dig_out[dst[k]] = v;
            if (nd_out) nd_out[dst[k]] = (unsigned char)((u32)(v >> nd_shift) & nd_mask);
        }
    }
    if (HAS_B) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) {
            u32 li = l0 + r * 64;
            if (FILTER ? ((vmask >> r) & 1u) != 0 : li < count) stage[lpos[r]] = lv[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS_IPT; k++) {
            const u32 i = k * RS_THREADS + tid;
            if (i < staged) oth_out[dst[k]] = stage[i];
        }
    }
    if (HAS_C) {
        u32 *stage32 = reinterpret_cast<u32 *>(stage);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_IPT; r++) {
            u32 li = l0 + r * 64;
            if (FILTER ? ((vmask >> r) & 1u) != 0 : li < count) stage32[lpos[r]] = cv[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS_IPT; k++) {
            const u32 i = k * RS_THREADS + tid;
            if (i < staged) c_out[dst[k]] = stage32[i];
        }
    }
}

u32 radix_sort_tile_records(bool has_b, bool has_c) { return (has_b || has_c) ? 512 * RS_IPT : 256 * RS_IPT; }

int SortScratch::reserve(u64 n)
{
    size_t tiles = (size_t)((n + RS_TILE_MIN - 1) / RS_TILE_MIN);
    size_t chunks = (tiles + RS_CHUNK_TILES - 1) / RS_CHUNK_TILES;
    if (external && (tiles > cap_tiles || chunks > cap_chunks)) return (int)hipErrorInvalidValue;
    if (tiles > cap_tiles) {
        if (tile_hist) (void)hipFree(tile_hist);
        tile_hist = nullptr; cap_tiles = 0;
        RS_CHECK(hipMalloc(&tile_hist, tiles * 256 * sizeof(u32)));
        cap_tiles = tiles;
    }
    if (chunks > cap_chunks) {
        if (chunk_sums) (void)hipFree(chunk_sums);
        chunk_sums = nullptr; cap_chunks = 0;
        RS_CHECK(hipMalloc(&chunk_sums, (2 * chunks * 256 + 2) * sizeof(u32)));     // sums, then offsets, then one 8-byte total (radix_sort_count_kept)
        cap_chunks = chunks;
    }
    if (!xcd_group) {
        // 64 consecutive tiles per XCD: the ~512 workgroups in flight then form ONE run of neighbours per XCD.  Against 16: the
        // same at 6e7 records, +4-6 % at 1e9 and 2.6e9 (tools/sortbench <n> <reps> sorts under RS_XCD_GROUP)
        xcd_group = 64;
        if (getenv("RS_XCD_GROUP")) xcd_group = (u32)atoi(getenv("RS_XCD_GROUP"));
        if (!xcd_group) xcd_group = 1;
    }
    return 0;
}

int SortScratch::reserve_next_digit(u64 n)
{
    if (external) return (int)hipErrorInvalidValue;
    if (lease_next_digit && n <= cap_lease_next_digit) { next_digit = lease_next_digit; return 0; }
    if (n > RS_OWN_NEXT_DIGIT_MAX) return (int)hipErrorOutOfMemory;          // (no lease for a sort this large: key-word histograms)
    if (n > cap_own_next_digit) {
        if (own_next_digit) (void)hipFree(own_next_digit);
        own_next_digit = nullptr; cap_own_next_digit = 0;
        const size_t want = (size_t)n + (size_t)(n / 8) + 4096;
        RS_CHECK(hipMalloc(&own_next_digit, want));
        cap_own_next_digit = want;
    }
    next_digit = own_next_digit;
    return 0;
}

void SortScratch::bind_external(u32 *hist, size_t tiles, u32 *chunks, size_t chunk_count)
{
    release();
    external = true;
    tile_hist = hist; cap_tiles = tiles;
    chunk_sums = chunks; cap_chunks = chunk_count;
}

void SortScratch::release()
{
    if (tile_hist && !external) (void)hipFree(tile_hist);
    if (chunk_sums && !external) (void)hipFree(chunk_sums);
    if (own_next_digit) (void)hipFree(own_next_digit);
    own_next_digit = next_digit = nullptr; cap_own_next_digit = 0;
    external = false;
    if (ev[0]) for (int i = 0; i < 2 * RS_MAX_PASSES; i++) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
    tile_hist = nullptr; chunk_sums = nullptr; cap_tiles = cap_chunks = 0;
}

struct NextDigit { unsigned char *out = nullptr; int shift = 0; u32 mask = 0; };
template <bool HAS_B, bool HAS_C>
static void launch_scatter(hipStream_t s, bool dig_b, u32 tiles, const SortArrays &in, const SortArrays &out, u64 n,
                           int shift, u32 mask, const u32 *tile_off, u32 xg, u32 dmode, u32 world, u32 fmode, u32 fme, const NextDigit &nd)
{
    constexpr int T = (HAS_C || HAS_B) ? 512 : 256;
    if (fmode)                              // (the filtered first pass of a sort: the digit word is a)
        hipLaunchKernelGGL((rs_scatter<HAS_B, HAS_C, false, T, true>), dim3(tiles), dim3(T), 0, s, in.a, in.b, in.c, out.a, out.b,
                           out.c, n, shift, mask, tile_off, xg, dmode, world, fmode, fme, nd.out, nd.shift, nd.mask);
    else if (HAS_B && dig_b)
        hipLaunchKernelGGL((rs_scatter<HAS_B, HAS_C, true, T, false>), dim3(tiles), dim3(T), 0, s, in.a, in.b, in.c, out.a, out.b,
                           out.c, n, shift, mask, tile_off, xg, dmode, world, 0u, 0u, nd.out, nd.shift, nd.mask);
    else
        hipLaunchKernelGGL((rs_scatter<HAS_B, HAS_C, false, T, false>), dim3(tiles), dim3(T), 0, s, in.a, in.b, in.c, out.a, out.b,
                           out.c, n, shift, mask, tile_off, xg, dmode, world, 0u, 0u, nd.out, nd.shift, nd.mask);
}

static void launch_scatter_any(hipStream_t s, bool has_b, bool has_c, bool dig_b, u32 tiles, const SortArrays &in, const SortArrays &out, u64 n,
                               int shift, u32 mask, const u32 *tile_off, u32 xg, u32 dmode = 0, u32 world = 0, u32 fmode = 0, u32 fme = 0,
                               const NextDigit &nd = NextDigit())
{
    if (has_b) {
        if (has_c) launch_scatter<true, true>(s, dig_b, tiles, in, out, n, shift, mask, tile_off, xg, dmode, world, fmode, fme, nd);
        else launch_scatter<true, false>(s, dig_b, tiles, in, out, n, shift, mask, tile_off, xg, dmode, world, fmode, fme, nd);
    } else {
        if (has_c) launch_scatter<false, true>(s, false, tiles, in, out, n, shift, mask, tile_off, xg, dmode, world, fmode, fme, nd);
        else launch_scatter<false, false>(s, false, tiles, in, out, n, shift, mask, tile_off, xg, dmode, world, fmode, fme, nd);
    }
}

static void launch_hist(hipStream_t s, bool wide, u32 tiles, const u64 *word, u64 n, int shift, u32 mask, u32 *th, u32 dmode = 0, u32 world = 0,
                        u32 fmode = 0, u32 fme = 0)
{
    if (wide) hipLaunchKernelGGL(rs_hist<512>, dim3(tiles), dim3(512), 0, s, word, n, shift, mask, th, dmode, world, fmode, fme);
    else hipLaunchKernelGGL(rs_hist<256>, dim3(tiles), dim3(256), 0, s, word, n, shift, mask, th, dmode, world, fmode, fme);
}

static void launch_hist_bytes(hipStream_t s, bool wide, u32 tiles, const unsigned char *nd, u64 n, u32 *th)
{
    if (wide) hipLaunchKernelGGL(rs_hist_bytes<512>, dim3(tiles), dim3(512), 0, s, nd, n, th);
    else hipLaunchKernelGGL(rs_hist_bytes<256>, dim3(tiles), dim3(256), 0, s, nd, n, th);
}

static void launch_offsets(hipStream_t s, u32 *th, u32 tiles, SortScratch &scratch, unsigned long long *grand_total = nullptr)
{
    const u32 chunks = (tiles + RS_CHUNK_TILES - 1) / RS_CHUNK_TILES;
    hipLaunchKernelGGL(rs_colsum, dim3(chunks), dim3(256), 0, s, th, tiles, scratch.chunk_sums, grand_total);
    u32 *chunk_off = scratch.chunk_sums + scratch.cap_chunks * 256;
    hipLaunchKernelGGL(rs_chunkscan, dim3(1), dim3(1024), 0, s, scratch.chunk_sums, chunk_off, chunks);
    hipLaunchKernelGGL(rs_colscan, dim3(chunks), dim3(256), 0, s, th, tiles, chunk_off);
}

// The first half of a sort whose first pass reads ALL records of a multi-GPU build and keeps this rank's (SortSource):
// histogram of the first digit over the kept records + offsets, left in scratch.tile_hist; *n_kept = how many records the
// sort will hold (the caller sizes keys / tmp from it, then calls radix_sort with the same bit range and `from`).
int radix_sort_count_kept(hipStream_t s, const SortSource &from, int bit_lo, int bit_hi, SortScratch &scratch, u64 *n_kept)
{
    *n_kept = 0;
    if (from.n == 0) return 0;
    if (from.n >= (1ull << 32) || (from.fmode != 1 && from.fmode != 2) || from.world < 1 || from.world > 256 || from.me >= from.world || bit_hi <= bit_lo)
        return (int)hipErrorInvalidValue;
    bit_lo &= ~7;
    int rc = scratch.reserve(from.n);
    if (rc) return rc;
    const bool has_b = from.src.b != nullptr, has_c = from.src.c != nullptr;
    if (has_b && bit_lo < 64) return (int)hipErrorInvalidValue;         // the owner is read from word a: it must be the digit word too
    const u32 tile_recs = radix_sort_tile_records(has_b, has_c);
    const u32 tiles = (u32)((from.n + tile_recs - 1) / tile_recs);
    const int nbits = (bit_hi - bit_lo) < 8 ? (bit_hi - bit_lo) : 8;
    const int shift = has_b ? (bit_lo & 63) : bit_lo;
    if (scratch.external) return (int)hipErrorInvalidValue;             // (a caller-provided scratch has no room for the total)
    unsigned long long *total_dev = reinterpret_cast<unsigned long long *>(scratch.chunk_sums + 2 * scratch.cap_chunks * 256);
    RS_CHECK(hipMemsetAsync(total_dev, 0, sizeof(unsigned long long), s));
    launch_hist(s, has_b || has_c, tiles, from.src.a, from.n, shift, (1u << nbits) - 1u, scratch.tile_hist, 0u, from.world, (u32)from.fmode, from.me);
    launch_offsets(s, scratch.tile_hist, tiles, scratch, total_dev);
    unsigned long long h = 0;
    RS_CHECK(hipMemcpyAsync(&h, total_dev, sizeof h, hipMemcpyDeviceToHost, s));
    RS_CHECK(hipStreamSynchronize(s));
    *n_kept = h;
    RS_CHECK(hipGetLastError());
    return 0;
}

bool radix_sort_next_digit_wanted(const SortScratch &scratch, u64 n)
{
    static const bool nd_env = !(getenv("RS_NEXT_DIGIT") && atoi(getenv("RS_NEXT_DIGIT")) == 0);
    return nd_env && !scratch.external && n >= (1u << 16);
}

int radix_sort(hipStream_t s, SortArrays &keys, SortArrays &tmp, u64 n, int bit_lo, int bit_hi, SortScratch &scratch,
               SortTiming *timing, u32 *first_hist, const SortSource *from, bool first_digits_ready)
{
    if (n == 0 || bit_hi <= bit_lo) return 0;
    if (n >= (1ull << 32)) return (int)hipErrorInvalidValue;      // u32 offsets inside the kernels
    bit_lo &= ~7;
    if ((bit_hi - bit_lo + 7) / 8 > RS_MAX_PASSES) return (int)hipErrorInvalidValue;
    if (from && (first_hist || from->n < n)) return (int)hipErrorInvalidValue;
    const bool timed = timing && timing->enabled;
    int pass = 0;
    int rc = scratch.reserve(from ? from->n : n);                  // (never shrinks: what radix_sort_count_kept left is still there)
    if (rc) return rc;
    if (timed && !scratch.ev[0]) for (int i = 0; i < 2 * RS_MAX_PASSES; i++) RS_CHECK(hipEventCreate(&scratch.ev[i]));
    // records with a 4-byte payload or a second key word go through 512-thread tiles of 8192 (their output runs are twice as
    // long: +9 % scatter bandwidth for 12-byte records, 27.1 -> 24.7 ms per pass for two-word records at 2.6e9;
    // tools/spreadwrite.cpp shows what short unaligned runs cost: 3.8 TB/s at 136 B, 4.6 at 264 B), key-only records through
    // 256-thread tiles of 4096.  The second key word is fetched late inside rs_scatter (see there) so that the 512-thread
    // workgroups still fit two per CU.  1024-thread tiles of 16384 (one workgroup per CU, runs twice as long again) were
    // slower: 21.9 ms per pass against 20.4 at 2.6e9 two-word records
    const bool has_b = keys.b != nullptr, has_c = keys.c != nullptr;
    const u32 tile_recs = radix_sort_tile_records(has_b, has_c);
    const u32 tiles = (u32)((n + tile_recs - 1) / tile_recs);
    const u64 rec_bytes = 8 + (has_b ? 8 : 0) + (has_c ? 4 : 0);
    u64 moved[RS_MAX_PASSES];
    // The second word climbs.  Measured at 2.6e9 two-word records, wherever the arrays lie (tools/sortbench <n> 1 gaps; RS_TRACE
    // on the build): a pass that writes the second word ABOVE the array it reads it from takes 17.6-19.5 ms, one that writes it
    // below 19.6-21.9 ms; where the first word goes makes no difference.  A ping-pong has one pass of each kind in turn.  With
    // keys.b the first of a chain of arrays every pass writes into the next one up, and only the step back to the first goes down.
    // The next digit travels ahead.  A histogram pass reads the key word of every record (8 bytes, 4.3 ms at 2.9e9 records) for
    // one byte of it.  The scatter before it has that word in its hands: it writes the NEXT pass's digit of every record, one byte, at the
    // record's new place (runs of bytes as long as the runs of records), and the next histogram reads those bytes.  Only between
    // passes whose digits come from the same word, for sorts large enough to care, and not with a caller's scratch (no room).
    const bool nd_ok = radix_sort_next_digit_wanted(scratch, n) && scratch.reserve_next_digit(n) == 0;
    bool have_nd = nd_ok && first_digits_ready && !from && !first_hist;      // scratch.next_digit holds this pass's digits
    u64 *const chain = (has_b && keys.b_chain_n > 1) ? keys.b : nullptr;
    const int chain_n = chain ? keys.b_chain_n : 0;
    const size_t chain_stride = keys.b_stride;
    int chain_at = 0;
    for (int bit = bit_lo; bit < bit_hi; bit += 8, pass++) {
        const int nbits = (bit_hi - bit) < 8 ? (bit_hi - bit) : 8;
        const u32 mask = (1u << nbits) - 1u;
        const bool dig_b = has_b && bit < 64;
        const int shift = has_b ? (bit & 63) : bit;
        const u64 *word = dig_b ? keys.b : keys.a;
        u32 *th = scratch.tile_hist;
        NextDigit nd;
        if (nd_ok && bit + 8 < bit_hi && (has_b && bit + 8 < 64) == dig_b) {
            const int nb2 = (bit_hi - bit - 8) < 8 ? (bit_hi - bit - 8) : 8;
            nd.out = scratch.next_digit; nd.shift = has_b ? ((bit + 8) & 63) : bit + 8; nd.mask = (1u << nb2) - 1u;
        }
        if (pass == 0 && from) {
            // the kept records' offsets are in scratch.tile_hist already (radix_sort_count_kept, same bit range); the pass reads
            // the whole source and writes this rank's records, so what follows works on n records
            const u32 tiles0 = (u32)((from->n + tile_recs - 1) / tile_recs);
            if (chain) { chain_at = (chain_at + 1) % chain_n; tmp.b = chain + (size_t)chain_at * chain_stride; }      // (keys.b, the chain's first array, is not read by this pass)
            if (timed) RS_CHECK(hipEventRecord(scratch.ev[0], s));
            launch_scatter_any(s, has_b, has_c, false, tiles0, from->src, tmp, from->n, shift, mask, th, scratch.xcd_group, 0u, from->world, (u32)from->fmode, from->me, nd);
            if (timed) RS_CHECK(hipEventRecord(scratch.ev[1], s));
            have_nd = nd.out != nullptr;
            moved[0] = rec_bytes * (from->n + n) + (have_nd ? n : 0);
            SortArrays t0 = keys; keys = tmp; tmp = t0;
            continue;
        }
        if (pass == 0 && first_hist) th = first_hist;                    // the producer already counted this digit (tiles of radix_sort_tile_records())
        else if (have_nd) launch_hist_bytes(s, has_c || has_b, tiles, scratch.next_digit, n, th);
        else launch_hist(s, has_c || has_b, tiles, word, n, shift, mask, th);
        launch_offsets(s, th, tiles, scratch);
        if (chain) { chain_at = (chain_at + 1) % chain_n; tmp.b = chain + (size_t)chain_at * chain_stride; }
        if (timed) RS_CHECK(hipEventRecord(scratch.ev[2 * pass], s));
        launch_scatter_any(s, has_b, has_c, dig_b, tiles, keys, tmp, n, shift, mask, th, scratch.xcd_group, 0u, 0u, 0u, 0u, nd);
        if (timed) RS_CHECK(hipEventRecord(scratch.ev[2 * pass + 1], s));
        have_nd = nd.out != nullptr;
        moved[pass] = 2 * rec_bytes * n + (have_nd ? n : 0);          // (the digit bytes are part of what the pass writes)
        SortArrays t = keys; keys = tmp; tmp = t;
    }
    if (timed) {                                                   // read the events once, after the last pass
        RS_CHECK(hipEventSynchronize(scratch.ev[2 * pass - 1]));
        static const bool trace = getenv("RS_TRACE") != nullptr;      // per-pass times and where the arrays lie
        for (int q = 0; q < pass; q++) {
            float ms = 0.f;
            RS_CHECK(hipEventElapsedTime(&ms, scratch.ev[2 * q], scratch.ev[2 * q + 1]));
            if (trace) fprintf(stderr, "[rs] pass %d: scatter %.3f ms = %.0f GB/s  (result of the sort: a %p b %p c %p, spare: a %p b %p c %p)\n", q, ms,
                               moved[q] / ms / 1e6, (void *)keys.a, (void *)keys.b, (void *)keys.c, (void *)tmp.a, (void *)tmp.b, (void *)tmp.c);
            timing->scatter_ms += ms;
            timing->scatter_launches += 1;
            timing->scatter_bytes += moved[q];
        }
    }
    RS_CHECK(hipGetLastError());
    return 0;
}

// One stable pass that groups the records by the rank that owns them (rs_digit, dmode 1 or 2): the multi-GPU flow's
// partition before an exchange.  Every array of the record moves in this one pass; the digit comes from keys.a.
// counts[r] = records of rank r.  The result is in `keys` (the arrays are swapped, as in radix_sort).
int radix_partition(hipStream_t s, SortArrays &keys, SortArrays &tmp, u64 n, int dmode, u32 world, SortScratch &scratch, SortTiming *timing,
                    uint64_t *counts)
{
    for (u32 r = 0; r < world; r++) counts[r] = 0;
    if (n == 0) return 0;
    if (n >= (1ull << 32) || world < 1 || world > 256 || (dmode != 1 && dmode != 2)) return (int)hipErrorInvalidValue;
    const bool timed = timing && timing->enabled;
    int rc = scratch.reserve(n);
    if (rc) return rc;
    if (timed && !scratch.ev[0]) for (int i = 0; i < 2 * RS_MAX_PASSES; i++) RS_CHECK(hipEventCreate(&scratch.ev[i]));
    const bool has_b = keys.b != nullptr, has_c = keys.c != nullptr;
    const u32 tile_recs = radix_sort_tile_records(has_b, has_c);
    const u32 tiles = (u32)((n + tile_recs - 1) / tile_recs);
    u32 *th = scratch.tile_hist;
    launch_hist(s, has_c || has_b, tiles, keys.a, n, 0, 255u, th, (u32)dmode, world);
    launch_offsets(s, th, tiles, scratch);
    u32 starts[256];                                                 // row of tile 0: where each rank's records begin
    RS_CHECK(hipMemcpyAsync(starts, th, world * sizeof(u32), hipMemcpyDeviceToHost, s));
    if (timed) RS_CHECK(hipEventRecord(scratch.ev[0], s));
    launch_scatter_any(s, has_b, has_c, false, tiles, keys, tmp, n, 0, 255u, th, scratch.xcd_group, (u32)dmode, world);
    if (timed) RS_CHECK(hipEventRecord(scratch.ev[1], s));
    RS_CHECK(hipStreamSynchronize(s));
    if (timed) {
        float ms = 0.f;
        RS_CHECK(hipEventElapsedTime(&ms, scratch.ev[0], scratch.ev[1]));
        timing->scatter_ms += ms;
        timing->scatter_launches += 1;
        timing->scatter_bytes += 2 * (8 + (has_b ? 8 : 0) + (has_c ? 4 : 0)) * n;
    }
    for (u32 r = 0; r < world; r++) counts[r] = (r + 1 < world ? starts[r + 1] : (u64)n) - starts[r];
    SortArrays t = keys; keys = tmp; tmp = t;
    RS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// exclusive scan u32 -> u64, chunks of 4096 per workgroup, recursive over chunk sums
// ------------------------------------------------------------------------------------
constexpr int SC_IPT = 16;
constexpr int SC_CHUNK = 256 * SC_IPT;

__device__ __forceinline__ u64 block_exclusive_scan_256_u64(u64 v, u64 *lds_waves, u64 *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u64 y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds_waves[wv] = x;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { u64 s = lds_waves[w]; if (w < wv) base += s; tot += s; }
    if (total) *total = tot;
    __syncthreads();
    return base + x - v;
}

template <typename TIN>
__global__ __launch_bounds__(256) void sc_reduce(const TIN *__restrict__ in, u64 n, u64 *__restrict__ sums)
{
    __shared__ u64 ws[4];
    const u64 base = (u64)blockIdx.x * SC_CHUNK + (u64)threadIdx.x * SC_IPT;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SC_IPT; k++) if (base + k < n) s += in[base + k];
    u64 tot;
    (void)block_exclusive_scan_256_u64(s, ws, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename TIN>
__global__ __launch_bounds__(256) void sc_scan(const TIN *in, u64 *out, u64 n,
                                               const u64 *__restrict__ chunk_off, u64 *__restrict__ total)
{
    __shared__ u64 ws[4];
    const u64 base = (u64)blockIdx.x * SC_CHUNK + (u64)threadIdx.x * SC_IPT;
    u64 v[SC_IPT];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SC_IPT; k++) { v[k] = (base + k < n) ? (u64)in[base + k] : 0; s += v[k]; }
    u64 tot;
    u64 pre = block_exclusive_scan_256_u64(s, ws, &tot) + (chunk_off ? chunk_off[blockIdx.x] : 0);
#pragma unroll
    for (int k = 0; k < SC_IPT; k++) { if (base + k < n) out[base + k] = pre; pre += v[k]; }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total = pre;
}

size_t scan_scratch_elems(u64 n)
{
    size_t tot = 0;
    u64 m = n;
    while (m > 1) { m = (m + SC_CHUNK - 1) / SC_CHUNK; tot += (size_t)m + 1; if (m == 1) break; }
    return tot + 8;
}

static int scan_u64_inplace(hipStream_t s, u64 *data, u64 n, u64 *scratch);

template <typename TIN>
static int scan_impl(hipStream_t s, const TIN *in, u64 *out, u64 n, u64 *scratch, u64 *total_dev)
{
    if (n == 0) { if (total_dev) RS_CHECK(hipMemsetAsync(total_dev, 0, sizeof(u64), s)); return 0; }
    const u64 chunks = (n + SC_CHUNK - 1) / SC_CHUNK;
    if (chunks == 1) {
        hipLaunchKernelGGL((sc_scan<TIN>), dim3(1), dim3(256), 0, s, in, out, n, (const u64 *)nullptr, total_dev);
        return 0;
    }
    u64 *sums = scratch;
    hipLaunchKernelGGL((sc_reduce<TIN>), dim3((u32)chunks), dim3(256), 0, s, in, n, sums);
    int rc = scan_u64_inplace(s, sums, chunks, scratch + chunks);
    if (rc) return rc;
    hipLaunchKernelGGL((sc_scan<TIN>), dim3((u32)chunks), dim3(256), 0, s, in, out, n, (const u64 *)sums, total_dev);
    return 0;
}

static int scan_u64_inplace(hipStream_t s, u64 *data, u64 n, u64 *scratch)
{
    return scan_impl<u64>(s, data, data, n, scratch, nullptr);
}

int exclusive_scan_u32_to_u64(hipStream_t s, const u32 *in, u64 *out, u64 n, u64 *scratch_dev, u64 *total_dev)
{
    int rc = scan_impl<u32>(s, in, out, n, scratch_dev, total_dev);
    if (rc) return rc;
    RS_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cora
