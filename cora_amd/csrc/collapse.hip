// collapse.hip -- CORA's read-side read-mer collapsing on gfx950.
// CL = /root/reference/collapse.cpp: whole-input run mode (argv[13] = NONE, the form cora.cpp:1674-1692 issues) and
// fragmented mode (argv[13] = one group's signals, input = a fragment file of fastqSplit, cora.cpp:1733-1760).
//
// What the reference does with one hash map (CL:640-643) is done here with sorts:
//   * every N-free reference window of R bases (CL:656-711) and every read-mer of the input (reads, mates, splits in
//     file order, CL:923-2171) becomes a record (FNV-1a 64 of its canonical form, id); ids order the records the way
//     the reference inserts them: reference positions first (id = global position), then read-mers (id = G + index);
//   * a stable radix sort on the hash brings equal read-mers together in insertion order; neighbours with equal hashes
//     are compared base by base (different read-mers with one hash are re-ordered by a repair kernel);
//   * a group's first record is the one the reference would have inserted: it decides "in the reference or not",
//     position and stored direction (CL:686-695, 948-972); a read-mer's id digits are 2 * index + direction flag;
//   * the reference prints in the iteration order of its tr1::unordered_map (CL:2305): bucket = hash mod B ascending,
//     later insertions first inside a bucket.  A second sort on (bucket, ~id of the first record) gives that order;
//   * three kernels size, place (scans) and write the text of the three output files.
// The hash (tr1::hash<string> = FNV-1a 64) and B come from libstdc++ itself on the host (collapse_host.cpp).
// Integer / byte work; the heavy parts are the sequential FNV over R bases per record (ALU) and the sorts (HBM).
#include "job_internal.hpp"
#include <algorithm>
#include <string>
#include <unordered_set>
#include "reads_common.hpp"

using namespace cora;

namespace {

constexpr u64 FNV_BASIS = 14695981039346656037ull, FNV_PRIME = 1099511628211ull;
constexpr u32 ACGT_PACKED = 0x54474341u;       // 'A' 'C' 'G' 'T' by 2-bit code

struct Src {               // where read-mers live: both sides 2-bit packed, 32 bases per u64, base i at bits 62 - 2 (i % 32)
    const u64 *pk;         // the reference (kernels.hip: k_pack); windows that hold anything but A C G T never become records
    u64 pk_words;
    const u64 *rpk;        // the reads, sequence after sequence (k_reads_pack): read-mer idx starts at base idx * R
    const u32 *rbad;       // one bit per read base, MSB first: the base is an N (its code in rpk is 0)
    u64 rpk_words;
    const u64 *codes;      // fragmented mode: the number fastqSplit gave each read-mer (2 x its index in the whole input); else null
    u64 *rsw;              // one bit per read-mer of the input: its canonical form is the reverse complement (written by k_read_keys)
    u64 G;                 // ids below G are reference positions; id - G = running read-mer index
    u32 R, splits;
};

struct CollapseCounters {
    unsigned long long bad_read_bytes, n_mixed, n_owners, n_lrn, name_overflow;
};

// ------------------------------------------------------------------ a read-mer as words
// The canonical form (CL:97-122: the reverse complement when it is the lexicographically smaller string, ASCII order
// A < C < G < N < T) of a read-mer of up to 32 W bases: codes left-aligned in c[], N positions in m[] (code 0 there).
template <int W>
struct Mer { u64 c[W]; u32 m[W]; u32 sw; };

__device__ __forceinline__ u64 rev2(u64 x)     // order of the 32 two-bit groups reversed
{
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

__device__ __forceinline__ u64 spread2(u32 m)  // bit i -> bits 2i and 2i + 1
{
    u64 x = m;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}

__device__ __forceinline__ u64 ldw(const u64 *p, long long i, u64 n) { return (i >= 0 && (u64)i < n) ? p[i] : 0ull; }
__device__ __forceinline__ u32 ldm(const u32 *p, long long i, u64 n) { return (i >= 0 && (u64)i < n) ? p[i] : 0u; }

// The forward words come from a funnel shift that puts the first base at the top of word 0; the reverse complement from
// a second one that puts the LAST base at the bottom of word W - 1, so that reversing it is a fixed permutation of words.
template <int W>
__device__ __forceinline__ void canon(const Src &s, u32 id, Mer<W> &o)
{
    const bool is_ref = id < s.G;
    const u64 base = is_ref ? (u64)id : ((u64)id - s.G) * s.R;
    const u64 *p = is_ref ? s.pk : s.rpk;
    const u64 nw = is_ref ? s.pk_words : s.rpk_words;
    const u32 R = s.R;
    const long long lw0 = (long long)(base >> 5);
    const u32 lo = (u32)(base & 31);
    const long long s0 = (long long)(base + R) - 32 * W;
    const long long rw0 = s0 >> 5;
    const u32 ro = (u32)(s0 & 31);
    u64 fw[W], rc[W];
    {
        u64 prev = ldw(p, lw0, nw);
#pragma unroll
        for (int k = 0; k < W; k++) {
            const u64 next = (32u * (k + 1) < lo + R) ? ldw(p, lw0 + k + 1, nw) : 0ull;
            fw[k] = lo ? (prev << (2 * lo)) | (next >> (64 - 2 * lo)) : prev;
            prev = next;
        }
    }
    {
        u64 prev = ((rw0 + 1) * 32 > (long long)base) ? ldw(p, rw0, nw) : 0ull;
#pragma unroll
        for (int k = 0; k < W; k++) {
            const u64 next = ((rw0 + k + 2) * 32 > (long long)base) ? ldw(p, rw0 + k + 1, nw) : 0ull;
            const u64 x = ro ? (prev << (2 * ro)) | (next >> (64 - 2 * ro)) : prev;
            rc[W - 1 - k] = ~rev2(x);
            prev = next;
        }
    }
    u32 ml[W], mr[W];
    u32 any_n = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
        const u32 nb = R > 32u * k ? min(32u, R - 32u * k) : 0u;
        const u64 mask = nb == 32 ? ~0ull : nb ? ~0ull << (64 - 2 * nb) : 0ull;
        fw[k] &= mask; rc[k] &= mask;
        ml[k] = mr[k] = 0;
    }
    if (!is_ref) {
        const u32 *b = s.rbad;
        u32 prev = ldm(b, lw0, nw);
#pragma unroll
        for (int k = 0; k < W; k++) {
            const u32 next = (32u * (k + 1) < lo + R) ? ldm(b, lw0 + k + 1, nw) : 0u;
            const u32 nb = R > 32u * k ? min(32u, R - 32u * k) : 0u;
            const u32 mask = nb == 32 ? ~0u : nb ? ~0u << (32 - nb) : 0u;
            ml[k] = (lo ? (prev << lo) | (next >> (32 - lo)) : prev) & mask;
            any_n |= ml[k];
            prev = next;
        }
        if (any_n) {
            prev = ((rw0 + 1) * 32 > (long long)base) ? ldm(b, rw0, nw) : 0u;
#pragma unroll
            for (int k = 0; k < W; k++) {
                const u32 next = ((rw0 + k + 2) * 32 > (long long)base) ? ldm(b, rw0 + k + 1, nw) : 0u;
                mr[W - 1 - k] = __brev(ro ? (prev << ro) | (next >> (32 - ro)) : prev);
                prev = next;
            }
#pragma unroll
            for (int k = 0; k < W; k++) {
                const u32 nb = R > 32u * k ? min(32u, R - 32u * k) : 0u;
                mr[k] &= nb == 32 ? ~0u : nb ? ~0u << (32 - nb) : 0u;
                rc[k] &= ~spread2(mr[k]);
            }
        }
    }
    u32 sw = 0;
    bool decided = false;
    if (!any_n) {
#pragma unroll
        for (int k = 0; k < W; k++)
            if (!decided && fw[k] != rc[k]) { decided = true; sw = fw[k] > rc[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < W; k++) {
            const u64 D = (fw[k] ^ rc[k]) | spread2(ml[k] ^ mr[k]);
            if (!decided && D) {
                decided = true;
                const u32 q = (u32)__clzll((long long)D) >> 1;
                const u32 a = (u32)(fw[k] >> (62 - 2 * q)) & 3u, bb = (u32)(rc[k] >> (62 - 2 * q)) & 3u;
                const u32 ra = ((ml[k] >> (31 - q)) & 1u) ? 3u : (a == 3 ? 4u : a);
                const u32 rb = ((mr[k] >> (31 - q)) & 1u) ? 3u : (bb == 3 ? 4u : bb);
                sw = ra > rb;
            }
        }
    }
    o.sw = sw;
#pragma unroll
    for (int k = 0; k < W; k++) { o.c[k] = sw ? rc[k] : fw[k]; o.m[k] = sw ? mr[k] : ml[k]; }
}

// tr1::hash<string> of the canonical read-mer: FNV-1a 64 over its R characters (sequential by definition: ALU work)
template <int W>
__device__ __forceinline__ u64 fnv_of(const Mer<W> &m, u32 R)
{
    u64 h = FNV_BASIS;
#pragma unroll
    for (int k = 0; k < W; k++) {
        const u32 n = R > 32u * k ? min(32u, R - 32u * k) : 0u;
        u64 c = m.c[k];
        u32 mm = m.m[k];
        for (u32 j = 0; j < n; j++) {
            u32 ch = (ACGT_PACKED >> (8 * (u32)(c >> 62))) & 0xFFu;
            if (mm & 0x80000000u) ch = 'N';
            h = (h ^ ch) * FNV_PRIME;
            c <<= 2; mm <<= 1;
        }
    }
    return h;
}

// the same for a read-mer known to hold no N (every reference window: k_win_count drops windows with one): three
// instructions of seventeen fewer per base
template <int W>
__device__ __forceinline__ u64 fnv_of_clean(const Mer<W> &m, u32 R)
{
    u64 h = FNV_BASIS;
#pragma unroll
    for (int k = 0; k < W; k++) {
        const u32 n = R > 32u * k ? min(32u, R - 32u * k) : 0u;
        u64 c = m.c[k];
        for (u32 j = 0; j < n; j++) {
            const u32 ch = (ACGT_PACKED >> (8 * (u32)(c >> 62))) & 0xFFu;
            h = (h ^ ch) * FNV_PRIME;
            c <<= 2;
        }
    }
    return h;
}

template <int W, class P>
__device__ __forceinline__ void put_mer(P dst, const Mer<W> &m, u32 R)
{
    u32 i = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
        const u32 n = R > 32u * k ? min(32u, R - 32u * k) : 0u;
        u64 c = m.c[k];
        u32 mm = m.m[k];
        for (u32 j = 0; j < n; j++) {
            u32 ch = (ACGT_PACKED >> (8 * (u32)(c >> 62))) & 0xFFu;
            if (mm & 0x80000000u) ch = 'N';
            dst[i++] = (u8)ch;
            c <<= 2; mm <<= 1;
        }
    }
}

// <0, 0, >0: a total order on canonical read-mers (any one will do: it only has to bring equal ones together)
template <int W>
__device__ __forceinline__ int mer_cmp(const Mer<W> &a, const Mer<W> &b)
{
    int r = 0;
#pragma unroll
    for (int k = 0; k < W; k++) {
        if (r == 0 && a.c[k] != b.c[k]) r = a.c[k] < b.c[k] ? -1 : 1;
        if (r == 0 && a.m[k] != b.m[k]) r = a.m[k] < b.m[k] ? -1 : 1;
    }
    return r;
}

template <int W>
__device__ __forceinline__ int seq_cmp(const Src &s, u32 ida, u32 idb)
{
    Mer<W> a, b;
    canon<W>(s, ida, a);
    canon<W>(s, idb, b);
    return mer_cmp<W>(a, b);
}

template <int W>
__device__ __forceinline__ u32 strand_of(const Src &s, u32 id)
{
    Mer<W> a;
    canon<W>(s, id, a);
    return a.sw;
}

// The first RL = splits * R bases of every sequence line, packed one read behind the other: one thread per 32 bases.
// A byte outside A C G T N (a line end included: the line is shorter than RL) is counted; the run is refused then.
// rec_stride != 0: the input is a fragment file of fastqSplit, line r = the R characters behind the 8-byte number of record r.
__global__ __launch_bounds__(CT) void k_reads_pack(const u8 *__restrict__ rd, const u64 *__restrict__ seq_at, u64 n_rd, u32 RL, u32 rec_stride,
                                                   u64 *__restrict__ rpk, u32 *__restrict__ rbad, u64 n_words, CollapseCounters *ctr)
{
    const u64 w = (u64)blockIdx.x * CT + threadIdx.x;
    if (w >= n_words) return;
    const u64 total = n_rd * RL, b0 = w * 32;
    u64 code = 0;
    u32 nm = 0, bad = 0;
    if (b0 < total) {
        u64 r = b0 / RL;
        u32 o = (u32)(b0 - r * RL);
        const u8 *p = rd + (rec_stride ? r * rec_stride + 8 : seq_at[r]);
        const u32 n = (u32)min((u64)32, total - b0);
        u32 j = 0;
        while (j < n) {                                      // one stretch per read the 32 bases touch
            const u32 len = min(n - j, RL - o);
            ByteRun run(p + o, len);
            while (run.more()) {
                u32 cnt;
                u32 four = run.next(cnt);
                // four bases at once: ((c >> 1) ^ (c >> 2)) & 3 is the code of A C G T, a byte permute turns the codes back
                // into letters, and if that gives the input again all four were A C G T
                const u32 c4 = ((four >> 1) ^ (four >> 2)) & 0x03030303u;
                if (cnt == 4 && __builtin_amdgcn_perm(0u, ACGT_PACKED, c4) == four) {
                    code |= (u64)((c4 * 0x40100401u) >> 24) << (56 - 2 * j);
                    j += 4;
                    continue;
                }
                for (u32 t = 0; t < cnt; t++, j++, four >>= 8) {
                    const u32 c = four & 0xFFu;
                    const u32 v = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
                    if (v == 4) { nm |= 0x80000000u >> j; bad += c != 'N'; }
                    else code |= (u64)v << (62 - 2 * j);
                }
            }
            o += len;
            if (o == RL) { o = 0; r++; if (r < n_rd) p = rd + (rec_stride ? r * rec_stride + 8 : seq_at[r]); }
        }
    }
    rpk[w] = code; rbad[w] = nm;
    if (bad) atomicAdd(&ctr->bad_read_bytes, (unsigned long long)bad);
}

__global__ __launch_bounds__(CT) void k_frag_codes(const u8 *__restrict__ rd, u64 n, u32 rec_stride, u64 *__restrict__ codes)
{
    const u64 r = (u64)blockIdx.x * CT + threadIdx.x;
    if (r >= n) return;
    const u8 *p = rd + r * rec_stride;
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v |= (u64)p[i] << (8 * i);
    codes[r] = v;
}

// ------------------------------------------------------------------ reference windows
struct FragFilter { u32 len; u32 mask[8]; };   // len 0: every window; else bit (c0 * 4 + c1 [* 4 + c2) * 4 + c3]) of the 2-bit codes of a listed signal

__device__ __forceinline__ u32 base_code(const u64 *__restrict__ pk, u64 b) { return (u32)(pk[b >> 5] >> (62 - 2 * (b & 31))) & 3u; }

__global__ __launch_bounds__(CT) void k_win_count(u64 G, u32 R, const u32 *__restrict__ bad, const u64 *__restrict__ pk, FragFilter ff,
                                                  u32 *__restrict__ blk, u32 *__restrict__ wvalid)
{
    __shared__ u32 wsum[4];
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    bool ok = g + R <= G;
    u64 p = g;
    u32 left = R;
    while (ok && left) {
        const u32 o = (u32)(p & 31), n = min(32u - o, left);
        const u32 mask = (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1u) << (32 - o - n));
        if (bad[p >> 5] & mask) ok = false;
        p += n; left -= n;
    }
    if (ok && ff.len) {                                     // fragmented mode: only windows whose signal this run lists (CL:712-835)
        u32 v;
        if (ff.len == 2) v = base_code(pk, g) * 4 + base_code(pk, g + R - 1);
        else v = ((base_code(pk, g) * 4 + base_code(pk, g + 1)) * 4 + base_code(pk, g + R - 2)) * 4 + base_code(pk, g + R - 1);
        ok = (ff.mask[v >> 5] >> (v & 31)) & 1u;
    }
    const u64 bal = __ballot(ok);
    const int lane = threadIdx.x & 63;
    if (lane == 0) wsum[threadIdx.x >> 6] = (u32)__popcll(bal);
    if (lane == 0 || lane == 32) wvalid[g >> 5] = __brev((u32)(bal >> lane));
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <int W>
__global__ __launch_bounds__(CT) void k_win_emit(Src s, const u32 *__restrict__ wvalid, const u64 *__restrict__ blk_off,
                                                 u64 *__restrict__ key, u32 *__restrict__ id, int clean_hash)
{
    __shared__ u32 lds[4];
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    const bool ok = g < s.G && ((wvalid[g >> 5] >> (31 - (u32)(g & 31))) & 1u);
    const u32 r = block_excl_scan(ok ? 1u : 0u, lds, nullptr);
    if (!ok) return;
    Mer<W> m;
    canon<W>(s, (u32)g, m);
    const u64 at = blk_off[blockIdx.x] + r;
    key[at] = clean_hash ? fnv_of_clean<W>(m, s.R) : fnv_of<W>(m, s.R); id[at] = (u32)g;
}

// ------------------------------------------------------------------ read-mers of the input
template <int W>
__global__ __launch_bounds__(CT) void k_read_keys(Src s, u64 n_items, u64 out_base, u64 *__restrict__ key, u32 *__restrict__ id)
{
    const u64 idx = (u64)blockIdx.x * CT + threadIdx.x;
    const bool live = idx < n_items;
    Mer<W> m;
    m.sw = 0;
    if (live) canon<W>(s, (u32)(s.G + idx), m);
    const u64 bal = __ballot(m.sw != 0);
    if ((threadIdx.x & 63) == 0) s.rsw[idx >> 6] = bal;
    if (!live) return;
    key[out_base + idx] = fnv_of<W>(m, s.R); id[out_base + idx] = (u32)(s.G + idx);
}

// ------------------------------------------------------------------ groups of equal read-mers
// head: 1 = first record of a group; 2 = same hash as the record before but another read-mer (repaired below)
template <int W>
__global__ __launch_bounds__(CT) void k_heads(Src s, const u64 *__restrict__ key, const u32 *__restrict__ id, u64 n, u64 mask,
                                              u32 *__restrict__ head, CollapseCounters *ctr)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i >= n) return;
    u32 h = 1;
    if (i > 0 && ((key[i] ^ key[i - 1]) & mask) == 0) {
        h = seq_cmp<W>(s, id[i], id[i - 1]) == 0 ? 0u : 2u;
        if (h == 2) atomicAdd(&ctr->n_mixed, 1ull);
    }
    head[i] = h;
}

// the first marked record of an equal-hash run owns the run's repair
__global__ __launch_bounds__(CT) void k_mixed_owners(const u64 *__restrict__ key, const u32 *__restrict__ head, u64 n, u64 mask,
                                                     u32 *__restrict__ owners, CollapseCounters *ctr)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i >= n || head[i] != 2) return;
    for (u64 q = i; q > 0 && ((key[q] ^ key[q - 1]) & mask) == 0; q--)
        if (q < i && head[q] == 2) return;
    owners[atomicAdd(&ctr->n_owners, 1ull)] = (u32)i;
}

// one thread per run: stable insertion sort of the run's ids by read-mer, then the group heads inside the run
template <int W>
__global__ __launch_bounds__(64) void k_mixed_fix(Src s, u64 *__restrict__ key, u32 *__restrict__ id, u64 n, u64 mask,
                                                  u32 *__restrict__ head, const u32 *__restrict__ owners, u64 n_owners)
{
    const u64 t = (u64)blockIdx.x * 64 + threadIdx.x;
    if (t >= n_owners) return;
    u64 st = owners[t], en = st + 1;
    while (st > 0 && ((key[st] ^ key[st - 1]) & mask) == 0) st--;
    while (en < n && ((key[en] ^ key[en - 1]) & mask) == 0) en++;
    for (u64 a = st + 1; a < en; a++) {
        const u32 x = id[a];
        const u64 kx = key[a];                  // the full hashes differ inside a run when the sort looked at fewer bits
        Mer<W> mx;
        canon<W>(s, x, mx);
        u64 b = a;
        while (b > st) {
            Mer<W> mo;
            canon<W>(s, id[b - 1], mo);
            if (mer_cmp<W>(mx, mo) >= 0) break;
            id[b] = id[b - 1]; key[b] = key[b - 1]; b--;
        }
        id[b] = x; key[b] = kx;
    }
    for (u64 q = st + 1; q < en; q++) head[q] = seq_cmp<W>(s, id[q], id[q - 1]) == 0 ? 0u : 1u;
}

__global__ __launch_bounds__(CT) void k_group_starts(const u32 *__restrict__ head, const u64 *__restrict__ gex, u64 n, u32 *__restrict__ gstart,
                                                     u64 n_groups)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i == 0) gstart[n_groups] = (u32)n;
    if (i < n && head[i]) gstart[gex[i]] = (u32)i;
}

__device__ __forceinline__ u32 first_read_of(const u32 *__restrict__ id, u32 st, u32 en, u64 G)
{
    u32 lo = st, hi = en;
    while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (id[mid] < G) lo = mid + 1; else hi = mid; }
    return lo;
}

// groups that hold at least one read-mer of the input are printed (CL:2311-2314)
__global__ __launch_bounds__(CT) void k_group_flags(const u32 *__restrict__ id, const u32 *__restrict__ gstart, u64 n_groups, u64 G,
                                                    u32 *__restrict__ flag)
{
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    if (g >= n_groups) return;
    const u32 st = gstart[g], en = gstart[g + 1];
    flag[g] = first_read_of(id, st, en, G) < en;
}

__global__ __launch_bounds__(CT) void k_group_emit(const u64 *__restrict__ key, const u32 *__restrict__ id, const u32 *__restrict__ gstart,
                                                   const u32 *__restrict__ flag, const u64 *__restrict__ oex, u64 n_groups, u64 B,
                                                   u64 *__restrict__ okey, u32 *__restrict__ oval)
{
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    if (g >= n_groups || !flag[g]) return;
    const u32 st = gstart[g];
    const u64 o = oex[g];
    okey[o] = ((key[st] % B) << 32) | (u64)(0xFFFFFFFFu - id[st]);     // bucket ascending, later insertions first
    oval[o] = (u32)g;
}

// ------------------------------------------------------------------ the order after the table has grown
// The reference's table re-buckets itself as it fills (collapse_host.cpp: cora_hip_collapse_table_growth).  tr1's
// _M_rehash walks the old buckets in ascending order, each from head to tail, and pushes every node at the HEAD of its
// new bucket; a new element goes to the head of its bucket.  With S(k) the iteration order while the table has its
// k-th bucket count B(k):
//     S(k) = stable sort by (hash mod B(k)) of  [ elements inserted under B(k), latest first ]  ++  reverse(S(k-1)).
// The relative order of two elements depends on those two alone, so only the printed groups go through this.
// An element's number in the insertion order (over distinct read-mers) comes from a bitmap over the ids: a group's first
// record is its insertion; elements before it = set bits below its id.
constexpr int MAX_GROWTH = 40;      // (rehash(1) reaches 2^32 elements in 33 steps)
struct Growth { u64 at[MAX_GROWTH], buckets[MAX_GROWTH]; int n; };

__global__ __launch_bounds__(CT) void k_mark_first(const u32 *__restrict__ id, const u32 *__restrict__ gstart, u64 n_groups, u32 *__restrict__ bm)
{
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    if (g >= n_groups) return;
    const u32 i = id[gstart[g]];
    atomicOr(&bm[i >> 5], 1u << (i & 31));
}
__global__ __launch_bounds__(CT) void k_word_popc(const u32 *__restrict__ bm, u64 n_words, u32 *__restrict__ cnt)
{
    const u64 w = (u64)blockIdx.x * CT + threadIdx.x;
    if (w < n_words) cnt[w] = (u32)__popc(bm[w]);
}
// Reference bases that are neither A C G T nor N (after upper-casing, CL:388-400), with their places: the windows that hold
// one enter the reference's table under keys of their own (CL:86-122: the complement table is zero for such a letter), which
// the host derives from this list (other_letter_elements below).
struct OtherLetter { u32 g, ch; };
__global__ __launch_bounds__(CT) void k_other_list(DevParams p, const u8 *__restrict__ raw, const u32 *__restrict__ bad, OtherLetter *__restrict__ out,
                                                   u32 cap, unsigned long long *__restrict__ n)
{
    const u64 w = (u64)blockIdx.x * CT + threadIdx.x;
    if (w >= p.n_words) return;
    u32 b = bad[w];
    while (b) {
        const int i = __clz((int)b);
        b &= ~(0x80000000u >> i);
        const u64 g = (w << 5) + (u64)i;
        u32 lo = 0, hi = p.n_contigs;
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (p.ctg_gstart[mid] <= g) lo = mid; else hi = mid; }
        const u64 local = g - p.ctg_gstart[lo];
        if (local >= p.ctg_len[lo]) continue;                                  // padding between contigs
        const u64 at = p.ctg_off[lo] + local + local / p.line_width;
        u32 ch = at < p.image_bytes ? raw[at] : (u32)'N';
        if (ch > 'Z') ch -= 32;
        if (ch == 'N') continue;
        const unsigned long long k = atomicAdd(n, 1ull);
        if (k < cap) { out[k].g = (u32)g; out[k].ch = ch; }
    }
}

// printed group -> (generation << 32 | ~(element number - 1)) : ascending = generation by generation, latest element first
__global__ __launch_bounds__(CT) void k_growth_keys(const u32 *__restrict__ id, const u32 *__restrict__ gstart, const u32 *__restrict__ flag,
                                                    const u64 *__restrict__ oex, u64 n_groups, const u32 *__restrict__ bm,
                                                    const u64 *__restrict__ wex, const u32 *__restrict__ other_ids, u32 n_other, Growth gr,
                                                    u64 *__restrict__ okey, u32 *__restrict__ oval)
{
    const u64 g = (u64)blockIdx.x * CT + threadIdx.x;
    if (g >= n_groups || !flag[g]) return;
    const u32 i = id[gstart[g]];
    u32 lo = 0, hi = n_other;                                  // elements under other-letter keys inserted before this one
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (other_ids[mid] < i) lo = mid + 1; else hi = mid; }
    const u64 element = wex[i >> 5] + (u64)__popc(bm[i >> 5] & ((1u << (i & 31)) - 1u)) + lo + 1;      // 1-based
    u32 gen = 0;
    for (int k = 1; k < gr.n; k++) if (gr.at[k] <= element) gen = (u32)k;
    const u64 o = oex[g];
    okey[o] = ((u64)gen << 32) | (u64)(0xFFFFFFFFu - (u32)(element - 1));
    oval[o] = (u32)g;
}
__global__ __launch_bounds__(CT) void k_gen_starts(const u64 *__restrict__ key, u64 m, u32 *__restrict__ start /* [MAX_GROWTH], preset to 0xFFFFFFFF */)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i >= m) return;
    const u32 g = (u32)(key[i] >> 32);
    if (i == 0 || (u32)(key[i - 1] >> 32) != g) start[g] = (u32)i;
}
__global__ __launch_bounds__(CT) void k_gather_hash(const u64 *__restrict__ key, const u32 *__restrict__ gstart, const u32 *__restrict__ grp, u64 m,
                                                    u64 *__restrict__ hs)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i < m) hs[i] = key[gstart[grp[i]]];
}
// the sequence a growth step sorts: the newcomers (a range of the generation-sorted list), then the order so far, reversed
__global__ __launch_bounds__(CT) void k_level_build(const u64 *__restrict__ hs, u64 B, u32 new_lo, u32 n_new, const u32 *__restrict__ prev, u32 n_prev,
                                                    u64 *__restrict__ okey, u32 *__restrict__ oval)
{
    const u64 j = (u64)blockIdx.x * CT + threadIdx.x;
    if (j >= (u64)n_new + n_prev) return;
    const u32 e = j < n_new ? new_lo + (u32)j : prev[n_prev - 1 - (u32)(j - n_new)];
    okey[j] = hs[e] % B;
    oval[j] = e;
}
__global__ __launch_bounds__(CT) void k_take(const u32 *__restrict__ from, const u32 *__restrict__ idx, u64 m, u32 *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * CT + threadIdx.x;
    if (i < m) out[i] = from[idx[i]];
}

// ------------------------------------------------------------------ text of the three files
struct OutParams {
    u32 D, mapper, n_contigs;      // mapper: 0 GENERIC, 1 BWA, 2 BWA_MEM, 3 BOWTIE_2
    u64 lrn_offset;
    const u64 *ctg_gstart;
};

__device__ __forceinline__ u32 dec_digits(u64 v) { u32 n = 1; while (v >= 10) { v /= 10; n++; } return n; }
template <class P>
__device__ __forceinline__ u32 put_dec(P dst, u64 v)
{
    const u32 n = dec_digits(v);
    for (u32 k = n; k-- > 0;) { dst[k] = (u8)('0' + v % 10); v /= 10; }
    return n;
}

// id digits of the read-mer at sorted position q of a group whose first record is `first` (CL:948-972, 276-289, 491-506)
template <int W>
__device__ __forceinline__ u64 id_value(const Src &s, const u32 *__restrict__ id, u32 q, bool in_ref, u32 ref_dir)
{
    const u64 idx = (u64)id[q] - s.G;                                   // always a read-mer of the input
    const u32 sw = (u32)(s.rsw[idx >> 6] >> (idx & 63)) & 1u;
    return (s.codes ? s.codes[idx] : 2 * idx) + (in_ref ? (sw != ref_dir) : sw);     // CL:491-506 on the number fastqSplit wrote
}

__device__ __forceinline__ u32 lrn_len(u64 number, u32 D)
{
    const u32 n = dec_digits(number);
    return n + 2 + ((n + 2) % D == 0 ? 1u : 0u);                        // CL:312-326
}

template <int W>
__global__ __launch_bounds__(CT) void k_out_flags(Src s, OutParams op, const u32 *__restrict__ id, const u32 *__restrict__ gstart,
                                                  const u32 *__restrict__ oval, u64 m, u32 *__restrict__ lrn_flag, CollapseCounters *ctr)
{
    const u64 j = (u64)blockIdx.x * CT + threadIdx.x;
    if (j >= m) return;
    const u32 g = oval[j], st = gstart[g], en = gstart[g + 1];
    const bool in_ref = id[st] < s.G;
    u32 lrn = 0;
    if (!in_ref) {
        const u64 name_len = (u64)op.D * (en - st);
        if (name_len > 0xF0000000ull) atomicAdd(&ctr->name_overflow, 1ull);
        if (op.mapper == 0 || op.mapper == 3) lrn = name_len > 194;                                    // CL:2321, 2381
        else {
            const u64 v = id_value<W>(s, id, en - 1, false, 0);
            const u32 c2 = (u32)(v % 94) + 33, c1 = (u32)((v / 94) % 94) + 33;                         // D >= 2
            lrn = name_len > 5000000 || (c1 == '/' && (op.mapper == 1 ? (c2 == '1' || c2 == '2') : (c2 >= '0' && c2 <= '9')));   // CL:2339-2342, 2360-2363
        }
    }
    lrn_flag[j] = lrn;
}

__device__ __forceinline__ void ref_place(const OutParams &op, u64 g, u32 &code, u32 &pos)
{
    u32 lo = 0, hi = op.n_contigs;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (op.ctg_gstart[mid] <= g) lo = mid; else hi = mid; }
    code = lo + 1; pos = (u32)(g - op.ctg_gstart[lo]) + 1;
}

__global__ __launch_bounds__(CT) void k_out_sizes(Src s, OutParams op, const u32 *__restrict__ id, const u32 *__restrict__ gstart,
                                                  const u32 *__restrict__ oval, u64 m, const u32 *__restrict__ lrn_flag,
                                                  const u64 *__restrict__ lrn_rank, u32 *__restrict__ sz_r, u32 *__restrict__ sz_p,
                                                  u32 *__restrict__ sz_l)
{
    const u64 j = (u64)blockIdx.x * CT + threadIdx.x;
    if (j >= m) return;
    const u32 g = oval[j], st = gstart[g], en = gstart[g + 1];
    const bool in_ref = id[st] < s.G;
    const u32 fr = first_read_of(id, st, en, s.G);
    const u32 name_len = op.D * (en - fr);
    u32 r = 0, p = 0, l = 0;
    if (in_ref) {
        u32 code, pos;
        ref_place(op, id[st], code, pos);
        p = name_len + 1 + dec_digits(code) + 1 + dec_digits(pos) + 1 + dec_digits(s.R) + 1;          // CL:2400-2401
    } else {
        u32 shown = name_len;
        if (lrn_flag[j]) { shown = lrn_len(op.lrn_offset + lrn_rank[j] + 1, op.D); l = shown + 1 + name_len + 1; }
        r = 1 + shown + 1 + s.R + 1 + (op.mapper == 3 ? 2 + s.R + 1 : 0);
    }
    sz_r[j] = r; sz_p[j] = p; sz_l[j] = l;
}

template <int W, class P>
__device__ __forceinline__ void put_name(const Src &s, u32 D, const u32 *__restrict__ id, u32 fr, u32 en, bool in_ref, u32 ref_dir, P dst)
{
    for (u32 q = fr; q < en; q++, dst += D) {
        u64 v = id_value<W>(s, id, q, in_ref, ref_dir);
        for (u32 k = D; k-- > 0;) { dst[k] = (u8)(v % 94 + 33); v /= 94; }
    }
}

struct Group { u32 first, fr, en, name_len; bool in_ref; };

// one line of the perfect-match file (CL:2400-2401)
template <int W, class P>
__device__ __forceinline__ void write_perfect(const Src &s, const OutParams &op, const u32 *__restrict__ id, const Group &g, u32 head_sw, P w)
{
    put_name<W>(s, op.D, id, g.fr, g.en, true, head_sw, w);
    w += g.name_len;
    u32 code, pos;
    ref_place(op, g.first, code, pos);
    *w++ = '\t'; w += put_dec(w, code);
    *w++ = '\t'; w += put_dec(w, pos);
    *w++ = '\t'; w += put_dec(w, s.R);
    *w = '\n';
}

// one record of the collapsed-reads file (CL:2317-2396); `shown` = 0: the name itself, else the long-read-name number
template <int W, class P>
__device__ __forceinline__ void write_read(const Src &s, const OutParams &op, const u32 *__restrict__ id, const Group &g, const Mer<W> &head,
                                           u64 lrn_number, P w)
{
    *w++ = op.mapper == 3 ? '@' : '>';
    if (lrn_number) {
        const u32 n = dec_digits(lrn_number), extra = (n + 2) % op.D == 0;
        *w++ = '#'; w += put_dec(w, lrn_number); *w++ = '#';
        if (extra) *w++ = '#';
    } else {
        put_name<W>(s, op.D, id, g.fr, g.en, false, 0, w);
        w += g.name_len;
    }
    *w++ = '\n';
    put_mer<W>(w, head, s.R);
    w += s.R;
    *w++ = '\n';
    if (op.mapper == 3) {                                                                              // CL:2394-2395
        *w++ = '+'; *w++ = '\n';
        for (u32 i = 0; i < s.R; i++) w[i] = 'S';
        w[s.R] = '\n';
    }
}

// one line of the long-read-name file (CL:312-326)
template <int W, class P>
__device__ __forceinline__ void write_lrn(const Src &s, const OutParams &op, const u32 *__restrict__ id, const Group &g, u64 lrn_number, P l)
{
    const u32 n = dec_digits(lrn_number), extra = (n + 2) % op.D == 0;
    *l++ = '#'; l += put_dec(l, lrn_number); *l++ = '#';
    if (extra) *l++ = '#';
    *l++ = '\t';
    put_name<W>(s, op.D, id, g.fr, g.en, false, 0, l);
    l[g.name_len] = '\n';
}

constexpr u32 STAGE_BYTES = 48 * 1024;

// bytes [0, total) staged at lds[lead ...] -> out[base ...], 16 bytes per lane where whole chunks exist (lead = base % 16, so
// LDS and global addresses are aligned alike)
__device__ __forceinline__ void flush_stage(const u8 *lds, u8 *__restrict__ out, u64 base, u32 total)
{
    const u32 lead = (u32)(base & 15);
    u8 *abase = out + (base - lead);
    const u32 end = lead + total, chunks = (end + 15) / 16;
    for (u32 c = threadIdx.x; c < chunks; c += CT) {
        const u32 lo = c * 16, hi = lo + 16;
        if (lo >= lead && hi <= end) *reinterpret_cast<uint4 *>(abase + lo) = *reinterpret_cast<const uint4 *>(lds + lo);
        else for (u32 b = max(lo, lead); b < min(hi, end); b++) abase[b] = lds[b];
    }
}

// A workgroup's records are consecutive in each of the three files, so it builds its piece of each file in LDS and
// writes it out in whole 16-byte chunks; a piece that does not fit (long names: heavily repeated read-mers) is written
// record by record straight to memory.
template <int W>
__global__ __launch_bounds__(CT) void k_out_write(Src s, OutParams op, const u32 *__restrict__ id, const u32 *__restrict__ gstart,
                                                  const u32 *__restrict__ oval, u64 m, const u32 *__restrict__ lrn_flag,
                                                  const u64 *__restrict__ lrn_rank, const u64 *__restrict__ off_r,
                                                  const u64 *__restrict__ off_p, const u64 *__restrict__ off_l,
                                                  const u32 *__restrict__ sz_r, const u32 *__restrict__ sz_p, const u32 *__restrict__ sz_l,
                                                  u8 *__restrict__ out_r, u8 *__restrict__ out_p, u8 *__restrict__ out_l)
{
    __shared__ __align__(16) u8 stage[STAGE_BYTES + 16];
    const u64 j0 = (u64)blockIdx.x * CT, j = j0 + threadIdx.x;
    const u64 jl = min(j0 + CT, m) - 1;
    const bool live = j < m;
    Group g;
    g.first = 0; g.fr = g.en = 0; g.name_len = 0; g.in_ref = false;
    Mer<W> head;
    u64 lrn_number = 0;
    if (live) {
        const u32 gi = oval[j], st = gstart[gi];
        g.en = gstart[gi + 1];
        g.first = id[st];
        g.in_ref = g.first < s.G;
        g.fr = first_read_of(id, st, g.en, s.G);
        g.name_len = op.D * (g.en - g.fr);
        canon<W>(s, g.first, head);
        if (!g.in_ref && lrn_flag[j]) lrn_number = op.lrn_offset + lrn_rank[j] + 1;
    }
    {   // collapsed reads
        const u64 base = off_r[j0], total = off_r[jl] + sz_r[jl] - base;
        const bool mine = live && !g.in_ref;
        if (total <= STAGE_BYTES) {
            if (mine) write_read<W>(s, op, id, g, head, lrn_number, stage + (u32)(base & 15) + (u32)(off_r[j] - base));
            __syncthreads();
            flush_stage(stage, out_r, base, (u32)total);
            __syncthreads();
        } else if (mine) write_read<W>(s, op, id, g, head, lrn_number, out_r + off_r[j]);
    }
    {   // perfect matches
        const u64 base = off_p[j0], total = off_p[jl] + sz_p[jl] - base;
        const bool mine = live && g.in_ref;
        if (total <= STAGE_BYTES) {
            if (mine) write_perfect<W>(s, op, id, g, head.sw, stage + (u32)(base & 15) + (u32)(off_p[j] - base));
            __syncthreads();
            flush_stage(stage, out_p, base, (u32)total);
            __syncthreads();
        } else if (mine) write_perfect<W>(s, op, id, g, head.sw, out_p + off_p[j]);
    }
    {   // long read names
        const u64 base = off_l[j0], total = off_l[jl] + sz_l[jl] - base;
        const bool mine = live && lrn_number != 0;
        if (total == 0) return;
        if (total <= STAGE_BYTES) {
            if (mine) write_lrn<W>(s, op, id, g, lrn_number, stage + (u32)(base & 15) + (u32)(off_l[j] - base));
            __syncthreads();
            flush_stage(stage, out_l, base, (u32)total);
        } else if (mine) write_lrn<W>(s, op, id, g, lrn_number, out_l + off_l[j]);
    }
}


// kernels over read-mers exist for 4 words (up to 128 bases) and 10 words (up to 320; the reference's lines end at 298 bases, CL:27)
#define LAUNCH_W(kernel, grid, block, stream, ...)                                                              \
    do {                                                                                                        \
        if (R <= 128) hipLaunchKernelGGL(kernel<4>, grid, block, 0, stream, __VA_ARGS__);                        \
        else hipLaunchKernelGGL(kernel<10>, grid, block, 0, stream, __VA_ARGS__);                                \
    } while (0)

}  // namespace

// The table elements that come from reference windows with a letter outside A C G T N: CL:656-835 hashes every N-free window;
// CheckIfComplementIsEarlier (CL:97-122) compares it with a "reverse complement" whose entries for such letters are NUL, and
// the key is the window itself or that string cut at its first NUL.  No read-mer can equal such a key (they are never
// printed), but they are elements: they count towards the table's growth and sit in its insertion order.  Returns the ids
// (global position of the first window under each distinct key), ascending.  Host work on the few places concerned.
static inline char oth_comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N'; } return 0; }
static void other_letter_elements(const HostGenome &hg, u32 R, const FragFilter &ff, const std::vector<u64> &pk, const std::vector<u32> &bad,
                                  std::vector<OtherLetter> &letters, std::vector<u32> &ids)
{
    std::sort(letters.begin(), letters.end(), [](const OtherLetter &a, const OtherLetter &b) { return a.g < b.g; });
    std::unordered_set<std::string> seen;
    std::string win(R, 'A'), key(R, 'A');
    auto letter_at = [&](u64 g) -> char {
        if ((bad[g >> 5] >> (31 - (g & 31))) & 1u) {
            auto it = std::lower_bound(letters.begin(), letters.end(), (u32)g, [](const OtherLetter &a, u32 v) { return a.g < v; });
            return (it != letters.end() && it->g == (u32)g) ? (char)it->ch : 'N';
        }
        return "ACGT"[(pk[g >> 5] >> (62 - 2 * (g & 31))) & 3u];
    };
    u64 next_start = 0;                                            // windows below this start have been looked at
    for (size_t li = 0; li < letters.size(); li++) {
        const u64 g = letters[li].g;
        const u32 c = hg.contig_of(g);
        const u64 cs = hg.gstart[c], ce = cs + hg.len[c];
        if (hg.len[c] < R) continue;
        u64 lo = g >= cs + (R - 1) ? g - (R - 1) : cs, hi = std::min<u64>(g, ce - R);       // window starts that cover the letter
        if (lo < next_start) lo = next_start;
        for (u64 st = lo; st <= hi && hi != (u64)-1; st++) {
            bool has_n = false;
            for (u32 k = 0; k < R; k++) { win[k] = letter_at(st + k); if (win[k] == 'N') { has_n = true; break; } }
            if (has_n) continue;
            if (ff.len) {                                          // fragmented mode: the window's end letters must spell a listed signal
                u32 code = 0;
                bool acgt = true;
                const int pos[4] = {0, ff.len == 2 ? (int)R - 1 : 1, (int)R - 2, (int)R - 1};
                for (u32 q = 0; q < ff.len; q++) {
                    const char ch = win[pos[q]];
                    const char *f = strchr("ACGT", ch);
                    if (!f || !ch) { acgt = false; break; }
                    code = code * 4 + (u32)(f - "ACGT");
                }
                if (!acgt || !((ff.mask[code >> 5] >> (code & 31)) & 1u)) continue;
            }
            size_t klen = R;
            bool done = false;
            for (u32 i = 0; i < R && !done; i++) {                 // CL:97-122
                const char cc = oth_comp(win[R - 1 - i]);
                if (win[i] < cc) { key = win; done = true; }
                else if (win[i] > cc) {
                    for (u32 k = 0; k < R; k++) key[k] = oth_comp(win[R - 1 - k]);
                    for (u32 k = 0; k < R; k++) if (!key[k]) { klen = k; break; }
                    done = true;
                }
            }
            if (!done) key = win;
            if (seen.insert(key.substr(0, klen)).second) ids.push_back((u32)st);
        }
        if (hi != (u64)-1 && hi + 1 > next_start) next_start = hi + 1;
    }
}

struct cora_hip_collapse {
    cora_hip_collapse_params prm;
    u32 R = 0;
    u64 B = 0, first_rehash_at = 0;
    Growth growth = {};                       // the table's bucket counts and the element numbers at which it grows
    hipStream_t stream = nullptr;
    cora_hip_job *ref_job = nullptr;          // holds the packed reference (pk, bad, contig table)
    std::vector<u8> h_rd;                     // FASTQ images, 16-byte aligned starts
    std::vector<ReadFile> files;                  // in list order: sample 0 mate 1, sample 0 mate 2, ...
    HBuf<u8> out_r, out_p, out_l;
    u64 n_r = 0, n_p = 0, n_l = 0;
    SortScratch sort_scratch;
    DevPool pool;                             // device memory of a run, kept for the next run of this object
    FragFilter frag = {};                     // fragmented mode (cora_hip_collapse_set_fragment): the input is one fragment file
    bool fragmented = false;
    bool has_dev_in = false;                  // fragmented mode: a fragment that is already in HBM (cora_hip_collapse_add_reads_device)
    const u8 *dev_in = nullptr;
    size_t dev_in_bytes = 0;
    cora_hip_collapse_stats st;
    ~cora_hip_collapse()
    {
        sort_scratch.release();
        pool.destroy();
        if (ref_job) cora_hip_job_destroy(ref_job);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int run_impl(cora_hip_collapse *c)
{
    hipStream_t s = c->stream;
    const cora_hip_collapse_params &prm = c->prm;
    cora_hip_collapse_stats &st = c->st;
    memset(&st, 0, sizeof st);
    const u32 R = c->R, splits = (u32)prm.splits;
    const auto t_begin = std::chrono::steady_clock::now();
    hipEvent_t ev[6];
    for (auto &e : ev) HIPQ(hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 6; i++) (void)hipEventDestroy(e[i]); } } guard{ev};

    DBuf<CollapseCounters> ctr;
    DBuf<u64> tot;
    RCQ(ctr.alloc(1)); RCQ(tot.alloc(4));
    HIPQ(hipMemsetAsync(ctr.p, 0, sizeof(CollapseCounters), s));

    // ---- the reads: images to HBM, sequence line of every record ----
    DBuf<u8> rd;
    if (!c->has_dev_in) {
        RCQ(rd.alloc(c->h_rd.size() + 1024));
        HIPQ(hipMemsetAsync(rd.p + c->h_rd.size(), 0, 1024, s));
        if (!c->h_rd.empty()) HIPQ(hipMemcpyAsync(rd.p, c->h_rd.data(), c->h_rd.size(), hipMemcpyHostToDevice, s));
    } else if (!c->fragmented || !c->files.empty()) return fail(19, "a device-resident input is one fragment file: the fragmented mode, and no other input");
    HIPQ(hipEventRecord(ev[0], s));
    DBuf<u64> seq_at, rpk, rsw, codes;
    DBuf<u32> rbad;
    u64 n_rd = 0, n_items = 0;                              // sequence lines, read-mers
    u64 rpk_words = 0;
    if (!c->fragmented) {
        RCQ(locate_sequence_lines(s, rd.p, c->h_rd, c->files, prm.paired != 0, tot.p, seq_at, n_rd));
        n_items = n_rd * splits;
        rpk_words = (n_items * R + 31) / 32 + 12;
        RCQ(rpk.alloc(rpk_words)); RCQ(rbad.alloc(rpk_words)); RCQ(rsw.alloc(n_items / 64 + 8));
        hipLaunchKernelGGL(k_reads_pack, dim3(grid_of(rpk_words, CT)), dim3(CT), 0, s, rd.p, seq_at.p, n_rd, R * splits, 0u, rpk.p, rbad.p, rpk_words, ctr.p);
    } else {                                                // CL:2173-2284: whole records of 8 + R bytes (fread counts items)
        if (!c->has_dev_in && c->files.size() != 1) return fail(19, "the fragmented mode reads one fragment file (argv[1]); %zu were added", c->files.size());
        const u32 rec = R + 8;
        const u8 *frag = c->has_dev_in ? c->dev_in : rd.p + c->files[0].off;
        n_items = n_rd = (c->has_dev_in ? c->dev_in_bytes : c->files[0].bytes) / rec;
        rpk_words = (n_items * R + 31) / 32 + 12;
        RCQ(rpk.alloc(rpk_words)); RCQ(rbad.alloc(rpk_words)); RCQ(rsw.alloc(n_items / 64 + 8)); RCQ(codes.alloc(n_items + 1));
        hipLaunchKernelGGL(k_reads_pack, dim3(grid_of(rpk_words, CT)), dim3(CT), 0, s, frag, (const u64 *)nullptr, n_rd, R, rec, rpk.p,
                           rbad.p, rpk_words, ctr.p);
        if (n_items) hipLaunchKernelGGL(k_frag_codes, dim3(grid_of(n_items, CT)), dim3(CT), 0, s, frag, n_items, rec, codes.p);
    }

    // ---- the reference: packed by the table builder's kernel ----
    u64 G = 0, NW = 0, other_letters = 0;
    DevParams dp;
    memset(&dp, 0, sizeof dp);
    DBuf<u32> wvalid, wblk;
    DBuf<u64> woff;
    cora_hip_job *rj = c->ref_job;
    if (rj) {
        PoolScope no_pool(nullptr);
        RCQ(st_pack(rj));
        DevCounters h;
        RCQ(read_counters(rj, h));
        FORMAT_ERROR_CHECK(h);
        other_letters = h.other_letters;
        dp = rj->dp;
        G = dp.G;
        const u32 wb = grid_of(G, CT);
        RCQ(wvalid.alloc((u64)wb * CT / 32 + 4)); RCQ(wblk.alloc(wb)); RCQ(woff.alloc(wb));
        hipLaunchKernelGGL(k_win_count, dim3(wb), dim3(CT), 0, s, G, R, rj->bad.p, rj->pk.p, c->frag, wblk.p, wvalid.p);
        RCQ(scan_u32(s, wblk.p, woff.p, wb, tot.p));
        HIPQ(hipMemcpyAsync(&NW, tot.p, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
    }
    if (G + n_items >= 0xFFFFFFFFull) return fail(19, "reference positions + read-mers = %llu: 32-bit record ids do not hold that", (unsigned long long)(G + n_items));
    { u64 lim = 1; for (int i = 0; i < prm.digit_len; i++) lim *= 94;
      if (!c->fragmented && n_items && 2 * n_items - 1 >= lim) return fail(19, "digit length %d cannot number %llu read-mers", prm.digit_len, (unsigned long long)n_items); }
    const u64 N = NW + n_items;
    st.ref_windows = NW; st.read_mers = n_items; st.reads = n_rd;
    HIPQ(hipEventRecord(ev[1], s));

    // ---- records ----
    Src src;
    src.pk = rj ? rj->pk.p : nullptr; src.pk_words = rj ? rj->pk.n : 0; src.rpk = rpk.p; src.rbad = rbad.p; src.rpk_words = rpk_words; src.rsw = rsw.p; src.codes = c->fragmented ? codes.p : nullptr;
    src.G = G; src.R = R; src.splits = splits;
    DBuf<u64> key, tkey;
    DBuf<u32> id, tid;
    RCQ(key.alloc(N + 1)); RCQ(tkey.alloc(N + 1)); RCQ(id.alloc(N + 1)); RCQ(tid.alloc(N + 1));
    const int clean_hash = getenv("CORA_COLLAPSE_CLEAN_HASH") ? atoi(getenv("CORA_COLLAPSE_CLEAN_HASH")) : 0;   // the loop without the N test
    if (rj && G) LAUNCH_W(k_win_emit, dim3(grid_of(G, CT)), dim3(CT), s, src, wvalid.p, woff.p, key.p, id.p, clean_hash);
    if (n_items) LAUNCH_W(k_read_keys, dim3(grid_of(n_items, CT)), dim3(CT), s, src, n_items, NW, key.p, id.p);
    CollapseCounters hc;
    HIPQ(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    if (hc.bad_read_bytes)
        return fail(78, "%llu read bytes are outside A C G T N or a sequence line is shorter than %u (the reference would cut such keys at a NUL, CL:86-122)",
                    (unsigned long long)hc.bad_read_bytes, R * splits);
    wvalid.release(); wblk.release(); woff.release();
    HIPQ(hipEventRecord(ev[2], s));

    // ---- equal read-mers side by side, in insertion order ----
    int hash_bits = 64;
    if (const char *e = getenv("CORA_COLLAPSE_HASH_BITS")) { const int v = atoi(e); if (v >= 8 && v <= 64) hash_bits = v & ~7; }
    const u64 mask = hash_bits == 64 ? ~0ull : ((1ull << hash_bits) - 1);
    SortArrays ka, kt;
    ka.a = key.p; ka.c = id.p; kt.a = tkey.p; kt.c = tid.p;
    SortTiming timing;
    timing.enabled = true;
    if (N && radix_sort(s, ka, kt, N, 0, hash_bits, c->sort_scratch, &timing)) return fail(100, "radix sort failed");
    u64 *skey = ka.a;
    u32 *sid = ka.c;
    DBuf<u32> head;
    RCQ(head.alloc(N + 1));
    if (N) LAUNCH_W(k_heads, dim3(grid_of(N, CT)), dim3(CT), s, src, skey, sid, N, mask, head.p, ctr.p);
    HIPQ(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    st.hash_collisions = hc.n_mixed;
    if (hc.n_mixed) {
        DBuf<u32> owners;
        RCQ(owners.alloc(hc.n_mixed));
        hipLaunchKernelGGL(k_mixed_owners, dim3(grid_of(N, CT)), dim3(CT), 0, s, skey, head.p, N, mask, owners.p, ctr.p);
        HIPQ(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        LAUNCH_W(k_mixed_fix, dim3(grid_of(hc.n_owners, 64)), dim3(64), s, src, skey, sid, N, mask, head.p, owners.p, (u64)hc.n_owners);
        HIPQ(hipStreamSynchronize(s));
    }
    u64 n_groups = 0;
    DBuf<u64> gex;
    DBuf<u32> gstart;
    RCQ(gex.alloc(N + 1));
    if (N) {
        RCQ(scan_u32(s, head.p, gex.p, N, tot.p));
        HIPQ(hipMemcpyAsync(&n_groups, tot.p, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
    }
    st.distinct = n_groups;
    // The reference's table re-buckets itself with its first_rehash_at-th element, and again later (Growth): from then on the
    // order needs every element's number in the insertion order.  Windows with letters other than A C G T N are elements of
    // that table too (under keys of their own, never printed): which, and where in the insertion order, is worked out on the
    // host when the table may grow (other_letter_elements).
    const u64 garbage_bound = other_letters * R;
    std::vector<u32> other_ids;                                    // ids of the elements under other-letter keys, ascending
    DBuf<u32> d_other_ids;
    if (n_groups + garbage_bound + 1 >= c->first_rehash_at && other_letters) {
        if (other_letters > 100000)
            return fail(77, "%llu reference letters other than A C G T N and a hash map that grows (%llu read-mers): their windows' places in the "
                            "reference's insertion order are derived on the host, for up to 100000 such letters", (unsigned long long)other_letters,
                        (unsigned long long)n_groups);
        DBuf<OtherLetter> d_ol;
        DBuf<u64> d_n;
        RCQ(d_ol.alloc(other_letters + 1)); RCQ(d_n.alloc(1));
        HIPQ(hipMemsetAsync(d_n.p, 0, 8, s));
        hipLaunchKernelGGL(k_other_list, dim3(grid_of(dp.n_words, CT)), dim3(CT), 0, s, dp, rj->raw.p, rj->bad.p, d_ol.p, (u32)other_letters,
                           (unsigned long long *)d_n.p);
        u64 n_ol = 0;
        HIPQ(hipMemcpyAsync(&n_ol, d_n.p, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        if (n_ol != other_letters) return fail(100, "other-letter list holds %llu entries, the pack kernel counted %llu", (unsigned long long)n_ol, (unsigned long long)other_letters);
        std::vector<OtherLetter> letters(n_ol);
        std::vector<u64> h_pk(dp.n_words + 4);
        std::vector<u32> h_bad(dp.n_words + 4);
        HIPQ(hipMemcpy(letters.data(), d_ol.p, n_ol * sizeof(OtherLetter), hipMemcpyDeviceToHost));
        HIPQ(hipMemcpy(h_pk.data(), rj->pk.p, (dp.n_words + 4) * sizeof(u64), hipMemcpyDeviceToHost));
        HIPQ(hipMemcpy(h_bad.data(), rj->bad.p, (dp.n_words + 4) * sizeof(u32), hipMemcpyDeviceToHost));
        other_letter_elements(rj->hg, R, c->frag, h_pk, h_bad, letters, other_ids);
        RCQ(d_other_ids.alloc(other_ids.size() + 1));
        if (!other_ids.empty()) HIPQ(hipMemcpyAsync(d_other_ids.p, other_ids.data(), other_ids.size() * sizeof(u32), hipMemcpyHostToDevice, s));
    } else RCQ(d_other_ids.alloc(1));
    const u64 elements = n_groups + other_ids.size();              // of the reference's table
    const bool grown = elements >= c->first_rehash_at;
    RCQ(gstart.alloc(n_groups + 1));
    hipLaunchKernelGGL(k_group_starts, dim3(grid_of(N, CT)), dim3(CT), 0, s, head.p, gex.p, N, gstart.p, n_groups);
    head.release(); gex.release();
    HIPQ(hipEventRecord(ev[3], s));

    // ---- printed groups in the hash map's iteration order ----
    u64 M = 0;
    DBuf<u32> gflag;
    DBuf<u64> oex;
    RCQ(gflag.alloc(n_groups + 1)); RCQ(oex.alloc(n_groups + 1));
    if (n_groups) {
        hipLaunchKernelGGL(k_group_flags, dim3(grid_of(n_groups, CT)), dim3(CT), 0, s, sid, gstart.p, n_groups, G, gflag.p);
        RCQ(scan_u32(s, gflag.p, oex.p, n_groups, tot.p));
        HIPQ(hipMemcpyAsync(&M, tot.p, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
    }
    st.printed = M;
    DBuf<u64> okey, tokey;
    DBuf<u32> oval, toval, grown_ord;
    RCQ(okey.alloc(M + 1)); RCQ(tokey.alloc(M + 1)); RCQ(oval.alloc(M + 1)); RCQ(toval.alloc(M + 1));
    SortArrays oa, ot;
    oa.a = okey.p; oa.c = oval.p; ot.a = tokey.p; ot.c = toval.p;
    const u32 *ord = nullptr;
    st.table_growth_steps = 0;
    if (!grown) {
        if (M) hipLaunchKernelGGL(k_group_emit, dim3(grid_of(n_groups, CT)), dim3(CT), 0, s, skey, sid, gstart.p, gflag.p, oex.p, n_groups, c->B, okey.p, oval.p);
        gflag.release(); oex.release();
        int bucket_bits = 1;
        while ((1ull << bucket_bits) < c->B) bucket_bits++;
        if (M && radix_sort(s, oa, ot, M, 0, 32 + bucket_bits, c->sort_scratch, &timing)) return fail(100, "radix sort failed");
        ord = oa.c;
    } else {
        // ---- the table grew (see k_mark_first ...): element numbers, generations, then one stable sort per bucket count ----
        const Growth &gr = c->growth;
        int g_last = 0;
        for (int k = 1; k < gr.n; k++) if (gr.at[k] <= elements) g_last = k;
        st.table_growth_steps = (u32)g_last;
        const u64 id_words = (std::min<u64>((u64)G + N, 1ull << 32) + 31) / 32 + 1;    // ids: reference positions below G, then G + read-mer index
        DBuf<u32> bm, wcnt, gen_start;
        DBuf<u64> wex, hs;
        RCQ(bm.alloc(id_words)); RCQ(wcnt.alloc(id_words)); RCQ(wex.alloc(id_words + 1)); RCQ(gen_start.alloc(MAX_GROWTH)); RCQ(hs.alloc(M + 1));
        RCQ(grown_ord.alloc(M + 1));
        HIPQ(hipMemsetAsync(bm.p, 0, id_words * sizeof(u32), s));
        hipLaunchKernelGGL(k_mark_first, dim3(grid_of(n_groups, CT)), dim3(CT), 0, s, sid, gstart.p, n_groups, bm.p);
        hipLaunchKernelGGL(k_word_popc, dim3(grid_of(id_words, CT)), dim3(CT), 0, s, bm.p, id_words, wcnt.p);
        RCQ(scan_u32(s, wcnt.p, wex.p, id_words, tot.p));
        if (M) hipLaunchKernelGGL(k_growth_keys, dim3(grid_of(n_groups, CT)), dim3(CT), 0, s, sid, gstart.p, gflag.p, oex.p, n_groups, bm.p, wex.p, d_other_ids.p, (u32)other_ids.size(), gr, okey.p, oval.p);
        gflag.release(); oex.release(); wcnt.release();
        if (M && radix_sort(s, oa, ot, M, 0, 40, c->sort_scratch, &timing)) return fail(100, "radix sort failed");
        // oa: printed groups, generation by generation, latest element first; where each generation starts
        u32 h_start[MAX_GROWTH + 1];
        HIPQ(hipMemsetAsync(gen_start.p, 0xFF, MAX_GROWTH * sizeof(u32), s));
        if (M) hipLaunchKernelGGL(k_gen_starts, dim3(grid_of(M, CT)), dim3(CT), 0, s, oa.a, M, gen_start.p);
        HIPQ(hipMemcpyAsync(h_start, gen_start.p, MAX_GROWTH * sizeof(u32), hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        h_start[MAX_GROWTH] = (u32)M;
        for (int k = MAX_GROWTH - 1; k >= 0; k--) if (h_start[k] == 0xFFFFFFFFu) h_start[k] = h_start[k + 1];       // a generation without printed groups
        if (M) hipLaunchKernelGGL(k_gather_hash, dim3(grid_of(M, CT)), dim3(CT), 0, s, skey, gstart.p, oa.c, M, hs.p);
        // the generation-sorted group list stays where it is (gen_grp); the levels sort indices into it
        DBuf<u32> gen_grp, lv_a, lv_b, cur;
        DBuf<u64> lk_a, lk_b;
        RCQ(gen_grp.alloc(M + 1)); RCQ(lv_a.alloc(M + 1)); RCQ(lv_b.alloc(M + 1)); RCQ(cur.alloc(M + 1)); RCQ(lk_a.alloc(M + 1)); RCQ(lk_b.alloc(M + 1));
        if (M) HIPQ(hipMemcpyAsync(gen_grp.p, oa.c, M * sizeof(u32), hipMemcpyDeviceToDevice, s));
        u32 n_cur = 0;
        for (int k = 0; k <= g_last && M; k++) {
            const u32 new_lo = h_start[k], n_new = h_start[k + 1] - h_start[k];
            const u64 n = (u64)n_new + n_cur;
            if (!n) continue;
            hipLaunchKernelGGL(k_level_build, dim3(grid_of(n, CT)), dim3(CT), 0, s, hs.p, gr.buckets[k], new_lo, n_new, cur.p, n_cur, lk_a.p, lv_a.p);
            SortArrays la, lb;
            la.a = lk_a.p; la.c = lv_a.p; lb.a = lk_b.p; lb.c = lv_b.p;
            int bits = 1;
            while ((1ull << bits) < gr.buckets[k]) bits++;
            if (radix_sort(s, la, lb, n, 0, bits, c->sort_scratch, &timing)) return fail(100, "radix sort failed");
            HIPQ(hipMemcpyAsync(cur.p, la.c, n * sizeof(u32), hipMemcpyDeviceToDevice, s));
            n_cur = (u32)n;
        }
        if (M) hipLaunchKernelGGL(k_take, dim3(grid_of(M, CT)), dim3(CT), 0, s, gen_grp.p, cur.p, M, grown_ord.p);
        HIPQ(hipStreamSynchronize(s));
        ord = grown_ord.p;
    }
    HIPQ(hipEventRecord(ev[4], s));

    // ---- sizes, places, bytes ----
    OutParams op;
    op.D = (u32)prm.digit_len; op.mapper = (u32)prm.mapper; op.n_contigs = dp.n_contigs; op.lrn_offset = prm.lrn_offset; op.ctg_gstart = dp.ctg_gstart;
    DBuf<u32> lrn_flag, sz_r, sz_p, sz_l;
    DBuf<u64> lrn_rank, off_r, off_p, off_l;
    RCQ(lrn_flag.alloc(M + 1)); RCQ(sz_r.alloc(M + 1)); RCQ(sz_p.alloc(M + 1)); RCQ(sz_l.alloc(M + 1));
    RCQ(lrn_rank.alloc(M + 1)); RCQ(off_r.alloc(M + 1)); RCQ(off_p.alloc(M + 1)); RCQ(off_l.alloc(M + 1));
    u64 totals[4] = {0, 0, 0, 0};
    c->n_r = c->n_p = c->n_l = 0;
    if (M) {
        const u32 gb = grid_of(M, CT);
        LAUNCH_W(k_out_flags, dim3(gb), dim3(CT), s, src, op, sid, gstart.p, ord, M, lrn_flag.p, ctr.p);
        RCQ(scan_u32(s, lrn_flag.p, lrn_rank.p, M, tot.p + 3));
        hipLaunchKernelGGL(k_out_sizes, dim3(gb), dim3(CT), 0, s, src, op, sid, gstart.p, ord, M, lrn_flag.p, lrn_rank.p, sz_r.p, sz_p.p, sz_l.p);
        RCQ(scan_u32(s, sz_r.p, off_r.p, M, tot.p));
        RCQ(scan_u32(s, sz_p.p, off_p.p, M, tot.p + 1));
        RCQ(scan_u32(s, sz_l.p, off_l.p, M, tot.p + 2));
        HIPQ(hipMemcpyAsync(totals, tot.p, 32, hipMemcpyDeviceToHost, s));
        HIPQ(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        if (hc.name_overflow) return fail(19, "a read-mer's name exceeds 4 GiB");
        DBuf<u8> d_r, d_p, d_l;
        RCQ(d_r.alloc(totals[0] + 16)); RCQ(d_p.alloc(totals[1] + 16)); RCQ(d_l.alloc(totals[2] + 16));
        LAUNCH_W(k_out_write, dim3(gb), dim3(CT), s, src, op, sid, gstart.p, ord, M, lrn_flag.p, lrn_rank.p, off_r.p, off_p.p, off_l.p,
                 sz_r.p, sz_p.p, sz_l.p, d_r.p, d_p.p, d_l.p);
        HIPQ(hipEventRecord(ev[5], s));
        RCQ(c->out_r.reserve(totals[0] + 1)); RCQ(c->out_p.reserve(totals[1] + 1)); RCQ(c->out_l.reserve(totals[2] + 1));
        if (totals[0]) HIPQ(hipMemcpyAsync(c->out_r.p, d_r.p, totals[0], hipMemcpyDeviceToHost, s));
        if (totals[1]) HIPQ(hipMemcpyAsync(c->out_p.p, d_p.p, totals[1], hipMemcpyDeviceToHost, s));
        if (totals[2]) HIPQ(hipMemcpyAsync(c->out_l.p, d_l.p, totals[2], hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        c->n_r = totals[0]; c->n_p = totals[1]; c->n_l = totals[2];
    } else {
        HIPQ(hipEventRecord(ev[5], s));
        HIPQ(hipStreamSynchronize(s));
    }
    st.lrn_names = totals[3];
    st.reads_bytes = c->n_r; st.perfect_bytes = c->n_p; st.lrn_bytes = c->n_l;
    HIPQ(hipEventElapsedTime(&st.ms_parse, ev[0], ev[1]));
    HIPQ(hipEventElapsedTime(&st.ms_keys, ev[1], ev[2]));
    HIPQ(hipEventElapsedTime(&st.ms_group, ev[2], ev[3]));
    HIPQ(hipEventElapsedTime(&st.ms_order, ev[3], ev[4]));
    HIPQ(hipEventElapsedTime(&st.ms_text, ev[4], ev[5]));
    HIPQ(hipEventElapsedTime(&st.ms_device, ev[0], ev[5]));
    st.ms_sort_scatter = timing.scatter_ms; st.n_sort_scatter = timing.scatter_launches; st.sort_scatter_bytes = timing.scatter_bytes;
    st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

}  // namespace

extern "C" {

int cora_hip_collapse_create(const cora_hip_collapse_params *p, cora_hip_collapse **out)
{
    if (!p || !out) return fail(10, "null argument");
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (p->splits < 1 || p->splits > 3) return fail(9, "Split mode unrecognized");                     // CL:584-588
    if (p->mapper < 0 || p->mapper > 3) return fail(19, "mapper mode out of range");
    if (p->digit_len < 2 || p->digit_len > 10) return fail(19, "digit length %d outside 2..10 (CL:32)", p->digit_len);
    // a line of 299 bases no longer fits the reference's fgets(.., 300) with its line end (CL:27, 923-935): its records fall apart there
    if (p->read_len < p->splits || p->read_len > 298) return fail(19, "read length %d outside %d..298 (CL:27)", p->read_len, p->splits);
    cora_hip_collapse *c = new cora_hip_collapse();
    c->prm = *p;
    c->R = (u32)(p->read_len / p->splits);                                                              // CL:566-583
    {
        // the table as the reference sets it up (CL:640-643: rehash(200000000)); CORA_COLLAPSE_TABLE_REQUEST makes it small, so
        // that tests reach the growth steps with a few thousand read-mers
        u64 request = 200000000ull;
        if (const char *e = getenv("CORA_COLLAPSE_TABLE_REQUEST")) { const long long v = atoll(e); if (v >= 1) request = (u64)v; }
        uint64_t at[MAX_GROWTH], bk[MAX_GROWTH];
        const int n = cora_hip_collapse_table_growth(request, 1ull << 32, at, bk, MAX_GROWTH);
        if (n < 2 || bk[0] < 2 || at[n - 1] < (1ull << 32)) { delete c; return fail(19, "cannot determine the hash table's growth (%d steps)", n); }
        c->growth.n = n;
        for (int k = 0; k < n; k++) { c->growth.at[k] = at[k]; c->growth.buckets[k] = bk[k]; }
        c->B = bk[0]; c->first_rehash_at = at[1];
    }
    memset(&c->st, 0, sizeof c->st);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(100, "hipStreamCreate failed"); }
    *out = c;
    return 0;
}

int cora_hip_collapse_set_reference(cora_hip_collapse *c, const cora_hip_fasta *fa)
{
    if (!c || !fa) return fail(50, "null argument");
    cora_hip_fasta_info fi;
    RCQ(cora_hip_fasta_get_info(fa, &fi));
    if (!c->prm.chr_short && fi.contigs > 250) return fail(19, "%llu contigs need the 2-byte layout (collapse_L; CL:62-77)", (unsigned long long)fi.contigs);
    if (fi.contigs > 65530) return fail(19, "too many contigs (%llu, CL:330)", (unsigned long long)fi.contigs);
    if (c->ref_job) { cora_hip_job_destroy(c->ref_job); c->ref_job = nullptr; }
    cora_hip_params jp;
    memset(&jp, 0, sizeof jp);
    jp.K = 33; jp.d = 1; jp.chr_short = 1; jp.line_width = 1; jp.n_groups = 1; jp.n_threads = 1;      // only the packing stage of the job is used
    memset(jp.sig2, 1, sizeof jp.sig2); memset(jp.sig4, 1, sizeof jp.sig4);
    RCQ(cora_hip_job_create(&jp, &c->ref_job));
    const int rc = cora_hip_job_set_reference_fasta(c->ref_job, fa);
    if (rc) { cora_hip_job_destroy(c->ref_job); c->ref_job = nullptr; return rc; }
    return 0;
}

// CL:414-462: an optional "N_" in front (the flag it sets is never read), then signals of `signal_len` characters, each followed
// by one separator character.  Reference windows are N-free, so only signals over A C G T can ever match one (CL:464-488).
int cora_hip_collapse_set_fragment(cora_hip_collapse *c, const char *signals, int32_t signal_len)
{
    if (!c || !signals) return fail(10, "null argument");
    if (signal_len != 2 && signal_len != 4) return fail(6, "fragSignalLen can only be 2 or 4");            // CL:557-561
    if ((int)c->R < signal_len) return fail(19, "read-mers of %u bases are shorter than the signal", c->R);
    const char *t = signals;
    size_t n = strlen(t);
    if (t[0] == 'N') { const size_t skip = n >= 2 ? 2 : n; t += skip; n -= skip; }
    const size_t step = (size_t)signal_len + 1;
    if (n % step) return fail(19, "fragmentation signals: not a whole number of signals (the reference asserts, CL:428, 452)");
    if (n / step > 1000) return fail(19, "more than 1000 fragmentation signals (CL:410)");
    FragFilter ff;
    memset(&ff, 0, sizeof ff);
    ff.len = (u32)signal_len;
    for (size_t i = 0; i < n / step; i++) {
        u32 v = 0;
        bool acgt = true;
        for (int q = 0; q < signal_len; q++) {
            const char ch = t[i * step + q];
            const int code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
            if (code < 0) acgt = false; else v = v * 4 + (u32)code;
        }
        if (acgt) ff.mask[v >> 5] |= 1u << (v & 31);
    }
    c->frag = ff;
    c->fragmented = true;
    c->files.clear(); c->h_rd.clear();          // a new group: its own input (the reference and the device pool stay)
    c->has_dev_in = false; c->dev_in = nullptr; c->dev_in_bytes = 0;
    return 0;
}

int cora_hip_collapse_set_lrn_offset(cora_hip_collapse *c, uint64_t lrn_offset)
{
    if (!c) return fail(10, "null argument");
    c->prm.lrn_offset = lrn_offset;
    return 0;
}

int cora_hip_collapse_add_reads(cora_hip_collapse *c, const uint8_t *fastq, size_t bytes)
{
    if (!c || (!fastq && bytes)) return fail(5, "null argument");
    ReadFile f;
    f.off = (c->h_rd.size() + 15) & ~(size_t)15;
    f.bytes = bytes;
    c->h_rd.resize(f.off + bytes);
    if (bytes) memcpy(c->h_rd.data() + f.off, fastq, bytes);
    c->files.push_back(f);
    return 0;
}

int cora_hip_collapse_add_reads_device(cora_hip_collapse *c, const void *fragment_dev, size_t bytes)
{
    if (!c || (!fragment_dev && bytes)) return fail(10, "null argument");
    if (!c->fragmented) return fail(19, "a device-resident input needs the fragmented mode (cora_hip_collapse_set_fragment first)");
    if (!c->files.empty()) return fail(19, "the fragmented mode reads one fragment file; one was already added");
    c->has_dev_in = true;
    c->dev_in = (const u8 *)fragment_dev;
    c->dev_in_bytes = bytes;
    return 0;
}

int cora_hip_collapse_run(cora_hip_collapse *c)
{
    if (!c) return fail(10, "null argument");
    PoolScope scope(&c->pool);
    return run_impl(c);
}

int cora_hip_collapse_outputs(const cora_hip_collapse *c, const uint8_t **reads, size_t *reads_bytes, const uint8_t **perfect,
                              size_t *perfect_bytes, const uint8_t **lrn, size_t *lrn_bytes)
{
    if (!c) return fail(10, "null argument");
    if (reads) *reads = c->out_r.p; if (reads_bytes) *reads_bytes = c->n_r;
    if (perfect) *perfect = c->out_p.p; if (perfect_bytes) *perfect_bytes = c->n_p;
    if (lrn) *lrn = c->out_l.p; if (lrn_bytes) *lrn_bytes = c->n_l;
    return 0;
}

int cora_hip_collapse_get_stats(const cora_hip_collapse *c, cora_hip_collapse_stats *out)
{
    if (!c || !out) return fail(10, "null argument");
    *out = c->st;
    return 0;
}

void cora_hip_collapse_destroy(cora_hip_collapse *c) { delete c; }

}  // extern "C"
