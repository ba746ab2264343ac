// kernels.hip -- the homology-table pipeline on gfx950: 2-bit pack straight from the FASTA
// file image, canonical K-mer keys, class/representative detection over the sorted keys,
// pigeonhole seed codes, equal-run pair verification by XOR + popcount on the packed genome,
// directed items and diagonal block merging.  Integer work only; HBM / L2 bound.
// HTS = /root/reference/homTable_setup.cpp (cited per kernel for the behaviour restated).
#include "kernels.hpp"

namespace cora {

constexpr u64 M55 = 0x5555555555555555ull;

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ u32 find_contig(const DevParams &p, u64 g)
{
    // largest c with ctg_gstart[c] <= g   (ctg_gstart[n_contigs] = G)
    u32 lo = 0, hi = p.n_contigs;
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (p.ctg_gstart[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void load_window(const DevParams &p, const u64 *__restrict__ pk, u64 g, u64 &hi, u64 &lo)
{
    const u64 w = g >> 5;
    const int sh = (int)(g & 31) * 2;
    const u64 x0 = pk[w], x1 = pk[w + 1], x2 = pk[w + 2];
    if (sh) { hi = (x0 << sh) | (x1 >> (64 - sh)); lo = (x1 << sh) | (x2 >> (64 - sh)); }
    else { hi = x0; lo = x1; }
    lo &= p.kmask_lo;
}

__device__ __forceinline__ u64 rev2(u64 x)
{
    x = __brevll(x);
    return ((x >> 1) & M55) | ((x & M55) << 1);
}

// reverse complement of a left-aligned K-mer (HTS:1941-1951 restated on two words)
__device__ __forceinline__ void rc_window(int K, u64 hi, u64 lo, u64 &rhi, u64 &rlo)
{
    const u64 H = ~rev2(lo), L = ~rev2(hi);
    const int s = 2 * (64 - K);
    if (s) { rhi = (H << s) | (L >> (64 - s)); rlo = L << s; }
    else { rhi = H; rlo = L; }
}

__device__ __forceinline__ u32 base_at(u64 hi, u64 lo, int i)
{
    return (u32)((i < 32 ? hi >> (62 - 2 * i) : lo >> (62 - 2 * (i - 32))) & 3u);
}

__device__ __forceinline__ bool bad_bit(const u32 *__restrict__ bad, u64 g)
{
    return (bad[g >> 5] >> (31 - (u32)(g & 31))) & 1u;
}

// Is there a K-mer window at g that the reference's scan would visit?
// N-free window (HTS:2619-2636) minus the Q4 case: GetFirstNonNreadmer (HTS:1954-1981)
// never *finds* a window ending on the contig's last base, so a final N-free stretch of
// exactly K bases yields no window.
__device__ __forceinline__ bool window_ok(const DevParams &p, const u32 *__restrict__ bad, u64 g)
{
    if (g + (u64)p.K > p.G) return false;
    const u64 w = g >> 5;
    const int o = (int)(g & 31);
    u64 x = ((u64)bad[w] << 32) | bad[w + 1];
    if (o) x = (x << o) | (u64)(bad[w + 2] >> (32 - o));
    if (x >> (64 - p.K)) return false;
    if (bad_bit(bad, g + p.K)) {
        const u32 c = find_contig(p, g);
        if (g + p.K == p.ctg_gstart[c] + p.ctg_len[c]) {
            if (g == p.ctg_gstart[c]) return false;
            if (bad_bit(bad, g - 1)) return false;
        }
    }
    return true;
}

__device__ __forceinline__ u32 block_excl_scan_u32(u32 v, u32 *lds4, u32 *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds4[wv] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { u32 s = lds4[w]; if (w < wv) base += s; tot += s; }
    if (total) *total = tot;
    __syncthreads();
    return base + x - v;
}

// ------------------------------------------------------------------ pack
// One thread packs 32 bases of the global coordinate into one u64 (base i at bits 62-2i)
// and one validity word.  Bases are read from the FASTA image: base `local` of a contig
// sits at ctg_off + local + local / line_width (HTS:96-142: fixed line width, '\n' ends).
// Upper-casing and "everything but ACGT is N" follow HTS:131-136.
//
// Fast path (all 32 bases inside the contig and not its last ones, lines of at least 33 bases, nothing but ACGTN in
// either case and the line end where it belongs): the 33 bytes come in as three aligned 16-byte loads, the line end is
// squeezed out, and letters become codes eight at a time: for A C G T, ((ch >> 1) ^ (ch >> 2)) & 3 = 0 1 2 3, and a byte
// is a base iff it equals the letter its own code stands for.  Anything else takes the byte loop below, which is the
// definition.
constexpr u64 B01 = 0x0101010101010101ull, B7F = 0x7F7F7F7F7F7F7F7Full, B80 = 0x8080808080808080ull;
__device__ __forceinline__ u64 nonzero_bytes(u64 x) { return (((x & B7F) + B7F) | x) & B80; }      // 0x80 in every byte of x that is not 0

// eight letters (byte 0 = first base): codes into bits 15..0 (first base highest), not-ACGT flags into bits 7..0 alike;
// false when some byte is neither ACGT nor N (in either case)
__device__ __forceinline__ bool pack8(u64 c, u32 &codes, u32 &bads)
{
    const u64 u = c & 0xDFDFDFDFDFDFDFDFull;
    const u64 c0 = ((u >> 1) ^ (u >> 2)) & B01, c1 = ((u >> 2) ^ (u >> 3)) & B01, cc = c0 & c1;
    const u64 expect = 0x4141414141414141ull + (c0 << 1) + (c1 << 2) + (c1 << 1) + (cc << 3) + (cc << 1) + cc;   // A +2 C, +6 G, +19 T
    const u64 bad = nonzero_bytes(u ^ expect);
    const u64 not_n = nonzero_bytes(u ^ 0x4E4E4E4E4E4E4E4Eull);
    if (bad & not_n) return false;
    u64 y = __builtin_bswap64(c0 | (c1 << 1));               // first base in the top byte
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    codes = (u32)((y | (y >> 24)) & 0xFFFFull);
    u64 z = __builtin_bswap64(bad >> 7);
    z = (z | (z >> 7)) & 0x0003000300030003ull;
    z = (z | (z >> 14)) & 0x0000000F0000000Full;
    bads = (u32)((z | (z >> 28)) & 0xFFull);
    return true;
}

__device__ __forceinline__ bool pack32_fast(const DevParams &p, const u8 *__restrict__ raw, u64 a0, u32 n /* bases before the line end */,
                                            u64 &packed, u32 &badw)
{
    const uint4 *q = reinterpret_cast<const uint4 *>(raw + (a0 & ~15ull));
    const uint4 v0 = q[0], v1 = q[1], v2 = q[2];
    u64 w[7] = {((u64)v0.y << 32) | v0.x, ((u64)v0.w << 32) | v0.z, ((u64)v1.y << 32) | v1.x, ((u64)v1.w << 32) | v1.z,
                ((u64)v2.y << 32) | v2.x, ((u64)v2.w << 32) | v2.z, 0ull};
    if (a0 & 8) {
#pragma unroll
        for (int j = 0; j < 6; j++) w[j] = w[j + 1];
    }
    const int sh = (int)(a0 & 7) * 8;
    u64 b[5];                                               // bytes a0 .. a0+39
#pragma unroll
    for (int j = 0; j < 5; j++) b[j] = sh ? ((w[j] >> sh) | (w[j + 1] << (64 - sh))) : w[j];
    if (n <= 32) {                                          // the line ends inside (or right behind) these bases: byte n must be it
        u64 wn = b[0];
#pragma unroll
        for (int j = 1; j < 5; j++) if ((int)(n >> 3) == j) wn = b[j];
        if (((wn >> ((n & 7) * 8)) & 0xFF) != (u64)'\n') return false;
    }
    packed = 0; badw = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u64 c = b[j];
        if (n < 8u * (j + 1)) {                             // bytes from n on come from one place further on
            const u64 d = (b[j] >> 8) | (b[j + 1] << 56);
            if (n <= 8u * j) c = d;
            else { const u64 m = (1ull << ((n - 8u * j) * 8)) - 1ull; c = (b[j] & m) | (d & ~m); }
        }
        u32 codes, bads;
        if (!pack8(c, codes, bads)) return false;
        packed |= (u64)codes << (48 - 16 * j);
        badw |= bads << (24 - 8 * j);
    }
    return true;
}

__global__ __launch_bounds__(256) void k_pack(DevParams p, const u8 *__restrict__ raw, u64 *__restrict__ pk,
                                              u32 *__restrict__ bad, DevCounters *ctr)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= p.n_words + 4) return;
    u64 packed = 0;
    u32 badw = 0xFFFFFFFFu;
    if (w < p.n_words) {
        const u64 g0 = w << 5;
        const u32 c = find_contig(p, g0);
        const u64 local0 = g0 - p.ctg_gstart[c];
        const u32 len = p.ctg_len[c];
        if (local0 < len) {
            const u64 off0 = p.ctg_off[c];
            u64 line = local0 / p.line_width;
            u32 col = (u32)(local0 - line * p.line_width);
            const u64 a0 = off0 + local0 + line;
            bool done = false;
            if (p.line_width >= 33 && local0 + 32 < len && a0 + 48 <= p.image_bytes + 16 && a0 + 33 <= p.image_bytes)
                done = pack32_fast(p, raw, a0, p.line_width - col, packed, badw);
            if (!done) {
                u32 errs = 0, others = 0;
                packed = 0;
                badw = 0;
#pragma unroll 4
                for (int i = 0; i < 32; i++) {
                    const u64 local = local0 + i;
                    u32 code = 0, isbad = 1;
                    if (local < len) {
                        const u64 at = off0 + local + line;
                        u32 ch = (at < p.image_bytes) ? raw[at] : (u32)'\n';
                        if (ch == '\n' || ch == '>') errs++;
                        if (ch > 'Z') ch -= 32;
                        isbad = 0;
                        if (ch == 'A') code = 0; else if (ch == 'C') code = 1; else if (ch == 'G') code = 2; else if (ch == 'T') code = 3;
                        else { isbad = 1; others += (ch != 'N'); }
                        if (col + 1 == p.line_width || local + 1 == len) {      // a line must end here
                            const u64 nl = at + 1;
                            if (nl < p.image_bytes && raw[nl] != '\n') errs++;
                        }
                        if (++col == p.line_width) { col = 0; line++; }
                    }
                    packed |= (u64)code << (62 - 2 * i);
                    badw |= isbad << (31 - i);
                }
                if (errs) atomicAdd(&ctr->format_errors, (unsigned long long)errs);
                if (others) atomicAdd(&ctr->other_letters, (unsigned long long)others);
            }
        }
    }
    pk[w] = packed;
    bad[w] = badw;
}

// ------------------------------------------------------------------ valid windows -> keys
// One thread per word of 32 positions: the validity bits of positions 32w .. 32w+127 as a 128-bit string (first position
// in the top bit), "some bad base in [i, i+K)" for all 32 starts i at once by doubling (OR of the string shifted by 1, 2,
// 4, 8, 16 and K-32: six steps instead of 32 window tests of three loads each), and the Q4 case (window_ok) for the few
// starts whose window is followed by a bad base.  Per block of 256 positions (8 words = 8 neighbouring lanes): the count,
// as the record emission wants it.  window_ok() stays the definition; tests compare the two (cora_hip_test_window_bits).
__device__ __forceinline__ void shl128(u64 &hi, u64 &lo, int s /* 1..63 */)
{
    hi = (hi << s) | (lo >> (64 - s));
    lo <<= s;
}
__device__ __forceinline__ u32 window_bits(const DevParams &p, const u32 *__restrict__ bad, u64 w)
{
    if (w >= p.n_words) return 0;                       // (bad[] has 4 words of padding behind n_words, all ones)
    const u64 x_hi = ((u64)bad[w] << 32) | bad[w + 1], x_lo = ((u64)bad[w + 2] << 32) | bad[w + 3];
    u64 hi = x_hi, lo = x_lo;
#pragma unroll
    for (int sft = 1; sft <= 16; sft <<= 1) { u64 h = hi, l = lo; shl128(h, l, sft); hi |= h; lo |= l; }   // any bad in [i, i+32)
    if (p.K > 32) { u64 h = hi, l = lo; shl128(h, l, p.K - 32); hi |= h; }                                 // ... in [i, i+K)
    u32 ok = ~(u32)(hi >> 32);
    // Q4 (HTS:1954-1981): the base behind the window is bad -> window_ok decides (a handful of positions per N run)
    u64 nh = x_hi, nl = x_lo;
    if (p.K < 64) shl128(nh, nl, p.K); else nh = nl;
    u32 q = ok & (u32)(nh >> 32);
    while (q) {
        const int i = __clz((int)q);
        q &= ~(0x80000000u >> i);
        if (!window_ok(p, bad, (w << 5) + (u64)i)) ok &= ~(0x80000000u >> i);
    }
    return ok;
}

__global__ __launch_bounds__(256) void k_kmer_count(DevParams p, const u32 *__restrict__ bad, u32 *__restrict__ block_counts,
                                                    u32 *__restrict__ valid, u32 n_blocks)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;            // word t of the range: 8 words per block of 256 positions
    const bool in = t < (u64)n_blocks * 8;
    const u64 w = (p.pos_base >> 5) + t;
    const u32 ok = in ? window_bits(p, bad, w) : 0u;
    // the window-validity bitmap (bit 31-i of word w = position 32w+i), kept for the later passes over positions
    if (in && valid) valid[w] = ok;
    u32 c = (u32)__popc(ok);
    c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
    if (in && (threadIdx.x & 7) == 0) block_counts[t >> 3] = c;
}

// test hook: the validity word of every 32 positions by the bit-parallel path and by window_ok, one thread per word
__global__ __launch_bounds__(256) void k_test_window_bits(DevParams p, const u32 *__restrict__ bad, u32 *__restrict__ fast, u32 *__restrict__ slow)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= p.n_words) return;
    fast[w] = window_bits(p, bad, w);
    u32 r = 0;
    for (int i = 0; i < 32; i++) if (window_ok(p, bad, (w << 5) + (u64)i)) r |= 0x80000000u >> i;
    slow[w] = r;
}

// canonical key = lexicographically smaller of forward / reverse complement (HTS:2551-2565)
__global__ __launch_bounds__(256) void k_kmer_emit(DevParams p, const u64 *__restrict__ pk, const u32 *__restrict__ bad,
                                                   const u64 *__restrict__ block_off, u64 *__restrict__ key_hi,
                                                   u64 *__restrict__ key_lo, u32 *__restrict__ gpos)
{
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 g = p.pos_base + (u64)blockIdx.x * 256 + threadIdx.x;
    const bool ok = (g < p.G) && window_ok(p, bad, g);
    const u64 bal = __ballot(ok);
    if (lane == 0) wsum[wv] = (u32)__popcll(bal);
    __syncthreads();
    if (!ok) return;
    u32 pre = 0;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    const u64 at = block_off[blockIdx.x] + pre + (u32)__popcll(bal & ((1ull << lane) - 1ull));
    u64 hi, lo, rhi, rlo;
    load_window(p, pk, g, hi, lo);
    rc_window(p.K, hi, lo, rhi, rlo);
    const bool use_rc = (rhi < hi) || (rhi == hi && rlo < lo);
    key_hi[at] = use_rc ? rhi : hi;
    key_lo[at] = use_rc ? rlo : lo;
    gpos[at] = (u32)g;
}

// Record flags carried in the low 8 bits of a sort record (the hash occupies the bits above):
//   bit 0 strand (canonical form is the reverse complement, HTS:2563), bit 1 palindrome (fw == rc),
//   bits 2-5 physical group of the window's (first,last) base pair (HTS:2549).
__device__ __forceinline__ u32 kmer_flags(const DevParams &p, u64 hi, u64 lo, u64 rhi, u64 rlo)
{
    const bool use_rc = (rhi < hi) || (rhi == hi && rlo < lo);
    const bool pal = (rhi == hi) && (rlo == lo);
    const u32 grp = p.sig2[base_at(hi, lo, 0) * 4 + base_at(hi, lo, p.K - 1)];
    return (use_rc ? 1u : 0u) | (pal ? 2u : 0u) | (grp << 2);
}

// splitmix64 finaliser: full avalanche in both directions.  (A first version combined the two key words
// by multiplications only; a multiplication never moves a difference DOWN, so K-mers of one repeat family
// that differed in the top bits of both words collided a thousand times more often than chance.)
__device__ __forceinline__ u64 mix64(u64 z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ u64 key_hash(u64 hi, u64 lo)
{
    return mix64(mix64(hi) + lo * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
}

__device__ __forceinline__ void canonical_at(const DevParams &p, const u64 *__restrict__ pk, u64 g, u64 &chi, u64 &clo)
{
    u64 hi, lo, rhi, rlo;
    load_window(p, pk, g, hi, lo);
    rc_window(p.K, hi, lo, rhi, rlo);
    const bool use_rc = (rhi < hi) || (rhi == hi && rlo < lo);
    chi = use_rc ? rhi : hi; clo = use_rc ? rlo : lo;
}

// Hashed record of every valid window: rec = (hash of the canonical key & ~0xFF) | flags, plus gpos.
// Sorting 12-byte records on the top 48/56 hash bits replaces the 13-pass sort of 20-byte full keys;
// equal keys have equal hashes, and k_same_detect / k_mixed_fix make the grouping exact.
__global__ __launch_bounds__(256) void k_kmer_emit_hashed(DevParams p, const u64 *__restrict__ pk, const u32 *__restrict__ valid,
                                                          const u64 *__restrict__ block_off, u64 *__restrict__ rec,
                                                          u32 *__restrict__ gpos, unsigned char *__restrict__ first_digit /* null, or: byte fd_shift / 8
                                                          of every record, for the first histogram of its sort (radix_sort.hip, SortScratch::next_digit) */,
                                                          int fd_shift)
{
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 g = p.pos_base + (u64)blockIdx.x * 256 + threadIdx.x;
    const bool ok = (valid[g >> 5] >> (31 - (g & 31))) & 1u;          // written by k_kmer_count over the same blocks
    const u64 bal = __ballot(ok);
    if (lane == 0) wsum[wv] = (u32)__popcll(bal);
    __syncthreads();
    if (!ok) return;
    u32 pre = 0;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    const u64 at = block_off[blockIdx.x] + pre + (u32)__popcll(bal & ((1ull << lane) - 1ull));
    u64 hi, lo, rhi, rlo;
    load_window(p, pk, g, hi, lo);
    rc_window(p.K, hi, lo, rhi, rlo);
    const u32 fl = kmer_flags(p, hi, lo, rhi, rlo);
    const u64 h = (fl & 1u) ? key_hash(rhi, rlo) : key_hash(hi, lo);
    rec[at] = (h & ~0xFFull) | fl;
    gpos[at] = (u32)g;
    if (first_digit) first_digit[at] = (unsigned char)(h >> fd_shift);
}

// After the sort on the top `hb` hash bits: head_flag[i] = 1 when record i opens a class.  Records with
// equal truncated hashes have their canonical keys re-read from the packed genome and compared; a key
// SMALLER than its predecessor's inside an equal-hash run means two different keys collided and are not
// grouped yet: such positions are marked in mixed_mask (one ballot word per wave, like the heads; the list k_mixed_fix
// wants comes from k_mask_count / k_mask_emit.  A shared counter here cost 12 ns per listed place: 23 ms for the 2e6
// places of a 3e9-record sort on 40 hash bits).
// Each wave takes 256 consecutive records in four rounds of 64 (round k, lane l: record base + 64 k + l), all loads of the
// four rounds in flight together: with one record per thread the kernel was bound by latency x occupancy (12.8 ms for 35 GB
// at 2.9e9 records).  A round's ballot is one mask word.
__global__ __launch_bounds__(256) void k_same_detect(DevParams p, const u64 *__restrict__ pk, const u64 *__restrict__ rec,
                                                     const u32 *__restrict__ gpos, u64 n, int hb, u64 *__restrict__ head_mask,
                                                     u64 *__restrict__ mixed_mask)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 wbase = ((u64)blockIdx.x * 4 + wv) * 256;
    const u64 n_pad = ((n + 255) / 256) * 256;               // the masks are sized (and read) in blocks of 256 records
    if (wbase >= n_pad) return;                               // (wave-uniform)
    u64 r[4];
    u32 g[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = wbase + (u64)k * 64 + lane;
        r[k] = i < n ? __builtin_nontemporal_load(rec + i) : 0ull;
        g[k] = i < n ? gpos[i] : 0u;
    }
    u64 r_before = 0;                                         // the record before this wave's first
    u32 g_before = 0;
    if (lane == 0 && wbase > 0 && wbase < n) { r_before = rec[wbase - 1]; g_before = gpos[wbase - 1]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = wbase + (u64)k * 64 + lane;
        u64 rp = __shfl_up(r[k], 1, 64);
        u32 gp = __shfl_up(g[k], 1, 64);
        const u64 r_tail = k ? __shfl(r[k ? k - 1 : 0], 63, 64) : r_before;
        const u32 g_tail = k ? __shfl(g[k ? k - 1 : 0], 63, 64) : g_before;
        if (lane == 0) { rp = r_tail; gp = g_tail; }
        u32 head = i < n ? 1u : 0u;
        bool mixed = false;
        if (i < n && i > 0 && (r[k] >> (64 - hb)) == (rp >> (64 - hb))) {
            u64 ah, al, bh, bl;
            canonical_at(p, pk, gp, ah, al);
            canonical_at(p, pk, g[k], bh, bl);
            if (ah == bh && al == bl) head = 0;
            else mixed = bh < ah || (bh == ah && bl < al);
        }
        const u64 bal = __ballot(head != 0);                  // bit l of word i/64 = record 64*(i/64)+l opens a class
        const u64 bam = __ballot(mixed);
        if (lane == 0 && wbase + (u64)k * 64 < n_pad) { head_mask[i >> 6] = bal; mixed_mask[i >> 6] = bam; }
    }
}

// One thread per listed position.  The run of equal truncated hashes around it is found by binary search (the records
// are sorted on those bits); the first thread to claim the run's start repairs it: a run of at most 64 records is
// sorted by canonical key in registers (stable, so equal keys stay in position order), a longer one -- a K-mer with
// many copies that collided with another K-mer -- is listed for k_mixed_fix_long.
__global__ __launch_bounds__(64) void k_mixed_fix(DevParams p, const u64 *__restrict__ pk, u64 *__restrict__ rec,
                                                  u32 *__restrict__ gpos, u64 n, int hb, const u32 *__restrict__ mixed_list,
                                                  u32 n_mixed, u64 *head_mask, u32 *__restrict__ claim,
                                                  u64 *__restrict__ long_runs, u32 long_cap, DevCounters *ctr)
{
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_mixed) return;
    const u64 i = mixed_list[t];
    const int sh = 64 - hb;
    const u64 tr = rec[i] >> sh;
    u64 lo = 0, hi = i;                                    // first record of the run: lowest index whose bits are >= tr
    while (lo < hi) { const u64 m = (lo + hi) >> 1; if ((rec[m] >> sh) < tr) lo = m + 1; else hi = m; }
    const u64 s = lo;
    lo = i + 1; hi = n;                                    // one past its last record
    while (lo < hi) { const u64 m = (lo + hi) >> 1; if ((rec[m] >> sh) <= tr) lo = m + 1; else hi = m; }
    const u64 e = lo;
    const u32 bit = 1u << (s & 31);
    if (atomicOr(&claim[s >> 5], bit) & bit) return;       // another listed position of this run got here first
    if (e - s > 64) {
        const unsigned long long at = atomicAdd(&ctr->n_long, 1ull);
        if (at < long_cap) { long_runs[2 * at] = s; long_runs[2 * at + 1] = e; }
        else ctr->fallback = 1;
        return;
    }
    const int L = (int)(e - s);
    u64 r_[64], kh[64], kl[64];
    u32 g_[64];
    for (int k = 0; k < L; k++) { r_[k] = rec[s + k]; g_[k] = gpos[s + k]; canonical_at(p, pk, g_[k], kh[k], kl[k]); }
    for (int a = 1; a < L; a++) {                         // insertion sort: stable
        const u64 r0 = r_[a], h0 = kh[a], l0 = kl[a];
        const u32 g0 = g_[a];
        int b = a - 1;
        while (b >= 0 && (kh[b] > h0 || (kh[b] == h0 && kl[b] > l0))) { r_[b + 1] = r_[b]; g_[b + 1] = g_[b]; kh[b + 1] = kh[b]; kl[b + 1] = kl[b]; b--; }
        r_[b + 1] = r0; g_[b + 1] = g0; kh[b + 1] = h0; kl[b + 1] = l0;
    }
    for (int k = 0; k < L; k++) {                         // the run in key order, and its class heads
        rec[s + k] = r_[k]; gpos[s + k] = g_[k];
        const u64 hbit = 1ull << ((s + k) & 63);          // other runs may share the word
        if (k == 0 || kh[k] != kh[k - 1] || kl[k] != kl[k - 1]) atomicOr((unsigned long long *)&head_mask[(s + k) >> 6], hbit);
        else atomicAnd((unsigned long long *)&head_mask[(s + k) >> 6], ~hbit);
    }
}

// One workgroup per long run: it holds few distinct K-mers (one with many copies plus whatever collided with it), so
// the run is copied aside and rebuilt key by key in ascending key order, each key's records compacted in their
// original (= position) order.  A run with more than MIXED_MAX_KEYS distinct keys (only when tests shorten the hash)
// is put back as it was and the fallback flag is set.
constexpr int MIXED_MAX_KEYS = 8;
__global__ __launch_bounds__(256) void k_mixed_fix_long(DevParams p, const u64 *__restrict__ pk, u64 *__restrict__ rec, u32 *__restrict__ gpos,
                                                        u64 *__restrict__ rec_copy, u32 *__restrict__ gpos_copy,
                                                        const u64 *__restrict__ long_runs, u64 *head_mask, DevCounters *ctr)
{
    __shared__ u64 s_h[256], s_l[256];
    __shared__ u32 s_cnt[4];
    const u64 s = long_runs[2 * blockIdx.x], e = long_runs[2 * blockIdx.x + 1];
    const u64 L = e - s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (u64 k = tid; k < L; k += 256) { rec_copy[s + k] = rec[s + k]; gpos_copy[s + k] = gpos[s + k]; }
    __syncthreads();
    u64 last_h = 0, last_l = 0, out = 0;
    bool have_last = false, gave_up = false;
    for (int pass = 0; out < L; pass++) {
        if (pass == MIXED_MAX_KEYS) { gave_up = true; break; }
        u64 mh = ~0ull, ml = ~0ull;                        // smallest key above the last one (all ones is no canonical key: T^K's is A^K)
        bool any = false;
        for (u64 k = tid; k < L; k += 256) {
            u64 kh, kl;
            canonical_at(p, pk, gpos_copy[s + k], kh, kl);
            const bool above = !have_last || kh > last_h || (kh == last_h && kl > last_l);
            if (above && (!any || kh < mh || (kh == mh && kl < ml))) { mh = kh; ml = kl; any = true; }
        }
        s_h[tid] = mh; s_l[tid] = ml;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const u64 oh = s_h[tid + o], ol = s_l[tid + o];
                if (oh < s_h[tid] || (oh == s_h[tid] && ol < s_l[tid])) { s_h[tid] = oh; s_l[tid] = ol; }
            }
            __syncthreads();
        }
        mh = s_h[0]; ml = s_l[0];                          // some record is above `last` while out < L, so this is a real key
        __syncthreads();
        const u64 group_start = out;
        for (u64 base = 0; base < L; base += 256) {
            const u64 k = base + tid;
            bool f = false;
            if (k < L) { u64 kh, kl; canonical_at(p, pk, gpos_copy[s + k], kh, kl); f = (kh == mh && kl == ml); }
            const u64 bal = __ballot(f);
            if (lane == 0) s_cnt[wv] = (u32)__popcll(bal);
            __syncthreads();
            u32 pre = 0;
            for (int w = 0; w < wv; w++) pre += s_cnt[w];
            const u32 total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            if (f) {
                const u64 at = out + pre + (u32)__popcll(bal & ((1ull << lane) - 1ull));
                rec[s + at] = rec_copy[s + k]; gpos[s + at] = gpos_copy[s + k];
                const u64 hbit = 1ull << ((s + at) & 63);
                if (at == group_start) atomicOr((unsigned long long *)&head_mask[(s + at) >> 6], hbit);
                else atomicAnd((unsigned long long *)&head_mask[(s + at) >> 6], ~hbit);
            }
            out += total;
            __syncthreads();
        }
        last_h = mh; last_l = ml; have_last = true;
    }
    if (gave_up) {
        for (u64 k = tid; k < L; k += 256) { rec[s + k] = rec_copy[s + k]; gpos[s + k] = gpos_copy[s + k]; }
        if (tid == 0) ctr->fallback = 1;
    }
}

// Full-key path (used when k_mixed_fix gives up): class heads from the sorted 128-bit keys, and the
// flag byte rebuilt from the genome.
__global__ __launch_bounds__(256) void k_heads_from_keys(DevParams p, const u64 *__restrict__ pk, const u64 *__restrict__ hi,
                                                         const u64 *__restrict__ lo, const u32 *__restrict__ gpos, u64 n,
                                                         u64 *__restrict__ head_mask, u64 *__restrict__ rec)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool head = i < n && ((i == 0) || hi[i - 1] != hi[i] || lo[i - 1] != lo[i]);
    const u64 bal = __ballot(head);
    if ((threadIdx.x & 63) == 0) head_mask[i >> 6] = bal;
    if (i >= n) return;
    u64 fh, fl, rh, rl;
    load_window(p, pk, gpos[i], fh, fl);
    rc_window(p.K, fh, fl, rh, rl);
    rec[i] = kmer_flags(p, fh, fl, rh, rl);
}

// a contig without any window makes the reference read before its buffer (HTS:2530): count them
__global__ __launch_bounds__(256) void k_contig_check(DevParams p, const u32 *__restrict__ gpos, u64 n, DevCounters *ctr)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.n_contigs) return;
    const u64 lo_g = p.ctg_gstart[c], hi_g = lo_g + p.ctg_len[c];
    u64 a = 0, b = n;
    while (a < b) {
        const u64 mid = (a + b) >> 1;
        if (gpos[mid] < lo_g) a = mid + 1; else b = mid;
    }
    if (a >= n || gpos[a] >= hi_g) atomicAdd(&ctr->empty_contigs, 1ull);
}

// the same check straight from the validity bitmap, one block per contig (sharded runs: no rank holds every window)
__global__ __launch_bounds__(256) void k_contig_scan(DevParams p, const u32 *__restrict__ bad, DevCounters *ctr)
{
    const u32 c = blockIdx.x;
    const u64 lo_g = p.ctg_gstart[c], len = p.ctg_len[c];
    for (u64 base = 0; base < len; base += 256) {
        const u64 o = base + threadIdx.x;
        const bool ok = (o < len) && window_ok(p, bad, lo_g + o);
        if (__syncthreads_or(ok)) return;
    }
    if (threadIdx.x == 0) atomicAdd(&ctr->empty_contigs, 1ull);
}

// ------------------------------------------------------------------ classes over the sorted keys
__device__ __forceinline__ u32 window_group(const DevParams &p, const u64 *__restrict__ pk, u64 g)
{
    u64 hi, lo;
    load_window(p, pk, g, hi, lo);
    return p.sig2[base_at(hi, lo, 0) * 4 + base_at(hi, lo, p.K - 1)];     // HTS:2549
}

__device__ __forceinline__ bool key_is_palindrome(int K, u64 hi, u64 lo)
{
    if (K & 1) return false;
    u64 rhi, rlo;
    rc_window(K, hi, lo, rhi, rlo);
    return rhi == hi && rlo == lo;
}

// One thread per sorted record; head_mask bit i = record i opens a class.  Everything is read in record order (coalesced):
// a class is a singleton exactly when the next record opens a class too, so no class-start array is needed.
// Per block of 256 records: the number of classes with >= 2 entries (>= 2 occurrences, or a palindrome: Q2) that open in
// it and the smallest (group, position) of a singleton class.  The latter locates the first singleton in the reference's
// class-id order: classes created before it escape the direction flip (Q3, HTS:421-433).  Per-block partials instead of
// global counters: same-address atomics cost about 12 ns each chip-wide.  nons_mask: bit i = record i opens a class with
// >= 2 entries (one ballot word per wave).  Later members of classes (records that open nothing) are marked in nonrep:
// representatives = first (lowest) position of every class (HTS:2573-2578) = every valid window that is not marked; later
// members are few (the repeated K-mers), so they are what gets marked, by atomics, in a zeroed bitmap.
// (Round 2 first ran this per CLASS over an array of class starts: start -> record -> position was a chain of dependent
//  gathers, 14.9 ms at 2.6e9 classes, plus 5.4 ms for marking the later members and 1.9 ms for listing the starts.)
// Each wave takes 256 consecutive records in four rounds of 64, all loads in flight together (as k_same_detect): one wave =
// one entry of blk_nons / blk_min, one round = one mask word.
__global__ __launch_bounds__(256) void k_class_scan(const u64 *__restrict__ rec, const u32 *__restrict__ gpos,
                                                    const u64 *__restrict__ head_mask, u64 n, u32 *__restrict__ blk_nons,
                                                    unsigned long long *__restrict__ blk_min, u64 *__restrict__ nons_mask,
                                                    u32 *__restrict__ nonrep)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 wid = (u64)blockIdx.x * 4 + wv;                 // this wave's block of 256 records
    const u64 wbase = wid * 256;
    if (wbase >= n) return;                                    // (wave-uniform; the masks and partials cover ceil(n / 256) blocks)
    u64 r[4], hw[5];
    u32 g[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = wbase + (u64)k * 64 + lane;
        r[k] = i < n ? __builtin_nontemporal_load(rec + i) : 0ull;
        g[k] = i < n ? __builtin_nontemporal_load(gpos + i) : 0u;
        hw[k] = head_mask[(wbase >> 6) + k];                   // this round's 64 head bits
    }
    hw[4] = (wbase + 256 < n) ? head_mask[(wbase >> 6) + 4] : 1ull;      // does the record after the wave's last open a class?  (past the end: yes)
    unsigned long long key = ~0ull;
    u32 cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = wbase + (u64)k * 64 + lane;
        bool nons = false;
        if (i < n) {
            if ((hw[k] >> lane) & 1ull) {
                const bool next_opens = (i + 1 >= n) || (lane == 63 ? (hw[k + 1] & 1ull) : ((hw[k] >> (lane + 1)) & 1ull));
                const u32 fl = (u32)r[k] & 0xFFu;
                if (next_opens && !(fl & 2u)) {
                    const unsigned long long mine = ((unsigned long long)(fl >> 2) << 32) | g[k];
                    key = mine < key ? mine : key;
                } else nons = true;
            } else {
                atomicOr(&nonrep[g[k] >> 5], 1u << (31 - (g[k] & 31)));
            }
        }
        const u64 bn = __ballot(nons);
        cnt += (u32)__popcll(bn);
        if (lane == 0) nons_mask[(wbase >> 6) + k] = bn;       // kept for k_class_emit
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if (lane == 0) { blk_nons[wid] = cnt; blk_min[wid] = key; }
}

__global__ __launch_bounds__(256) void k_min_u64(const unsigned long long *__restrict__ in, u64 n, unsigned long long *out)
{
    __shared__ unsigned long long s_min[4];
    unsigned long long m = ~0ull;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) { const unsigned long long v = in[i]; m = v < m ? v : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o, 64);
        m = other < m ? other : m;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) m = s_min[w] < m ? s_min[w] : m;
        if (m != ~0ull) atomicMin(out, m);
    }
}

// the same from the stored window-validity bitmap (single-GPU run: k_kmer_count covered every position)
__global__ __launch_bounds__(256) void k_rep_mask_words(const u32 *__restrict__ valid, const u32 *nonrep, u32 *rep, u64 n_words)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w < n_words) rep[w] = valid[w] & ~nonrep[w];
}

// rep[w] = valid windows of positions 32w..32w+31 (bit 31-i = position 32w+i) minus nonrep[w]; rep may alias nonrep.
// One thread per word (window_bits); sharded runs, where no rank's k_kmer_count has covered every position.
__global__ __launch_bounds__(256) void k_rep_mask(DevParams p, const u32 *__restrict__ bad, const u32 *nonrep, u32 *rep)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w < p.n_words) rep[w] = window_bits(p, bad, w) & ~nonrep[w];
}

// Lists the classes with >= 2 entries as rep << 32 | index of their first record, and their sizes (records up to the next
// class head), at offsets from the scan of blk_nons.  One thread per mask word (64 records; a thread per record spent 6 ms
// starting 46 M waves that had nothing to list).  The next head is in the record's own mask word for all but the word's last
// head, for which it is in the next word -- or, for the long classes (Mbp of tandem repeats make classes of 10^6 records,
// 16 000 empty mask words), further on: those are finished by the whole wave, 64 words per step.
__global__ __launch_bounds__(256) void k_class_emit(const u32 *__restrict__ gpos, const u64 *__restrict__ head_mask, u64 n,
                                                    const u64 *__restrict__ nons_mask, u64 *__restrict__ cls, u32 *__restrict__ cls_m,
                                                    const u64 *__restrict__ nons_off)
{
    const u64 n_words = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool in = w < n_words;
    const u64 bn = in ? nons_mask[w] : 0ull;
    if (!__ballot(bn != 0)) return;                                    // (wave-uniform)
    const u64 hm = in ? head_mask[w] : 0ull;
    u64 nxt = __shfl_down(hm, 1, 64);                                  // the next word's heads
    if (lane == 63) nxt = (w + 1 < n_words) ? head_mask[w + 1] : 0ull;
    u64 at = 0;
    if (bn) {                                                          // nons_off is per 256 records = 4 words
        at = nons_off[w >> 2];
        for (u64 k = w & ~3ull; k < w; k++) at += (u32)__popcll(nons_mask[k]);
    }
    u64 pend_i = ~0ull, pend_at = 0;
    u64 bits = bn;
    while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const u64 i = w * 64 + (u64)b;
        const u64 rest = b == 63 ? 0ull : (hm >> (b + 1));
        u64 next = n;
        bool found = true;
        if (rest) next = i + 1 + (u64)(__ffsll((long long)rest) - 1);
        else if (w + 1 >= n_words) next = n;                           // the last class: it ends with the records
        else if (nxt) next = (w + 1) * 64 + (u64)(__ffsll((long long)nxt) - 1);
        else found = false;
        if (found) { cls[at] = ((u64)gpos[i] << 32) | i; cls_m[at] = (u32)(next - i); }
        else { pend_i = i; pend_at = at; }                             // (only the word's last head can be left over)
        at++;
    }
    u64 todo = __ballot(pend_i != ~0ull);
    while (todo) {                                                     // long classes, one at a time, 64 mask words per step
        const int src = __ffsll((long long)todo) - 1;
        const u64 from = (__shfl(pend_i, src, 64) >> 6) + 2;
        u64 res = n;
        for (u64 base = from; base < n_words; base += 64) {
            const u64 ww = base + lane;
            const u64 bb = ww < n_words ? head_mask[ww] : 0ull;
            const u64 any = __ballot(bb != 0);
            if (any) {
                const int l = __ffsll((long long)any) - 1;
                const u64 bw = __shfl(bb, l, 64);
                res = (base + l) * 64 + (u64)(__ffsll((long long)bw) - 1);
                break;
            }
        }
        if (lane == src) { cls[pend_at] = ((u64)gpos[pend_i] << 32) | pend_i; cls_m[pend_at] = (u32)(res - pend_i); }
        todo &= todo - 1;
    }
}

// ------------------------------------------------------------------ record mask -> positions of the set bits
// mask word w covers records 64w .. 64w+63 (bit l = record 64w+l).  One thread per word; blocks of 256 words.
__global__ __launch_bounds__(256) void k_mask_count(const u64 *__restrict__ mask, u64 n_words, u32 *__restrict__ blk_count)
{
    __shared__ u32 ws[4];
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 c = (w < n_words) ? (u32)__popcll(mask[w]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Dense masks (class heads: 96 % of the bits are set): a wave walks its 64 words, lane l takes bit l, so that consecutive
// lanes store consecutive outputs (one thread per word, each storing its own 60 positions, ran at a third of this rate).
__global__ __launch_bounds__(256) void k_mask_emit(const u64 *__restrict__ mask, u64 n_words, const u64 *__restrict__ blk_off,
                                                   u32 *__restrict__ pos)
{
    __shared__ u32 l4[4];
    const int lane = threadIdx.x & 63;
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;          // this lane's word; its wave's words are w - lane .. w - lane + 63
    const u64 mine = (w < n_words) ? mask[w] : 0;
    const u32 excl = block_excl_scan_u32((u32)__popcll(mine), l4, nullptr);
    const u64 at0 = blk_off[blockIdx.x];
    const u64 lt = (1ull << lane) - 1ull;
    for (int k = 0; k < 64; k++) {
        const u64 word = __shfl(mine, k, 64);
        const u32 before = __shfl(excl, k, 64);
        if ((word >> lane) & 1ull) pos[at0 + before + (u32)__popcll(word & lt)] = (u32)((w - lane + k) * 64 + lane);
    }
}

// ------------------------------------------------------------------ bitmap -> sorted position list
__global__ __launch_bounds__(256) void k_bitmap_count(const u32 *__restrict__ bits, u64 n_words, u32 *__restrict__ block_counts)
{
    __shared__ u32 l4[4];
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 c = (w < n_words) ? (u32)__popc(bits[w]) : 0u;
    u32 tot;
    (void)block_excl_scan_u32(c, l4, &tot);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

// (dense as well: nearly every valid window is a representative; bit 31-i of word w = position 32w+i)
__global__ __launch_bounds__(256) void k_bitmap_emit(const u32 *__restrict__ bits, u64 n_words, const u64 *__restrict__ block_off,
                                                     u32 *__restrict__ out, u32 base)
{
    __shared__ u32 l4[4];
    const int lane = threadIdx.x & 63;
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    const u32 mine = (w < n_words) ? bits[w] : 0u;
    const u32 excl = block_excl_scan_u32((u32)__popc(mine), l4, nullptr);
    const u64 at0 = block_off[blockIdx.x];
    const u64 lt = (1ull << lane) - 1ull;
    for (int k = 0; k < 64; k += 2) {                            // two words = 64 positions per step
        const u32 w0 = __shfl(mine, k, 64), w1 = __shfl(mine, k + 1, 64);
        const u32 before = __shfl(excl, k, 64);
        const u64 both = (u64)__brev(w0) | ((u64)__brev(w1) << 32);   // bit l = position 32 * (first word) + l
        if ((both >> lane) & 1ull) out[at0 + before + (u32)__popcll(both & lt)] = base + (u32)((w - lane + k) * 32 + lane);
    }
}

// ------------------------------------------------------------------ exact table pieces
// Classes, records and lists are handled one per thread; one longer than this (Mbp of tandem repeats make classes of
// 10^6 entries) is queued in LDS and then shared by the whole workgroup.
constexpr u32 COOP_MIN = 128;
// ... and one longer than this is noted in a short list (DevCounters::Huge) and spread over a whole grid by a follow-up
// kernel; if the list is full the workgroup path takes it.
constexpr u32 HUGE_MIN = 1u << 16;
constexpr u32 HUGE_GRID = 1024;
__device__ __forceinline__ bool note_huge(DevCounters::Huge *h, u32 id)
{
    const u32 at = atomicAdd(&h->n, 1u);
    if (at >= 31u) return false;
    h->id[at] = id;
    return true;
}

// cls[k] = rep<<32 | index of the class's first record, cls_m[k] = its records.  flags: bit0 palindrome, bit1 flip, bit2 rep strand, bit3 leading.
__global__ __launch_bounds__(256) void k_class_info(const u64 *__restrict__ rec, const u32 *__restrict__ cls_m,
                                                    const u64 *__restrict__ cls, u32 n_cls, const DevCounters *__restrict__ ctr,
                                                    u32 *__restrict__ cls_flags, u32 *__restrict__ cls_entries)
{
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_cls) return;
    const u64 rep = cls[k] >> 32;
    const u32 st = (u32)cls[k], m = cls_m[k];
    const u32 fl = (u32)rec[st] & 0xFFu;                        // flags of the representative's window
    const bool pal = fl & 2u, strand = fl & 1u;
    const unsigned long long ord = ((unsigned long long)(fl >> 2) << 32) | rep;
    const bool leading = ord < ctr->first_singleton_key;       // class id precedes the first singleton's (Q3)
    const bool flip = strand && !leading;
    cls_flags[k] = (pal ? 1u : 0u) | (flip ? 2u : 0u) | (strand ? 4u : 0u) | (leading ? 8u : 0u);
    cls_entries[k] = m * (pal ? 2u : 1u);
}

// entries of a class in the reference's list order (HTS:356-408): occurrences by position;
// dir = strandSwitched (HTS:2552-2565), flipped when the representative's is 1 (HTS:424-432);
// a palindromic K-mer contributes (0,pos),(1,pos) (Q2, HTS:2597-2616).
__global__ __launch_bounds__(256) void k_class_entries(const u64 *__restrict__ rec, const u32 *__restrict__ gpos,
                                                       const u32 *__restrict__ cls_m, const u64 *__restrict__ cls, u32 n_cls,
                                                       const u32 *__restrict__ cls_flags, const u64 *__restrict__ ent_off,
                                                       u32 *__restrict__ ent, u8 *__restrict__ ent_dir, DevCounters *ctr)
{
    __shared__ u32 s_big[256];
    __shared__ u32 s_nbig;
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < n_cls) {
        const u32 st = (u32)cls[k], m = cls_m[k], fl = cls_flags[k];
        if (m > HUGE_MIN && note_huge(&ctr->huge_entries, k)) {
            // left to k_class_entries_huge
        } else if (m > COOP_MIN) {
            s_big[atomicAdd(&s_nbig, 1u)] = k;                  // a long class (low-complexity sequence): the whole block shares it
        } else {
            u64 o = ent_off[k];
            for (u32 j = 0; j < m; j++) {
                const u32 g = gpos[st + j];
                if (fl & 1u) {
                    ent[o] = g; ent_dir[o] = 0; ent[o + 1] = g; ent_dir[o + 1] = 1; o += 2;
                } else {
                    ent[o] = g; ent_dir[o] = (u8)(((u32)rec[st + j] & 1u) ^ ((fl >> 1) & 1u)); o++;
                }
            }
        }
    }
    __syncthreads();
    const u32 nbig = s_nbig;
    for (u32 b = 0; b < nbig; b++) {
        const u32 kb = s_big[b];
        const u32 st = (u32)cls[kb], m = cls_m[kb], fl = cls_flags[kb];
        const u64 o = ent_off[kb];
        for (u32 j = threadIdx.x; j < m; j += 256) {
            const u32 g = gpos[st + j];
            if (fl & 1u) {
                ent[o + 2ull * j] = g; ent_dir[o + 2ull * j] = 0; ent[o + 2ull * j + 1] = g; ent_dir[o + 2ull * j + 1] = 1;
            } else {
                ent[o + j] = g; ent_dir[o + j] = (u8)(((u32)rec[st + j] & 1u) ^ ((fl >> 1) & 1u));
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_class_entries_huge(const u64 *__restrict__ rec, const u32 *__restrict__ gpos,
                                                            const u32 *__restrict__ cls_m, const u64 *__restrict__ cls,
                                                            const u32 *__restrict__ cls_flags, const u64 *__restrict__ ent_off,
                                                            u32 *__restrict__ ent, u8 *__restrict__ ent_dir, const DevCounters *ctr)
{
    const u32 nh = ctr->huge_entries.n < 31u ? ctr->huge_entries.n : 31u;
    for (u32 h = 0; h < nh; h++) {
        const u32 kb = ctr->huge_entries.id[h];
        const u32 st = (u32)cls[kb], m = cls_m[kb], fl = cls_flags[kb];
        const u64 o = ent_off[kb];
        for (u32 j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
            const u32 g = gpos[st + j];
            if (fl & 1u) {
                ent[o + 2ull * j] = g; ent_dir[o + 2ull * j] = 0; ent[o + 2ull * j + 1] = g; ent_dir[o + 2ull * j + 1] = 1;
            } else {
                ent[o + j] = g; ent_dir[o + j] = (u8)(((u32)rec[st + j] & 1u) ^ ((fl >> 1) & 1u));
            }
        }
    }
}

// Can class c be merged into the block of class c-1?  (HTS:455-497: same size, representative
// one base further, every other entry same direction and one base further along its strand.)
__global__ __launch_bounds__(256) void k_class_merge(const u64 *__restrict__ cls, u32 n_cls, const u32 *__restrict__ cls_entries,
                                                     const u64 *__restrict__ ent_off, const u32 *__restrict__ ent,
                                                     const u8 *__restrict__ ent_dir, u8 *__restrict__ merge_prev)
{
    __shared__ u32 s_big[256];
    __shared__ u32 s_nbig;
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c < n_cls) {
        u8 ok = 0;
        if (c > 0 && (cls[c] >> 32) == (cls[c - 1] >> 32) + 1 && cls_entries[c] == cls_entries[c - 1]) {
            const u64 oa = ent_off[c - 1], ob = ent_off[c];
            const u32 ne = cls_entries[c];
            if (ne > COOP_MIN) {
                s_big[atomicAdd(&s_nbig, 1u)] = c;
            } else {
                ok = 1;
                for (u32 k = 1; k < ne; k++) {
                    const u8 da = ent_dir[oa + k];
                    if (da != ent_dir[ob + k]) { ok = 0; break; }
                    const u32 want = da ? ent[oa + k] - 1u : ent[oa + k] + 1u;
                    if (ent[ob + k] != want) { ok = 0; break; }
                }
            }
        }
        merge_prev[c] = ok;
    }
    __syncthreads();
    const u32 nbig = s_nbig;
    for (u32 b = 0; b < nbig; b++) {                             // two long neighbours of equal size: compared by the whole block
        const u32 cb = s_big[b];
        const u64 oa = ent_off[cb - 1], ob = ent_off[cb];
        const u32 ne = cls_entries[cb];
        int ok = 1;
        for (u32 base = 1; base < ne; base += 256) {
            const u32 k = base + threadIdx.x;
            int mine = 1;
            if (k < ne) {
                const u8 da = ent_dir[oa + k];
                const u32 want = da ? ent[oa + k] - 1u : ent[oa + k] + 1u;
                mine = (da == ent_dir[ob + k]) && (ent[ob + k] == want);
            }
            if (!__syncthreads_and(mine)) { ok = 0; break; }      // uniform result: every thread leaves together
        }
        if (threadIdx.x == 0) merge_prev[cb] = (u8)ok;
    }
}

// ------------------------------------------------------------------ seeds
// 32-bit seed code (HTS:1793-1838): the seed's left positions s-1, s-1+n, ... ascending, then
// (8-q) literal A,T pairs, then its mirrored right positions ascending.
__device__ __forceinline__ u32 seed_code(const DevParams &p, u64 hi, u64 lo, int seed /*1-based*/)
{
    const int n = p.nseeds, q = p.seed_q;
    u32 v = 0;
    for (int j = 0; j < q; j++) v = (v << 2) | base_at(hi, lo, (seed - 1) + n * j);
    for (int j = 0; j < 8 - q; j++) v = (v << 4) | 3u;
    for (int j = q - 1; j >= 0; j--) v = (v << 2) | base_at(hi, lo, p.K - 1 - ((seed - 1) + n * j));
    return v;
}

__device__ __forceinline__ u32 rc_code(u32 c)      // HTS:1904-1911
{
    u32 r = __brev(c);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return ~r;
}

__global__ __launch_bounds__(256) void k_seed_codes(DevParams p, const u64 *__restrict__ pk, const u32 *__restrict__ reps,
                                                    u64 n_reps, int seed, u64 *__restrict__ out)
{
    const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reps) return;
    const u32 g = reps[r];
    u64 hi, lo;
    load_window(p, pk, g, hi, lo);
    const u32 fw = seed_code(p, hi, lo, seed), rc = rc_code(fw);
    out[r] = ((u64)(fw <= rc ? fw : rc) << 32) | g;          // canonical code (HTS:2732-2744)
}

// seed_code with the seed count known at compile time.  A seed's bases sit NS apart, so a code is two strided gathers:
// the left positions s, s+NS, ... from the K-mer's first word, the mirrored right ones from `tail` = its last 32 bases
// right-aligned.  gather_stride compacts eight 2-bit items that lie 2 NS bits apart (item j at bit 2 NS j) into 16
// adjacent bits (item j at bit 2j) in three mask-and-shift steps; the loop over bases it replaced cost 250 VALU
// instructions per representative and made k_seed_codes_all ALU-bound.
template <int NS>
__device__ __forceinline__ u32 gather_stride(u64 x)
{
    constexpr int D = 2 * NS;
    constexpr u64 M0 = (3ull) | (3ull << D) | (3ull << (2 * D)) | (3ull << (3 * D)) | (3ull << (4 * D)) | (3ull << (5 * D)) | (3ull << (6 * D)) | (3ull << (7 * D));
    constexpr u64 M1 = (0xFull) | (0xFull << (2 * D)) | (0xFull << (4 * D)) | (0xFull << (6 * D));
    constexpr u64 M2 = (0xFFull) | (0xFFull << (4 * D));
    x &= M0;
    x = (x | (x >> (D - 2))) & M1;
    x = (x | (x >> (2 * D - 4))) & M2;
    x = (x | (x >> (4 * D - 8))) & 0xFFFFull;
    return (u32)x;
}
template <int NS>
__device__ __forceinline__ u32 seed_code_fixed(u64 hi, u64 tail, int s /*0-based seed*/, int q)
{
    // left: base s + NS j at bits 62 - 2 (s + NS j) of hi; shifted so that j = 7 is item 0, the result has j = 0 on top
    const u32 left = gather_stride<NS>(hi >> (62 - 2 * s - 14 * NS)) >> (16 - 2 * q);
    // right: base s + NS j from the end at bits 2 (s + NS j) of tail: item j, j = 0 lowest
    const u32 right = gather_stride<NS>(tail >> (2 * s)) & ((1u << (2 * q)) - 1u);
    const u32 fill = q < 8 ? (0x33333333u >> (4 * q)) << (2 * q) : 0u;        // (8 - q) literal A,T pairs in the middle
    return (q < 8 ? left << (32 - 2 * q) : left << 16) | fill | right;
}

// All seeds of a run in one pass over the representatives: one window load, NS codes, and -- because a block covers
// exactly one 4096-record tile of the key-only sort -- the per-tile histogram of each code array's first sort digit
// (bits 32..39 of the record = low byte of the code), which saves that sort its first histogram pass.
//
// Sketch of a record of seed s (optional): the codes of ALL the other seeds, s+1, s+2, ... (cyclically), in the
// orientation that made this seed's code canonical (the code of seed s' of the reverse complement is rc_code of the
// forward one: seeds are mirror-symmetric position sets, HTS:2015-2066).  Two records with the same canonical, non-
// palindromic code are the same K-mer positions in that orientation, so the number of differing bases between their
// sketches is exactly the number of mismatches inside the other seeds: the candidate kernels dismiss a partner on it without
// touching the genome.  For K = 50, d = 2 the three seeds cover 48 of the 50 bases.  Layout: NS = 2: one code in skc;
// NS = 3: two codes in skb (first other seed in the high word); NS = 4: two in skb, the third in skc.
template <int NS>
__global__ __launch_bounds__(256) void k_seed_codes_all(DevParams p, const u64 *__restrict__ pk, const u32 *__restrict__ reps, u64 n_reps,
                                                        u64 *__restrict__ out0, u64 *__restrict__ out1, u64 *__restrict__ out2,
                                                        u64 *__restrict__ out3, u32 *__restrict__ hist0, u32 *__restrict__ hist1,
                                                        u32 *__restrict__ hist2, u32 *__restrict__ hist3, SeedSketches sk, int per_thread, int only,
                                                        int hist_shift /* the first sort digit: code bits hist_shift .. + 7 */)
{
    __shared__ u32 h[NS][256];
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NS; s++) h[s][tid] = 0;
    __syncthreads();
    u64 *const outs[4] = {out0, out1, out2, out3};
    const u64 base = (u64)blockIdx.x * 256 * per_thread;    // one tile of the sort that follows (16 or 32 records per thread)
    // four records at a time: their positions first, then their windows, then the codes -- so that four gathers are in
    // flight per thread instead of one (position -> window is a dependent chain; per_thread is a multiple of 4)
    for (int k0 = 0; k0 < per_thread; k0 += 4) {
        u32 gq[4];
        u64 hq[4], lq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64 r = base + (u64)(k0 + j) * 256 + tid;
            gq[j] = r < n_reps ? __builtin_nontemporal_load(reps + r) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64 r = base + (u64)(k0 + j) * 256 + tid;
            hq[j] = lq[j] = 0;
            if (r < n_reps) load_window(p, pk, gq[j], hq[j], lq[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u64 r = base + (u64)(k0 + j) * 256 + tid;
            if (r >= n_reps) break;
            const u32 g = gq[j];
            const u64 hi = hq[j], lo = lq[j];
            const int sh = 128 - 2 * p.K;                     // 0 .. 62
            const u64 tail = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
            u32 fwc[NS], rcc[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) { fwc[s] = seed_code_fixed<NS>(hi, tail, s, p.seed_q); rcc[s] = rc_code(fwc[s]); }
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (only >= 0 && s != only) continue;             // one seed per launch (large references: memory)
                const bool fwd = fwc[s] <= rcc[s];
                const u32 c = fwd ? fwc[s] : rcc[s];
                outs[s][r] = ((u64)c << 32) | g;
                atomicAdd(&h[s][(c >> hist_shift) & 255u], 1u);
                if (sk.on) {                                       // (an array that is not wanted is null: the sharded flow moves b only)
                    const u32 o1 = fwd ? fwc[(s + 1) % NS] : rcc[(s + 1) % NS];
                    if (NS == 2) { if (sk.c[s]) sk.c[s][r] = o1; }
                    else {
                        const u32 o2 = fwd ? fwc[(s + 2) % NS] : rcc[(s + 2) % NS];
                        if (sk.b[s]) sk.b[s][r] = ((u64)o1 << 32) | o2;
                        if (NS == 4 && sk.c[s]) sk.c[s][r] = fwd ? fwc[(s + 3) % NS] : rcc[(s + 3) % NS];
                    }
                }
            }
        }
    }
    __syncthreads();
    u32 *const hists[4] = {hist0, hist1, hist2, hist3};
#pragma unroll
    for (int s = 0; s < NS; s++) if ((only < 0 || s == only) && hists[s]) hists[s][(size_t)blockIdx.x * 256 + tid] = h[s][tid];
}

// The candidate kernels' test of a pair of records with one canonical, non-palindromic code on their sketches (k_seed_codes_all):
// false = no slot.  More than d differing bases cannot be reported (HTS:2138-2265 aborts at > d); a pair that agrees on an
// earlier seed's whole code has been reported by that seed (HTS:2222-2232; k_judge would dismiss it): DevParams::sk_early.
struct EarlyWords {
    u32 nh, nl, nc;                                       // 0 where the word holds an earlier seed's code, else 1
    __device__ __forceinline__ explicit EarlyWords(u32 early) : nh((early & 1u) ? 0u : 1u), nl((early & 2u) ? 0u : 1u), nc((early & 4u) ? 0u : 1u) {}
};
template <bool SKB, bool SKC>
__device__ __forceinline__ bool sketch_pass(u64 bi, u64 bj, u32 ci, u32 cj, int d, const EarlyWords &e)
{
    int mm = 0;
    bool seen = false;
    if (SKB) {
        const u64 x = bi ^ bj;
        mm += __popcll((x | (x >> 1)) & M55);
        seen = (((u32)(x >> 32) | e.nh) == 0u) | (((u32)x | e.nl) == 0u);
    }
    if (SKC) {
        const u32 y = ci ^ cj;
        mm += __popc((y | (y >> 1)) & 0x55555555u);
        seen = seen | ((y | e.nc) == 0u);
    }
    return mm <= d && !seen;
}

struct Mism { u64 hi, lo; };
__device__ __forceinline__ Mism mism_of(u64 ah, u64 al, u64 bh, u64 bl)
{
    Mism m;
    const u64 xh = ah ^ bh, xl = al ^ bl;
    m.hi = (xh | (xh >> 1)) & M55;
    m.lo = (xl | (xl >> 1)) & M55;
    return m;
}

// smallest remaining 1-based mismatch offset (0 when none is left); clears it
__device__ __forceinline__ u32 pop_offset(u64 &eh, u64 &el)
{
    if (eh) { const int lz = __clzll((long long)eh); eh &= ~(0x8000000000000000ull >> lz); return (u32)(lz >> 1) + 1u; }
    if (el) { const int lz = __clzll((long long)el); el &= ~(0x8000000000000000ull >> lz); return 33u + (u32)(lz >> 1); }
    return 0;
}

// CompareAndReport (HTS:2138-2265) on one orientation's mismatch mask.  Returns meta or 0xFFFFFFFF.
__device__ __forceinline__ u32 judge(const DevParams &p, Mism m, int seed, u32 dir)
{
    const u64 eh = m.hi & ~p.seedmask_hi[seed - 1], el = m.lo & ~p.seedmask_lo[seed - 1];
    const int nerr = __popcll(eh) + __popcll(el);
    if (nerr == 0 || nerr > p.d) return 0xFFFFFFFFu;
    for (int s = 0; s < seed - 1; s++)                          // reported by the first seed that matches exactly (HTS:2222-2232)
        if (((m.hi & p.seedmask_hi[s]) | (m.lo & p.seedmask_lo[s])) == 0) return 0xFFFFFFFFu;
    u64 th = eh, tl = el;
    const u32 o1 = pop_offset(th, tl), o2 = pop_offset(th, tl), o3 = pop_offset(th, tl);
    return meta_pack(dir, o1, o2, o3, (u32)(seed - 1), 0, 0);
}

// Pair verification over the code-sorted records of one seed (HTS:2671-2800, 2138-2265), in three steps.
//
// 1. k_candidates: every record i (k-mer 1) meets every EARLIER record of its equal-code run (HTS:2753-2762: k-mer 1 is
//    the later one).  A workgroup streams a tile of VT_TILE consecutive records (plus the VT_HALO records before it) into
//    LDS with coalesced loads and every thread walks back from its records inside LDS; a run that reaches beyond the halo
//    is followed into global memory.  Nothing is gathered here: on a GRCh38-sized reference most records have a chance
//    partner (2.9e9 records, 2^31 canonical codes), so the walk itself is the work, and a lane that went to the genome
//    for one partner would stall its 63 neighbours for a memory round trip.  What a walk finds worth a look at the
//    genome is written out as a SLOT (k-mer 1, k-mer 2, kind).
//    Sketches (sk_b / sk_c, sorted along with the codes; either may be null): the other seeds' codes in the orientation
//    of the canonical code (k_seed_codes_all).  Their differing bases ARE the mismatches inside those seeds; a partner
//    with more than d of them cannot be reported (HTS:2138-2265 aborts at > d) and gets no slot.
//    Two passes and no global atomics: WRITE=false stores each thread's slot count and each workgroup's slot / comparison
//    sums, the workgroup sums are scanned, WRITE=true repeats the tiles that have slots and stores them.
//    Skew: the members of a long run (a repeat family whose copies share a seed) sit side by side, so the lanes of a
//    wave walk the same stretch together; their slots are judged by as many threads as there are slots.
// 2. k_judge: one thread per slot reads both K-mers from the packed genome and applies CompareAndReport (HTS:2138-2265);
//    the slot's kind becomes the pair's meta word, or SLOT_NONE.
// 3. k_slot_count / k_slot_compact: the reported pairs, compacted behind the earlier seeds' records.
//
// Slot kinds.  The orientation in which two K-mers are compared is the one in which the seed positions agree; a
// palindromic code agrees in both, and the reference then compares both ways, meets every earlier entry twice and
// starts with the K-mer itself (Q1, HTS:2769-2792): kinds 1-4 are the (orientation, copy) cases of such a pair, 5 the
// self comparison.  The order of the records is deterministic; nothing downstream depends on it: items are sorted by
// (source, target) and lists with tied targets are finished in the reference's discovery order, recomputed per item.
constexpr int VT_PER = 8;
constexpr int VT_TILE = 256 * VT_PER;        // records per workgroup
constexpr int VT_HALO = 256;                 // records before the tile kept in LDS as well
// The counting pass keeps what it finds: up to VT_QUOTA slots per tile, appended to a list in LDS (as tile-local indices,
// each with its thread and its rank among that thread's slots, so that its final place -- thread prefix + rank -- does not
// depend on the order of the appends) and written to the tile's quota of a staging array.  Only the tiles that do not fit
// -- dense repeat families, palindromic codes, runs that leave LDS -- are walked a second time by the WRITE pass.
constexpr int VT_QUOTA = 192;
constexpr u32 SLOT_NONE = 0xFFFFFFFFu;
enum { SLOT_PLAIN = 0, SLOT_PAL_FWD0, SLOT_PAL_FWD1, SLOT_PAL_REV0, SLOT_PAL_REV1, SLOT_PAL_SELF };

template <bool WRITE, bool SKB, bool SKC>
__global__ __launch_bounds__(256) void k_candidates(DevParams p, const u64 *__restrict__ codes, u64 n, u32 *__restrict__ cnt_slots /* [blocks * 256] */,
                                                    u32 *__restrict__ blk_slots, u64 *__restrict__ blk_cand, const u64 *__restrict__ blk_off,
                                                    PairRec *__restrict__ slots, const u64 *__restrict__ sk_b, const u32 *__restrict__ sk_c,
                                                    u64 *__restrict__ staged /* [blocks * VT_QUOTA] */, u8 *__restrict__ blk_staged,
                                                    const u8 *__restrict__ blk_mode /* 1: k_candidates_pairs has done this tile (null: none) */,
                                                    const u32 *__restrict__ tiles /* the tiles to do, one per workgroup (null: workgroup b does tile b) */)
{
    const u32 tile = tiles ? tiles[blockIdx.x] : blockIdx.x;
    if (blk_mode && blk_mode[tile]) return;
    __shared__ u64 s_key[VT_TILE + VT_HALO];
    __shared__ u64 s_b[SKB ? VT_TILE + VT_HALO : 1];
    __shared__ u32 s_c[SKC ? VT_TILE + VT_HALO : 1];
    __shared__ u32 s_val[WRITE ? 1 : VT_QUOTA];      // kept slots: (LDS slot of k-mer 1) << 16 | LDS slot of k-mer 2
    __shared__ uint16_t s_own[WRITE ? 1 : VT_QUOTA];      // ... thread << 8 | rank among the thread's slots
    __shared__ u32 s_pre[WRITE ? 1 : 256];           // slots of the threads before each thread
    __shared__ u32 s_n;
    __shared__ u32 l4[4];
    __shared__ u64 c4[4];
    const int tid = threadIdx.x;
    const u64 base = (u64)tile * VT_TILE;
    u64 at = 0;
    u32 mine = 1;
    bool spill = false;                                      // counting pass: this thread's slots cannot all be kept
    if (!WRITE && tid == 0) s_n = 0;                         // (published by the barrier behind the tile loads)
    if (WRITE) {
        if (blk_slots[tile] == 0 || blk_staged[tile]) return;   // nothing worth a look in this tile, or the counting pass kept it all (uniform)
        mine = cnt_slots[(u64)tile * 256 + tid];
        at = blk_off[tile] + block_excl_scan_u32(mine, l4, nullptr);
    }
    const u64 lo = base >= (u64)VT_HALO ? base - VT_HALO : 0;       // first record held in LDS, at slot `first`
    const int first = (int)(VT_HALO - (base - lo));
    const u64 hi = base + VT_TILE < n ? base + VT_TILE : n;
    {
        // every load of the tile is issued before the first is used (one load -> one LDS store at a time left the
        // workgroup waiting for ten memory round trips in a row)
        constexpr int LD = (VT_TILE + VT_HALO) / 256;
        u64 rk[LD], rb[SKB ? LD : 1];
        u32 rc[SKC ? LD : 1];
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const u64 g = lo + (u64)(q * 256 + tid);
            const bool in = g < hi;
            rk[q] = in ? __builtin_nontemporal_load(codes + g) : 0ull;
            if (SKB) rb[q] = in ? __builtin_nontemporal_load(sk_b + g) : 0ull;
            if (SKC) rc[q] = in ? __builtin_nontemporal_load(sk_c + g) : 0u;
        }
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const int slot = first + q * 256 + tid;
            if (slot < VT_TILE + VT_HALO) {
                s_key[slot] = rk[q];
                if (SKB) s_b[slot] = rb[q];
                if (SKC) s_c[slot] = rc[q];
            }
        }
    }
    __syncthreads();
    const bool have_sk = SKB || SKC;
    const EarlyWords early(p.sk_early);
    u32 nslot = 0;
    u64 ncand = 0;
#pragma unroll 1
    for (int k = 0; k < (mine ? VT_PER : 0); k++) {         // (second pass: a thread without slots does not walk again)
        const int li = VT_HALO + k * 256 + tid;
        if (base + (u64)(k * 256 + tid) >= n) break;
        const u64 ci = s_key[li];
        const u32 code = (u32)(ci >> 32), gi = (u32)ci;
        // the record before this one is in LDS too (slot li - 1 >= first) unless this is record 0 of the whole list
        const bool has_prev = li > first && (u32)(s_key[li - 1] >> 32) == code;
        const bool pal = rc_code(code) == code;
        if (!has_prev && !pal) continue;
        const bool filter = have_sk && !pal;
        const u64 bi = (filter && SKB) ? s_b[li] : 0ull;
        const u32 qi = (filter && SKC) ? s_c[li] : 0u;
        u32 nc = 0;                                             // comparisons of this record (added to ncand below)
        if (pal) {                                              // second pass starts with the k-mer itself (HTS:2781-2787)
            nc++;
            if (WRITE) { slots[at].p1 = gi; slots[at].p2 = gi; slots[at].meta = SLOT_PAL_SELF; at++; }
            else { nslot++; spill = true; }                     // (kept slots are plain ones)
        }
        // one earlier record of the run (HTS:2753-2762): a comparison the reference makes, and perhaps a slot
        auto visit = [&](u32 gj, u64 bj, u32 qj, int lj /* LDS slot of the partner, -1: it came from global memory */) {
            if (pal) {
                nc += 4;
                if (WRITE) for (u32 c = SLOT_PAL_FWD0; c <= SLOT_PAL_REV1; c++, at++) { slots[at].p1 = gi; slots[at].p2 = gj; slots[at].meta = c; }
                else nslot += 4;
                return;
            }
            nc++;                                               // whatever the sketches say
            if (filter && !sketch_pass<SKB, SKC>(bi, bj, qi, qj, p.d, early)) return;
            if (WRITE) { slots[at].p1 = gi; slots[at].p2 = gj; slots[at].meta = SLOT_PLAIN; at++; }
            else {
                const u32 e = (lj >= 0 && nslot < 256u) ? atomicAdd(&s_n, 1u) : (u32)VT_QUOTA;
                if (e < (u32)VT_QUOTA) { s_val[WRITE ? 0 : e] = ((u32)li << 16) | (u32)lj; s_own[WRITE ? 0 : e] = (uint16_t)((tid << 8) | nslot); }
                else spill = true;
                nslot++;
            }
        };
        // walk back inside LDS: slots li-1 .. first  (a tight loop of its own: the common case)
        int j = li - 1;
        bool more = has_prev;
        while (more) {
            visit((u32)s_key[j], SKB ? s_b[j] : 0ull, SKC ? s_c[j] : 0u, j);
            j--;
            more = j >= first && (u32)(s_key[j] >> 32) == code;
        }
        ncand += nc;
        // a run that reaches the start of the halo may go on in global memory: records lo-1 .. 0
        if (has_prev && j < first && lo > 0) {
            for (u64 g = lo; g-- > 0;) {
                const u64 cj = codes[g];
                if ((u32)(cj >> 32) != code) break;
                nc = 0;
                visit((u32)cj, SKB ? sk_b[g] : 0ull, SKC ? sk_c[g] : 0u, -1);
                ncand += nc;
            }
        }
    }
    if (!WRITE) {
        u64 c = ncand;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0) c4[tid >> 6] = c;
        u32 total;
        const u32 pre = block_excl_scan_u32(nslot, l4, &total);            // (its barriers also publish c4)
        s_pre[WRITE ? 0 : tid] = pre;
        const bool keep_all = !__syncthreads_or(spill) && total <= (u32)VT_QUOTA;      // (the barrier also publishes s_pre)
        if (keep_all) {                                                     // the tile's slots, in thread order, at its place in the staging array
            if ((u32)tid < total) {
                const u32 e = s_val[WRITE ? 0 : tid], o = s_own[WRITE ? 0 : tid];
                staged[(u64)tile * VT_QUOTA + s_pre[WRITE ? 0 : (o >> 8)] + (o & 255u)] = ((u64)(u32)s_key[e >> 16] << 32) | (u32)s_key[e & 0xFFFFu];
            }
        } else cnt_slots[(u64)tile * 256 + tid] = nslot;              // the WRITE pass walks this tile again
        if (tid == 0) {
            blk_slots[tile] = total;
            blk_staged[tile] = keep_all ? 1 : 0;
            blk_cand[tile] = c4[0] + c4[1] + c4[2] + c4[3];
        }
    }
}

// Inclusive scans over the 64 lanes of a wave in six DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15
// and 31 carry the row totals on): VALU moves, where __shfl_up goes through the LDS crossbar (ds_bpermute) every step.
// 0 is the identity of both operations used here, and the value a lane keeps when its source lies outside its row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_or_zero(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ u32 wave_scan_max(u32 v)
{
    u32 t;
    t = dpp_or_zero<0x111, 0xF>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x112, 0xF>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x114, 0xF>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x118, 0xF>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x142, 0xA>(v); v = v > t ? v : t;
    t = dpp_or_zero<0x143, 0xC>(v); v = v > t ? v : t;
    return v;
}
__device__ __forceinline__ u32 wave_scan_add(u32 v)
{
    v += dpp_or_zero<0x111, 0xF>(v);
    v += dpp_or_zero<0x112, 0xF>(v);
    v += dpp_or_zero<0x114, 0xF>(v);
    v += dpp_or_zero<0x118, 0xF>(v);
    v += dpp_or_zero<0x142, 0xA>(v);
    v += dpp_or_zero<0x143, 0xC>(v);
    return v;
}

// k_candidates, balanced.  The walk above gives every record to one lane, and a wave goes on until its longest walk has
// ended: with 1.4 partners per record on average and runs of 20 in young repeat families most lanes idle most of the time
// (8 100 VALU wave instructions per tile, a third of the wave cycles issuing).  Here the PAIRS of a wave's 512 records are
// numbered and dealt out 64 at a time:
//   1. run rank r of every record (how many earlier records of its run are in LDS) from a segmented scan of the head flags;
//   2. pair q of the wave (q < W = sum of r) belongs to the record whose prefix sum P is the last one <= q: every record
//      with r > 0 drops a marker (its index, P) at place P of a per-wave LDS strip, 512 places at a time, and an inclusive
//      max-scan over the strip tells every lane whose pair it holds; the partner is the record q - P + 1 places before it.
//   3. the sketch test, and a slot for what passes, ranked by (wave, q): the order of the slots does not depend on timing.
// Tiles this cannot do -- a palindromic code (HTS:2769-2792: other slot kinds), a run that may go on before the LDS copy --
// are left to k_candidates (blk_mode 0); k_candidates skips the others.
constexpr int VP_STRIP = 128;                          // u32 words of a wave's strip
constexpr int PS_STRIP = 2 * VP_STRIP;                 // pair places per pass over the strip: 16-bit markers
template <bool WRITE, bool SKB, bool SKC>
__global__ __launch_bounds__(256) void k_candidates_pairs(DevParams p, const u64 *__restrict__ codes, u64 n, u32 *__restrict__ cnt_slots /* [blocks * 256] */,
                                                          u32 *__restrict__ blk_slots, u64 *__restrict__ blk_cand, const u64 *__restrict__ blk_off,
                                                          PairRec *__restrict__ slots, const u64 *__restrict__ sk_b, const u32 *__restrict__ sk_c,
                                                          u64 *__restrict__ staged /* [blocks * VT_QUOTA] */, u8 *__restrict__ blk_staged,
                                                          u8 *__restrict__ blk_mode, const u32 *__restrict__ tiles /* as in k_candidates */)
{
    const u32 tile = tiles ? tiles[blockIdx.x] : blockIdx.x;
    __shared__ u64 s_key[VT_TILE + VT_HALO];
    __shared__ u64 s_b[SKB ? VT_TILE + VT_HALO : 1];
    __shared__ u32 s_c[SKC ? VT_TILE + VT_HALO : 1];
    __shared__ u32 s_mk[4][VP_STRIP];
    __shared__ u32 s_val[WRITE ? 1 : VT_QUOTA];      // kept slots: (LDS slot of k-mer 1) << 16 | LDS slot of k-mer 2
    __shared__ uint16_t s_own[WRITE ? 1 : VT_QUOTA]; // ... wave << 12 | rank among that wave's slots
    __shared__ u32 s_seg[4], s_w[4];
    __shared__ u32 s_n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 base = (u64)tile * VT_TILE;
    if (WRITE) {
        if (!blk_mode[tile] || blk_slots[tile] == 0 || blk_staged[tile]) return;    // not ours, nothing found, or all kept
    }
    if (!WRITE && tid == 0) s_n = 0;
    const u64 lo = base >= (u64)VT_HALO ? base - VT_HALO : 0;       // first record held in LDS, at slot `first`
    const int first = (int)(VT_HALO - (base - lo));
    const u64 hi = base + VT_TILE < n ? base + VT_TILE : n;
    {
        constexpr int LD = (VT_TILE + VT_HALO) / 256;
        u64 rk[LD], rb[SKB ? LD : 1];
        u32 rc[SKC ? LD : 1];
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const u64 g = lo + (u64)(q * 256 + tid);
            const bool in = g < hi;
            rk[q] = in ? __builtin_nontemporal_load(codes + g) : 0ull;
            if (SKB) rb[q] = in ? __builtin_nontemporal_load(sk_b + g) : 0ull;
            if (SKC) rc[q] = in ? __builtin_nontemporal_load(sk_c + g) : 0u;
        }
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const int slot = first + q * 256 + tid;
            if (slot < VT_TILE + VT_HALO) {
                s_key[slot] = rk[q];
                if (SKB) s_b[slot] = rb[q];
                if (SKC) s_c[slot] = rc[q];
            }
        }
    }
    __syncthreads();
    // ---- 1. run ranks: lane l of wave w owns records 512 w + 8 l .. + 7 of the tile ----
    const int seg0 = VT_HALO + 512 * wv;                            // LDS slot of the wave's first record
    const int slot0 = seg0 + 8 * lane;
    const int n_tile = (int)(hi - base);                            // records of this tile (>= 1)
    u32 code[9];
    bool irregular = false;
#pragma unroll
    for (int j = 0; j < 9; j++) code[j] = (slot0 - 1 + j) >= first ? (u32)(s_key[slot0 - 1 + j] >> 32) : 0u;
    bool head[8];
    int last_head = -1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int sl = slot0 + j;
        const bool valid = (sl - VT_HALO) < n_tile;
        head[j] = !valid || sl == first || code[j + 1] != code[j];
        if (head[j]) last_head = j;
        if (valid && rc_code(code[j + 1]) == code[j + 1]) irregular = true;         // palindromic code: the walk's business
    }
    // Where each record's run starts (slot + 1): the latest head at or before it.  Across lanes that is a max-scan of every
    // lane's last head; the wave's first record, when it continues a run, finds its start by walking back.
    u32 carry_start_p1 = 0;
    if (lane == 0 && !head[0]) {
        int k = seg0 - 1;                                           // (equal to code[1], by !head[0])
        while (k - 1 >= first && (u32)(s_key[k - 1] >> 32) == code[1]) k--;
        carry_start_p1 = (u32)(k + 1);
    }
    const u32 incl = wave_scan_max(last_head >= 0 ? (u32)(slot0 + last_head + 1) : carry_start_p1);
    u32 cur = __shfl_up(incl, 1, 64);                               // run start + 1 of the previous lane's last record
    if (lane == 0) cur = carry_start_p1;
    u32 r[8];
    u32 lane_sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int sl = slot0 + j;
        if (head[j]) cur = (u32)(sl + 1);
        const u32 rank = (u32)(sl + 1) - cur;                       // earlier records of its run that are in LDS
        r[j] = rank;
        lane_sum += rank;
        if (rank && cur == (u32)(first + 1) && lo > 0) irregular = true;     // the run may go on before the LDS copy
    }
    irregular = __syncthreads_or(irregular);
    if (irregular) {
        if (!WRITE && tid == 0) blk_mode[tile] = 0;
        return;
    }
    // ---- 2. pairs of the wave, 64 at a time ----
    const u32 P = wave_scan_add(lane_sum);                          // inclusive scan of the lane sums
    const u32 W = (u32)__builtin_amdgcn_readlane((int)P, 63);
    u32 pr[8];
    {
        u32 x = P - lane_sum;
#pragma unroll
        for (int j = 0; j < 8; j++) { pr[j] = x; x += r[j]; }
    }
    u32 seg_pre = 0;                                                // WRITE: slots of the waves before this one
    if (WRITE) for (int w = 0; w < wv; w++) seg_pre += cnt_slots[(u64)tile * 256 + w];
    const u64 at0 = WRITE ? blk_off[tile] + seg_pre : 0;
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 wslots = 0, carry_o = 0, carry_P = 0;
    bool spill = false;
    // (16-bit markers: 256 pair places per pass of the strip in the wave's 512 bytes; a marker's place is its P, its value the record's
    //  index + 1 -- see k_candidates_pairs_ps)
    typedef uint16_t __attribute__((may_alias)) marker_t;
    marker_t *mk = reinterpret_cast<marker_t *>(s_mk[wv]);
    u32 *mk32 = s_mk[wv];
    const EarlyWords early(p.sk_early);
    for (u32 c0 = 0; c0 < W; c0 += PS_STRIP) {
#pragma unroll
        for (int x = 0; x < PS_STRIP / 128; x++) mk32[x * 64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (r[j] && pr[j] - c0 < (u32)PS_STRIP) mk[pr[j] - c0] = (marker_t)(8 * lane + j + 1);
        __builtin_amdgcn_wave_barrier();
        for (u32 t = 0; t < (u32)PS_STRIP && c0 + t < W; t += 64) {
            const u32 q = c0 + t + lane;
            const u32 here = mk[t + lane];
            const u32 v = wave_scan_max(here ? t + lane + 1u : 0u);          // place + 1 of the last marker at or before this lane's place; 0: none in this round
            const u32 o_rec = v ? (u32)mk[v - 1u] - 1u : carry_o;
            const u32 o_P = v ? c0 + v - 1u : carry_P;
            carry_o = (u32)__builtin_amdgcn_readlane((int)o_rec, 63); carry_P = (u32)__builtin_amdgcn_readlane((int)o_P, 63);
            bool pass = false;
            int li = 0, lj = 0;
            if (q < W) {
                li = seg0 + (int)o_rec;
                lj = li - (int)(q - o_P + 1u);
                pass = sketch_pass<SKB, SKC>(SKB ? s_b[li] : 0ull, SKB ? s_b[lj] : 0ull, SKC ? s_c[li] : 0u, SKC ? s_c[lj] : 0u, p.d, early);
            }
            const u64 bal = __ballot(pass);
            if (pass) {
                const u32 rank = wslots + (u32)__popcll(bal & lt_mask);
                if (WRITE) {
                    PairRec rec;
                    rec.p1 = (u32)s_key[li]; rec.p2 = (u32)s_key[lj]; rec.meta = SLOT_PLAIN;
                    slots[at0 + rank] = rec;
                } else {
                    const u32 e = rank < 4096u ? atomicAdd(&s_n, 1u) : (u32)VT_QUOTA;
                    if (e < (u32)VT_QUOTA) { s_val[WRITE ? 0 : e] = ((u32)li << 16) | (u32)lj; s_own[WRITE ? 0 : e] = (uint16_t)((wv << 12) | rank); }
                    else spill = true;
                }
            }
            wslots += (u32)__popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!WRITE) {
        if (lane == 0) { s_seg[wv] = wslots; s_w[wv] = W; }
        const bool keep = !__syncthreads_or(spill);                  // (the barrier also publishes s_seg, s_w, s_val, s_own)
        const u32 total = s_seg[0] + s_seg[1] + s_seg[2] + s_seg[3];
        const bool keep_all = keep && total <= (u32)VT_QUOTA;
        if (keep_all) {
            if ((u32)tid < total) {
                const u32 e = s_val[WRITE ? 0 : tid], o = s_own[WRITE ? 0 : tid];
                u32 pre = 0;
                for (u32 w = 0; w < (o >> 12); w++) pre += s_seg[w];
                staged[(u64)tile * VT_QUOTA + pre + (o & 4095u)] = ((u64)(u32)s_key[e >> 16] << 32) | (u32)s_key[e & 0xFFFFu];
            }
        } else if (tid < 4) cnt_slots[(u64)tile * 256 + tid] = s_seg[tid];      // the WRITE pass does this tile again
        if (tid == 0) {
            blk_mode[tile] = 1;
            blk_slots[tile] = total;
            blk_staged[tile] = keep_all ? 1 : 0;
            blk_cand[tile] = (u64)s_w[0] + s_w[1] + s_w[2] + s_w[3];
        }
    }
}

// ------------------------------------------------------------------ the last sort digit in LDS ("PS": partial sort)
// Round 3.  The seed sort was four global passes over 16-byte records: 1.1 TB of the build's 2.7 TB, and its scatter runs at
// 0.9 of a copy, so only fewer bytes make it faster.  In PS mode the records arrive sorted on the TOP 24 BITS of the code only
// (three passes), stable: a "bucket" (equal code >> 8; 170 records on average at 2.9e9 records, 340 for the smallest codes,
// since canonical codes crowd the low end) holds its records in position order with the low code byte unsorted.  The candidate
// kernels finish the sort where the records pass through LDS anyway: every bucket is ordered by its last byte by ONE wave -- the
// records go to registers, ranks come from the ballot match of the scatter kernel (rs_scatter) and a 256-counter histogram in
// the wave's strip of LDS, and they are written back to the bucket's own slots, in place.  What is in LDS afterwards is what
// the four-pass sort would have left there, and the pair logic above runs on it unchanged.
//
// Tiles.  A bucket must lie in ONE tile, so tiles are cut at bucket heads: with head(i) = record i opens a bucket (or i = 0, or
// i = n), C(x) = the first head in [x, x + PS_OVER], else x itself; tile t owns the records [C(t T), C((t + 1) T)).  Both ends
// depend on the records near them only, so neighbouring tiles agree.  LDS holds records t T - 1 .. t T + T + PS_OVER.  A cut that
// falls INSIDE a bucket (no head within PS_OVER records: a repeat family whose copies share the seed's first 12 bases) makes the
// tile that begins there "irregular"; so does a bucket of more than PS_MAXB records.  Irregular tiles are the walk's
// (k_candidates_ps): every record there meets the earlier records of its BUCKET, in LDS and on into global memory, and takes the ones with
// its own code -- in a bucket that large most have.  Palindromic codes (other slot kinds, Q1) are the walk's as well, but record by record:
// the balanced kernel skips such records and says so (blk_mode bit 1), the walk then visits only those in that tile.  A tile can
// therefore have slots from both kernels; the walk's counts live at index n_tiles + tile of every per-tile array.
constexpr int PS_TILE = 1792;
constexpr int PS_OVER = 510;
constexpr int PS_R = PS_TILE + PS_OVER + 2;          // 2304 LDS slots of the balanced kernel: slot k + 1 = record base + k
constexpr int PS_PER = PS_R / 256;                   // 9 slots per thread
constexpr int PS_WAVE = PS_R / 4;                    // 576 per wave
constexpr int PS_MAXB = 512;                         // a bucket is sorted in at most 8 rounds of 64
constexpr int PS_WHALO = 256;                        // the walk keeps that many records before the tile as well
constexpr int PS_WR = PS_WHALO + PS_TILE + PS_OVER + 2;      // 2560 slots: slot k + PS_WHALO = record base + k
static_assert(PS_R % 256 == 0 && PS_WR % 256 == 0 && PS_R % 64 == 0, "tile geometry");

__device__ __forceinline__ u32 ps_bucket(u64 key) { return (u32)(key >> 40); }

// Bit s of heads[] = "the record in slot s opens a bucket" for the slots of one wave round; rec = record index of the slot.
// A slot whose predecessor is not in LDS (slot 0) reports no head: it is never asked about.
__device__ __forceinline__ bool ps_is_head(const u64 *s_key, int slot, long long rec, u64 n)
{
    if (rec < 0 || (u64)rec > n) return false;
    if (rec == 0 || (u64)rec == n) return true;
    if (slot == 0) return false;
    return ps_bucket(s_key[slot]) != ps_bucket(s_key[slot - 1]);
}

// first set bit of the mask words in slots [sa, sb] (inclusive), or -1; uniform over the wave (broadcast LDS reads)
__device__ __forceinline__ int ps_first_head(const u64 *heads, int sa, int sb)
{
    if (sa > sb) return -1;
    int w = sa >> 6;
    u64 m = heads[w] & (~0ull << (sa & 63));
    for (;;) {
        if (m) { const int at = (w << 6) + __ffsll((long long)m) - 1; return at <= sb ? at : -1; }
        w++;
        if ((w << 6) > sb) return -1;
        m = heads[w];
    }
}

// The owned slot range [f, e) of a tile whose LDS copy starts HB records before `base`; cut_mid = the tile starts inside a bucket.
template <int HB>
__device__ __forceinline__ void ps_cuts(const u64 *heads, u64 base, u64 n, int &f, int &e, bool &cut_mid)
{
    const int sb = HB;                                                  // slot of record `base`
    const u64 xe = base + PS_TILE < n ? base + PS_TILE : n;
    const int se = sb + (int)(xe - base);
    const int lim_f = (int)((base + PS_OVER < n ? base + PS_OVER : n) - base) + sb;
    const int lim_e = (int)((xe + PS_OVER < n ? xe + PS_OVER : n) - base) + sb;
    const int hf = ps_first_head(heads, sb, lim_f), he = ps_first_head(heads, se, lim_e);
    cut_mid = hf < 0;
    f = hf < 0 ? sb : hf;
    e = he < 0 ? se : he;
}

// One wave groups the bucket in slots [h, h + B) (2 <= B <= PS_MAXB) by the low code byte, in place: records to registers,
// a 256-counter histogram in `hist` (the wave's 512 bytes: 16-bit counters, two to a word) filled by LDS atomics whose
// return value is the record's rank among the records of its byte, the histogram's scan, and back.  Every lane of the wave
// must call it with the same h, B.
// Order INSIDE a run of equal codes: records of different rounds keep their order; records of one round (64 consecutive
// slots) land in the order in which LDS serialises the atomics of one instruction.  Nothing here depends on that order: the pair
// kernels name the later position k-mer 1 by comparing the two positions (HTS:2753-2762: k-mer 1 is the genomically later one),
// every pair of a run is met exactly once whatever the order, and the order of the slots is of no consequence downstream
// (items are sorted; ties are settled from the items themselves).  The first version ranked by ballot match, as rs_scatter
// does (stable): 400 wave instructions per bucket, and the balanced kernel went from 18 to 39 ms -- issue-bound; this one takes 80.
// What is written back is (rank << 32) | position: a record's rank among the EARLIER records of its code is all the pair logic
// wants of the code from here on (the members of a run sit in rank order, so a record's partners are the `rank` slots before it).
// Records with a palindromic code get rank 0 -- their pairs are the walk's (HTS:2769-2792) -- and are reported by the return value.
template <bool SKB, bool SKC>
__device__ __forceinline__ bool ps_sort_bucket(u64 *s_key, u64 *s_b, u32 *s_c, u32 *hist /* [128] */, int h, int B, int lane)
{
    constexpr int RMAX = PS_MAXB / 64;
    u64 ka[RMAX], kb[SKB ? RMAX : 1];
    u32 kc[SKC ? RMAX : 1], rk[RMAX];
    bool any_pal = false;
    hist[lane] = 0u; hist[64 + lane] = 0u;
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        if (r * 64 < B) {
            const int i = r * 64 + lane;
            const bool in = i < B;
            ka[r] = in ? s_key[h + i] : 0ull;
            if (SKB) kb[r] = in ? s_b[h + i] : 0ull;
            if (SKC) kc[r] = in ? s_c[h + i] : 0u;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        if (r * 64 < B) {
            rk[r] = 0;
            if (r * 64 + lane < B) {
                const u32 d = (u32)(ka[r] >> 32) & 255u, sh = (d & 1u) * 16u;
                rk[r] = (atomicAdd(&hist[d >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of the 256 counters: lane l owns digits 4 l .. 4 l + 3 (words 2 l, 2 l + 1)
    {
        const u32 w0 = hist[2 * lane], w1 = hist[2 * lane + 1];
        const u32 c0 = w0 & 0xFFFFu, c1 = w0 >> 16, c2 = w1 & 0xFFFFu, c3 = w1 >> 16;
        const u32 sum = c0 + c1 + c2 + c3;
        const u32 ex = wave_scan_add(sum) - sum;
        __builtin_amdgcn_wave_barrier();
        hist[2 * lane] = ex | ((ex + c0) << 16);
        hist[2 * lane + 1] = (ex + c0 + c1) | ((ex + c0 + c1 + c2) << 16);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        if (r * 64 < B) {
            if (r * 64 + lane < B) {
                const u32 code = (u32)(ka[r] >> 32), d = code & 255u;
                const int dst = h + (int)((hist[d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu) + (int)rk[r];
                const bool pal = rc_code(code) == code;
                any_pal |= pal;
                s_key[dst] = ((u64)(pal ? 0u : rk[r]) << 32) | (u32)ka[r];
                if (SKB) s_b[dst] = kb[r];
                if (SKC) s_c[dst] = kc[r];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return any_pal;
}

// k_candidates_pairs on records sorted on the top 24 code bits (see above).  cnt_slots / blk_* / staged are indexed by tile.
// blk_mode (counting pass writes it): 0 the walk does the whole tile, 1 done here, 3 done here except its palindromic records.
template <bool WRITE, bool SKB, bool SKC>
__global__ __launch_bounds__(256) void k_candidates_pairs_ps(DevParams p, const u64 *__restrict__ codes, u64 n, u32 *__restrict__ cnt_slots,
                                                             u32 *__restrict__ blk_slots, u64 *__restrict__ blk_cand, const u64 *__restrict__ blk_off,
                                                             PairRec *__restrict__ slots, const u64 *__restrict__ sk_b, const u32 *__restrict__ sk_c,
                                                             u64 *__restrict__ staged, u8 *__restrict__ blk_staged, u8 *__restrict__ blk_mode,
                                                             const u32 *__restrict__ tiles)
{
    const u32 tile = tiles ? tiles[blockIdx.x] : blockIdx.x;
    __shared__ u64 s_key[PS_R];
    __shared__ u64 s_b[SKB ? PS_R : 1];
    __shared__ u32 s_c[SKC ? PS_R : 1];
    __shared__ u32 s_mk[4][VP_STRIP];                // per wave: the histogram of the bucket sort, then the pair strip
    __shared__ u64 s_heads[PS_R / 64];
    __shared__ u32 s_val[WRITE ? 1 : VT_QUOTA];
    __shared__ uint16_t s_own[WRITE ? 1 : VT_QUOTA];
    __shared__ u32 s_seg[4], s_w[4];
    __shared__ u32 s_n;
    __shared__ u32 s_flag[2];                        // a bucket above PS_MAXB records / a palindromic code in the tile
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 base = (u64)tile * PS_TILE;
    if (WRITE) {
        if (!(blk_mode[tile] & 1) || blk_slots[tile] == 0 || blk_staged[tile]) return;
    }
    if (!WRITE && tid == 0) s_n = 0;
    {
        u64 rk[PS_PER], rb[SKB ? PS_PER : 1];
        u32 rc[SKC ? PS_PER : 1];
#pragma unroll
        for (int q = 0; q < PS_PER; q++) {
            const long long g = (long long)base - 1 + (q * 256 + tid);
            const bool in = g >= 0 && (u64)g < n;
            rk[q] = in ? __builtin_nontemporal_load(codes + g) : 0ull;
            if (SKB) rb[q] = in ? __builtin_nontemporal_load(sk_b + g) : 0ull;
            if (SKC) rc[q] = in ? __builtin_nontemporal_load(sk_c + g) : 0u;
        }
#pragma unroll
        for (int q = 0; q < PS_PER; q++) {
            const int slot = q * 256 + tid;
            s_key[slot] = rk[q];
            if (SKB) s_b[slot] = rb[q];
            if (SKC) s_c[slot] = rc[q];
        }
    }
    __syncthreads();
    // ---- bucket heads of all slots, the tile's cuts ----
    if (base >= 1 && base - 1 + PS_R <= n) {
        // every slot holds a record and none is the list's first or last: a slot opens a bucket iff its bucket differs from the one
        // before it -- the lane before (one DPP shift), for lane 0 the last lane of the round before
        u32 before = wv ? ps_bucket(s_key[wv * PS_WAVE - 1]) : ps_bucket(s_key[0]);       // (slot 0 is never asked about)
#pragma unroll
        for (int r = 0; r < PS_WAVE / 64; r++) {
            const int slot = wv * PS_WAVE + r * 64 + lane;
            const u32 bk = ps_bucket(s_key[slot]);
            const u32 prev = (u32)__builtin_amdgcn_update_dpp((int)before, (int)bk, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            const u64 bal = __ballot(bk != prev);
            if (lane == 0) s_heads[slot >> 6] = bal;
            before = (u32)__builtin_amdgcn_readlane((int)bk, 63);
        }
    } else {
#pragma unroll
        for (int r = 0; r < PS_WAVE / 64; r++) {
            const int slot = wv * PS_WAVE + r * 64 + lane;
            const u64 bal = __ballot(ps_is_head(s_key, slot, (long long)base - 1 + slot, n));
            if (lane == 0) s_heads[slot >> 6] = bal;
        }
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    int f, e;
    bool cut_mid;
    ps_cuts<1>(s_heads, base, n, f, e, cut_mid);
    if (cut_mid) {                                                   // (uniform) the tile begins inside a bucket: the walk's
        if (!WRITE && tid == 0) { blk_mode[tile] = 0; blk_slots[tile] = 0; blk_staged[tile] = 0; blk_cand[tile] = 0; }
        return;
    }
    // ---- every bucket of the owned range grouped by its last code byte, by the wave its head falls to; ranks into the keys ----
    {
        const int lo_w = f > wv * PS_WAVE ? f : wv * PS_WAVE, hi_w = e < (wv + 1) * PS_WAVE ? e : (wv + 1) * PS_WAVE;
        bool any_pal = false, too_big = false;
        int h = ps_first_head(s_heads, lo_w, hi_w - 1);
        while (h >= 0) {
            int h2 = ps_first_head(s_heads, h + 1, e - 1);
            const int end = h2 < 0 ? e : h2;
            const int B = end - h;
            if (B > PS_MAXB) too_big = true;
            else if (B > 1) any_pal |= ps_sort_bucket<SKB, SKC>(s_key, s_b, s_c, s_mk[wv], h, B, lane);
            else if (lane == 0) {                                   // a bucket of one: rank 0
                const u64 k1 = s_key[h];
                any_pal |= rc_code((u32)(k1 >> 32)) == (u32)(k1 >> 32);
                s_key[h] = (u64)(u32)k1;
            }
            h = (h2 >= 0 && h2 < hi_w) ? h2 : -1;
        }
        if (too_big) s_flag[0] = 1;                                  // (every writer writes the same value)
        if (any_pal) s_flag[1] = 1;
    }
    __syncthreads();
    if (s_flag[0]) {                                                 // a bucket too large to group here: the walk's
        if (!WRITE && tid == 0) { blk_mode[tile] = 0; blk_slots[tile] = 0; blk_staged[tile] = 0; blk_cand[tile] = 0; }
        return;
    }
    const bool has_pal = s_flag[1] != 0;
    // ---- run ranks, as the grouping left them: lane l of wave w owns the PS_PER slots from f + PS_WAVE w + PS_PER l ----
    const int seg0 = f + PS_WAVE * wv;
    const int slot0 = seg0 + PS_PER * lane;
    u32 r[PS_PER];
    u32 lane_sum = 0;
#pragma unroll
    for (int j = 0; j < PS_PER; j++) {
        const int sl = slot0 + j;
        r[j] = sl < e ? (u32)(s_key[sl] >> 32) : 0u;
        lane_sum += r[j];
    }
    // ---- pairs of the wave, 64 at a time (as k_candidates_pairs) ----
    const u32 P = wave_scan_add(lane_sum);
    const u32 W = (u32)__builtin_amdgcn_readlane((int)P, 63);
    u32 pr[PS_PER];
    {
        u32 x = P - lane_sum;
#pragma unroll
        for (int j = 0; j < PS_PER; j++) { pr[j] = x; x += r[j]; }
    }
    u32 seg_pre = 0;
    if (WRITE) for (int w = 0; w < wv; w++) seg_pre += cnt_slots[(u64)tile * 256 + w];
    const u64 at0 = WRITE ? blk_off[tile] + seg_pre : 0;
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 wslots = 0, carry_o = 0, carry_P = 0;
    bool spill = false;
    // The strip of k_candidates_pairs with 16-bit markers, so that the wave's 512 bytes hold PS_STRIP = 256 places instead of 128 (every
    // strip costs a pass over the lane's PS_PER records): a marker is its record's index + 1, and WHERE it sits is its P -- the max-scan
    // runs over (place + 1) of the occupied places, and the owner's index is read from the strip at the place the scan returns.
    typedef uint16_t __attribute__((may_alias)) marker_t;       // (the strip is cleared as words and read as halves)
    marker_t *mk = reinterpret_cast<marker_t *>(s_mk[wv]);
    u32 *mk32 = s_mk[wv];
    const EarlyWords early(p.sk_early);
    for (u32 c0 = 0; c0 < W; c0 += PS_STRIP) {
#pragma unroll
        for (int x = 0; x < PS_STRIP / 128; x++) mk32[x * 64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < PS_PER; j++)
            if (r[j] && pr[j] - c0 < (u32)PS_STRIP) mk[pr[j] - c0] = (marker_t)(PS_PER * lane + j + 1);
        __builtin_amdgcn_wave_barrier();
        for (u32 t = 0; t < (u32)PS_STRIP && c0 + t < W; t += 64) {
            const u32 q = c0 + t + lane;
            const u32 here = mk[t + lane];
            const u32 v = wave_scan_max(here ? t + lane + 1u : 0u);          // place + 1 of the last marker at or before this lane's place; 0: none in this round
            const u32 o_rec = v ? (u32)mk[v - 1u] - 1u : carry_o;
            const u32 o_P = v ? c0 + v - 1u : carry_P;
            carry_o = (u32)__builtin_amdgcn_readlane((int)o_rec, 63); carry_P = (u32)__builtin_amdgcn_readlane((int)o_P, 63);
            bool pass = false;
            int li = 0, lj = 0;
            if (q < W) {
                li = seg0 + (int)o_rec;
                lj = li - (int)(q - o_P + 1u);
                pass = sketch_pass<SKB, SKC>(SKB ? s_b[li] : 0ull, SKB ? s_b[lj] : 0ull, SKC ? s_c[li] : 0u, SKC ? s_c[lj] : 0u, p.d, early);
            }
            const u64 bal = __ballot(pass);
            if (pass) {
                const u32 rank = wslots + (u32)__popcll(bal & lt_mask);
                if (WRITE) {
                    PairRec rec;                                       // (k-mer 1 = the later position: ps_sort_bucket)
                    const u32 ga = (u32)s_key[li], gb = (u32)s_key[lj];
                    rec.p1 = ga > gb ? ga : gb; rec.p2 = ga > gb ? gb : ga; rec.meta = SLOT_PLAIN;
                    slots[at0 + rank] = rec;
                } else {
                    const u32 en = rank < 4096u ? atomicAdd(&s_n, 1u) : (u32)VT_QUOTA;
                    if (en < (u32)VT_QUOTA) { s_val[WRITE ? 0 : en] = ((u32)li << 16) | (u32)lj; s_own[WRITE ? 0 : en] = (uint16_t)((wv << 12) | rank); }
                    else spill = true;
                }
            }
            wslots += (u32)__popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!WRITE) {
        if (lane == 0) { s_seg[wv] = wslots; s_w[wv] = W; }
        const bool keep = !__syncthreads_or(spill);
        const u32 total = s_seg[0] + s_seg[1] + s_seg[2] + s_seg[3];
        const bool keep_all = keep && total <= (u32)VT_QUOTA;
        if (keep_all) {
            if ((u32)tid < total) {
                const u32 en = s_val[WRITE ? 0 : tid], o = s_own[WRITE ? 0 : tid];
                u32 pre = 0;
                for (u32 w = 0; w < (o >> 12); w++) pre += s_seg[w];
                const u32 ga = (u32)s_key[en >> 16], gb = (u32)s_key[en & 0xFFFFu];
                staged[(u64)tile * VT_QUOTA + pre + (o & 4095u)] = ga > gb ? ((u64)ga << 32) | gb : ((u64)gb << 32) | ga;
            }
        } else if (tid < 4) cnt_slots[(u64)tile * 256 + tid] = s_seg[tid];
        if (tid == 0) {
            blk_mode[tile] = has_pal ? 3 : 1;
            blk_slots[tile] = total;
            blk_staged[tile] = keep_all ? 1 : 0;
            blk_cand[tile] = (u64)s_w[0] + s_w[1] + s_w[2] + s_w[3];
        }
    }
}

// The walk in PS mode: the tiles the balanced kernel left whole (blk_mode 0) and, record by record, the palindromic codes of the
// tiles it did (blk_mode 3).  A record meets the earlier records of its BUCKET -- LDS first, then global memory -- and takes
// those with its own code (HTS:2753-2762; in a bucket too large for LDS most records carry the family's code).  The same tile
// cuts as above.  Outputs at index n_tiles + tile.
template <bool WRITE, bool SKB, bool SKC>
__global__ __launch_bounds__(256) void k_candidates_ps(DevParams p, const u64 *__restrict__ codes, u64 n, u32 n_tiles, u32 *__restrict__ cnt_slots,
                                                       u32 *__restrict__ blk_slots, u64 *__restrict__ blk_cand, const u64 *__restrict__ blk_off,
                                                       PairRec *__restrict__ slots, const u64 *__restrict__ sk_b, const u32 *__restrict__ sk_c,
                                                       u64 *__restrict__ staged, u8 *__restrict__ blk_staged, const u8 *__restrict__ blk_mode,
                                                       const u32 *__restrict__ tiles)
{
    const u32 tile = tiles[blockIdx.x];
    const u64 v = (u64)n_tiles + tile;
    const u32 mode = blk_mode[tile];
    if (mode == 1) return;                                           // (nothing here for the walk; such tiles are not listed)
    const bool pal_only = mode != 0;
    __shared__ u64 s_key[PS_WR];
    __shared__ u64 s_b[SKB ? PS_WR : 1];
    __shared__ u32 s_c[SKC ? PS_WR : 1];
    __shared__ u64 s_heads[PS_WR / 64];
    __shared__ u32 s_val[WRITE ? 1 : VT_QUOTA];
    __shared__ uint16_t s_own[WRITE ? 1 : VT_QUOTA];
    __shared__ u32 s_pre[WRITE ? 1 : 256];
    __shared__ u32 s_n;
    __shared__ u32 l4[4];
    __shared__ u64 c4[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 base = (u64)tile * PS_TILE;
    u64 at = 0;
    u32 mine = 1;
    bool spill = false;
    if (!WRITE && tid == 0) s_n = 0;
    if (WRITE) {
        if (blk_slots[v] == 0 || blk_staged[v]) return;
        mine = cnt_slots[v * 256 + tid];
        at = blk_off[v] + block_excl_scan_u32(mine, l4, nullptr);
    }
    const long long rec0 = (long long)base - PS_WHALO;              // record of slot 0
    const int first_slot = rec0 < 0 ? (int)-rec0 : 0;                // first slot that holds a record
    {
        constexpr int LD = PS_WR / 256;
        u64 rk[LD], rb[SKB ? LD : 1];
        u32 rc[SKC ? LD : 1];
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const long long g = rec0 + (q * 256 + tid);
            const bool in = g >= 0 && (u64)g < n;
            rk[q] = in ? __builtin_nontemporal_load(codes + g) : 0ull;
            if (SKB) rb[q] = in ? __builtin_nontemporal_load(sk_b + g) : 0ull;
            if (SKC) rc[q] = in ? __builtin_nontemporal_load(sk_c + g) : 0u;
        }
#pragma unroll
        for (int q = 0; q < LD; q++) {
            const int slot = q * 256 + tid;
            s_key[slot] = rk[q];
            if (SKB) s_b[slot] = rb[q];
            if (SKC) s_c[slot] = rc[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PS_WR / 256; r++) {
        const int slot = wv * (PS_WR / 4) + r * 64 + lane;
        const u64 bal = __ballot(ps_is_head(s_key, slot, rec0 + slot, n));
        if (lane == 0) s_heads[slot >> 6] = bal;
    }
    __syncthreads();
    int f, e;
    bool cut_mid;
    ps_cuts<PS_WHALO>(s_heads, base, n, f, e, cut_mid);
    const bool have_sk = SKB || SKC;
    const EarlyWords early(p.sk_early);
    u32 nslot = 0;
    u64 ncand = 0;
#pragma unroll 1
    for (int k = 0; k < (mine ? PS_PER + 1 : 0); k++) {
        const int li = f + k * 256 + tid;
        if (li >= e) break;
        const u64 ci = s_key[li];
        const u32 code = (u32)(ci >> 32), gi = (u32)ci, bkt = ps_bucket(ci);
        const bool pal = rc_code(code) == code;
        if (pal_only && !pal) continue;
        const bool filter = have_sk && !pal;
        const u64 bi = (filter && SKB) ? s_b[li] : 0ull;
        const u32 qi = (filter && SKC) ? s_c[li] : 0u;
        u32 nc = 0;
        if (pal) {
            nc++;
            if (WRITE) { slots[at].p1 = gi; slots[at].p2 = gi; slots[at].meta = SLOT_PAL_SELF; at++; }
            else { nslot++; spill = true; }
        }
        auto visit = [&](u32 gj, u64 bj, u32 qj, int lj) {
            if (pal) {
                nc += 4;
                if (WRITE) for (u32 c = SLOT_PAL_FWD0; c <= SLOT_PAL_REV1; c++, at++) { slots[at].p1 = gi; slots[at].p2 = gj; slots[at].meta = c; }
                else nslot += 4;
                return;
            }
            nc++;
            if (filter && !sketch_pass<SKB, SKC>(bi, bj, qi, qj, p.d, early)) return;
            if (WRITE) { slots[at].p1 = gi; slots[at].p2 = gj; slots[at].meta = SLOT_PLAIN; at++; }
            else {
                const u32 en = (lj >= 0 && nslot < 256u) ? atomicAdd(&s_n, 1u) : (u32)VT_QUOTA;
                if (en < (u32)VT_QUOTA) { s_val[WRITE ? 0 : en] = ((u32)li << 16) | (u32)lj; s_own[WRITE ? 0 : en] = (uint16_t)((tid << 8) | nslot); }
                else spill = true;
                nslot++;
            }
        };
        // the earlier records of the bucket: in LDS ...
        int j = li - 1;
        while (j >= first_slot && ps_bucket(s_key[j]) == bkt) {
            if ((u32)(s_key[j] >> 32) == code) visit((u32)s_key[j], SKB ? s_b[j] : 0ull, SKC ? s_c[j] : 0u, j);
            j--;
        }
        // ... and, when the bucket reaches the start of the LDS copy, on into global memory
        if (j < first_slot && rec0 > 0) {
            for (u64 g = (u64)rec0; g-- > 0;) {
                const u64 cj = codes[g];
                if (ps_bucket(cj) != bkt) break;
                if ((u32)(cj >> 32) == code) visit((u32)cj, SKB ? sk_b[g] : 0ull, SKC ? sk_c[g] : 0u, -1);
            }
        }
        ncand += nc;
    }
    if (!WRITE) {
        u64 c = ncand;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0) c4[tid >> 6] = c;
        u32 total;
        const u32 pre = block_excl_scan_u32(nslot, l4, &total);
        s_pre[WRITE ? 0 : tid] = pre;
        const bool keep_all = !__syncthreads_or(spill) && total <= (u32)VT_QUOTA;
        if (keep_all) {
            if ((u32)tid < total) {
                const u32 en = s_val[WRITE ? 0 : tid], o = s_own[WRITE ? 0 : tid];
                staged[v * VT_QUOTA + s_pre[WRITE ? 0 : (o >> 8)] + (o & 255u)] = ((u64)(u32)s_key[en >> 16] << 32) | (u32)s_key[en & 0xFFFFu];
            }
        } else cnt_slots[v * 256 + tid] = nslot;
        if (tid == 0) {
            blk_slots[v] = total;
            blk_staged[v] = keep_all ? 1 : 0;
            blk_cand[v] = c4[0] + c4[1] + c4[2] + c4[3];
        }
    }
}

// the kept slots of the tiles the counting pass could keep whole: staging array -> slot array
__global__ __launch_bounds__(VT_QUOTA) void k_staged_slots(const u64 *__restrict__ staged, const u32 *__restrict__ blk_slots, const u8 *__restrict__ blk_staged,
                                                           const u64 *__restrict__ blk_off, PairRec *__restrict__ slots)
{
    if (!blk_staged[blockIdx.x] || threadIdx.x >= blk_slots[blockIdx.x]) return;
    const u64 e = staged[(u64)blockIdx.x * VT_QUOTA + threadIdx.x];
    PairRec r;
    r.p1 = (u32)(e >> 32); r.p2 = (u32)e; r.meta = SLOT_PLAIN;
    slots[blk_off[blockIdx.x] + threadIdx.x] = r;
}

// One thread per slot: both K-mers from the packed genome, CompareAndReport; meta in place of the kind.  Also counts the
// reported pairs per block of 256 slots (for the compaction).
__global__ __launch_bounds__(256) void k_judge(DevParams p, const u64 *__restrict__ pk, PairRec *__restrict__ slots, u64 n_slots, int seed,
                                               u32 *__restrict__ blk_valid)
{
    __shared__ u32 l4[4];
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 meta = SLOT_NONE;
    if (t < n_slots) {
        const PairRec r = slots[t];
        u64 ih, il, jh, jl, rh, rl;
        load_window(p, pk, r.p1, ih, il);
        if (r.meta == SLOT_PAL_SELF) { jh = ih; jl = il; } else load_window(p, pk, r.p2, jh, jl);
        rc_window(p.K, jh, jl, rh, rl);
        const Mism same = mism_of(ih, il, jh, jl), rev = mism_of(ih, il, rh, rl);
        switch (r.meta) {
        case SLOT_PLAIN: {
            const bool same_ok = ((same.hi & p.seedmask_hi[seed - 1]) | (same.lo & p.seedmask_lo[seed - 1])) == 0;
            meta = same_ok ? judge(p, same, seed, 0) : judge(p, rev, seed, 1);
            break;
        }
        case SLOT_PAL_FWD0: case SLOT_PAL_FWD1:
            meta = judge(p, same, seed, 0);
            if (meta != SLOT_NONE) meta |= (r.meta - SLOT_PAL_FWD0) << 25;
            break;
        case SLOT_PAL_REV0: case SLOT_PAL_REV1:
            meta = judge(p, rev, seed, 1);
            if (meta != SLOT_NONE) meta |= (1u << 24) | ((r.meta - SLOT_PAL_REV0) << 25);
            break;
        default:                                                // SLOT_PAL_SELF
            meta = judge(p, rev, seed, 1);
            if (meta != SLOT_NONE) meta |= 1u << 24;
            break;
        }
        slots[t].meta = meta;
    }
    const u64 bal = __ballot(meta != SLOT_NONE);
    if ((threadIdx.x & 63) == 0) l4[threadIdx.x >> 6] = (u32)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) blk_valid[blockIdx.x] = l4[0] + l4[1] + l4[2] + l4[3];
}

__global__ __launch_bounds__(256) void k_slot_compact(const PairRec *__restrict__ slots, u64 n_slots, const u64 *__restrict__ blk_off,
                                                      PairRec *__restrict__ pairs)
{
    __shared__ u32 l4[4];
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    PairRec r;
    r.meta = SLOT_NONE;
    if (t < n_slots) r = slots[t];
    const bool ok = r.meta != SLOT_NONE;
    const u32 pre = block_excl_scan_u32(ok ? 1u : 0u, l4, nullptr);
    if (ok) pairs[blk_off[blockIdx.x] + pre] = r;
}

// out += sum of in[0..n)  (few blocks, one atomic each)
__global__ __launch_bounds__(256) void k_sum_u64(const u64 *__restrict__ in, u64 n, unsigned long long *out)
{
    __shared__ u64 c4[4];
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) s += in[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) c4[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { s = c4[0] + c4[1] + c4[2] + c4[3]; if (s) atomicAdd(out, (unsigned long long)s); }
}

// ------------------------------------------------------------------ directed items
// Each pair gives item 1->2 and item 2->1; for a reverse homology the 2->1 offsets are mirrored
// and re-sorted (HTS:1022-1103).
__global__ __launch_bounds__(256) void k_items_expand(DevParams p, const PairRec *__restrict__ pairs, u64 n_pairs,
                                                      u64 *__restrict__ item_key, u32 *__restrict__ item_meta)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const PairRec r = pairs[i];
    item_key[2 * i] = ((u64)r.p1 << 32) | r.p2;
    item_meta[2 * i] = r.meta;
    u32 m2 = r.meta;
    if (r.meta & META_DIR) {
        const u32 o1 = meta_o1(r.meta), o2 = meta_o2(r.meta), o3 = meta_o3(r.meta), K = (u32)p.K;
        u32 n1, n2 = 0, n3 = 0;
        if (o2 == 0) n1 = K - o1 + 1;
        else if (o3 == 0) { n1 = K - o2 + 1; n2 = K - o1 + 1; }
        else { n1 = K - o3 + 1; n2 = K - o2 + 1; n3 = K - o1 + 1; }
        m2 = (r.meta & ~(META_MATCH_MASK & ~1u)) | (n1 << 1) | (n2 << 8) | (n3 << 15);
    }
    item_key[2 * i + 1] = ((u64)r.p2 << 32) | r.p1;
    item_meta[2 * i + 1] = m2;
}

// List heads of the sorted items as a bit mask (bit l of word w = item 64 w + l opens the list of a new source position) and
// the number of heads per word; scanned, that gives every word its first list, and with a popcount every item its list.
// (Round 2 first kept a u32 flag and a u64 scanned list index per item: 6 GB written and read back at 5e8 items.)
// 256 items per wave in four rounds, all loads in flight together.
__global__ __launch_bounds__(256) void k_node_flags(const u64 *__restrict__ item_key, u64 m, u64 *__restrict__ head_mask, u32 *__restrict__ word_heads,
                                                    DevCounters *ctr)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 wbase = ((u64)blockIdx.x * 4 + wv) * 256;
    if (wbase >= m) return;                                    // (wave-uniform)
    u64 k[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 i = wbase + (u64)r * 64 + lane;
        k[r] = i < m ? item_key[i] : 0ull;
    }
    u64 before = 0;
    if (lane == 0 && wbase > 0) before = item_key[wbase - 1];
    bool tie = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 i = wbase + (u64)r * 64 + lane;
        u64 prev = __shfl_up(k[r], 1, 64);
        const u64 tail = r ? __shfl(k[r ? r - 1 : 0], 63, 64) : before;
        if (lane == 0) prev = tail;
        const bool head = i < m && (i == 0 || (prev >> 32) != (k[r] >> 32));
        tie = tie || (i < m && i > 0 && prev == k[r]);
        const u64 bal = __ballot(head);
        if (lane == 0 && wbase + (u64)r * 64 < m) { head_mask[(wbase >> 6) + r] = bal; word_heads[(wbase >> 6) + r] = (u32)__popcll(bal); }
    }
    if (tie) ctr->n_dirty = 1;                                 // any tie: a flag, not a count
}

// node_start[list] = its first item: one thread per mask word, word_first[w] = lists that open before word w
__global__ __launch_bounds__(256) void k_node_starts(const u64 *__restrict__ head_mask, const u64 *__restrict__ word_first, u64 n_words,
                                                     u32 *__restrict__ node_start)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    u64 bits = head_mask[w];
    u64 at = word_first[w];
    while (bits) {
        node_start[at++] = (u32)(w * 64 + (u64)(__ffsll((long long)bits) - 1));
        bits &= bits - 1;
    }
}

__device__ __forceinline__ bool same_window(const DevParams &p, u64 a, u64 b)
{
    if (!p.any_windowed) return true;
    const u32 c = find_contig(p, a);
    const u32 w = p.ctg_window[c];
    if (!w) return true;
    const u64 s = p.ctg_gstart[c];
    return (a - s) / w == (b - s) / w;
}

// index of the item (src, tgt) of node `nd` whose dir/offsets equal `want`, or NO_ITEM
constexpr u32 NO_ITEM = 0xFFFFFFFFu;
__device__ __forceinline__ u32 find_item(const u64 *__restrict__ item_key, const u32 *__restrict__ item_meta,
                                         const u32 *__restrict__ node_start, u32 nd, u64 key, u32 want)
{
    u32 a = node_start[nd], b = node_start[nd + 1];
    const u32 end = b;
    while (a < b) {
        const u32 mid = (a + b) >> 1;
        if (item_key[mid] < key) a = mid + 1; else b = mid;
    }
    for (; a < end && item_key[a] == key; a++)
        if ((item_meta[a] & META_MATCH_MASK) == want) return a;
    return NO_ITEM;
}

// Diagonal block merge (HTS:1203-1270) for lists without tied targets, as links: the successor of an item is the item
// of the NEXT position's list that continues its diagonal (target one base further along its strand, same direction,
// every offset one smaller, same compaction window).  An item that is somebody's successor is absorbed (block length -1);
// a surviving item's block length is the length of the chain it heads (k_chain_walk).  One search per item instead of
// one per chain step.  Lists with tied targets are redone by the host in the reference's order.
__global__ __launch_bounds__(256) void k_chain_link(DevParams p, const u64 *__restrict__ item_key, const u32 *__restrict__ item_meta,
                                                    const u64 *__restrict__ head_mask, const u64 *__restrict__ word_first,
                                                    const u32 *__restrict__ node_start, u32 n_nodes, u64 m, u32 *__restrict__ succ,
                                                    int16_t *__restrict__ blen)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    // the item's list: lists opened before its mask word + heads of the word up to and including the item - 1
    const u32 nd = (u32)(word_first[i >> 6] + (u64)__popcll(head_mask[i >> 6] & ((2ull << (i & 63)) - 1ull)) - 1ull);
    const u64 key = item_key[i];
    const u64 src = key >> 32, tgt = key & 0xFFFFFFFFull;
    const u32 meta = item_meta[i];
    const u32 dir = meta & 1u, o1 = meta_o1(meta), o2 = meta_o2(meta), o3 = meta_o3(meta);
    u32 s = NO_ITEM;
    const u32 nn = nd + 1;
    if (o1 > 1 && nn < n_nodes && (item_key[node_start[nn]] >> 32) == src + 1 && same_window(p, src, src + 1) && !(dir && tgt < 1)) {
        const u64 nt = dir ? tgt - 1 : tgt + 1;
        const u32 want = meta_pack(dir, o1 - 1, o2 ? o2 - 1 : 0, o3 ? o3 - 1 : 0, 0, 0, 0);
        s = find_item(item_key, item_meta, node_start, nn, ((src + 1) << 32) | nt, want);
    }
    succ[i] = s;
    if (s != NO_ITEM) blen[s] = -1;
}

// The items that survived (were absorbed by nobody) head a chain each; its length is their block length.  They are one item
// in eight, so they are listed first (k_alive_mask + the mask kernels) and the walk runs one thread per listed item: with a
// thread per item a wave held seven walkers and fifty-seven lanes that had left (7.9 ms at 5e8 items).
__global__ __launch_bounds__(256) void k_alive_mask(const int16_t *__restrict__ blen, u64 m, u64 *__restrict__ mask)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 wbase = ((u64)blockIdx.x * 4 + wv) * 256;      // 256 items per wave in four rounds, the loads in flight together
    if (wbase >= m) return;
    int16_t b[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 i = wbase + (u64)r * 64 + lane;
        b[r] = i < m ? blen[i] : (int16_t)-1;                // (blen was zeroed before k_chain_link; only absorbed items hold -1)
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 bal = __ballot(b[r] != -1);
        if (lane == 0 && wbase + (u64)r * 64 < m) mask[(wbase >> 6) + r] = bal;
    }
}

__global__ __launch_bounds__(256) void k_chain_walk(const u32 *__restrict__ succ, const u32 *__restrict__ alive, u64 n_alive, int16_t *__restrict__ blen)
{
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_alive) return;
    const u32 i = alive[t];
    u32 L = 1;
    for (u32 j = succ[i]; j != NO_ITEM; j = succ[j]) L++; // at most K steps: offset 1 shrinks by one per link
    blen[i] = (int16_t)L;
}

// ------------------------------------------------------------------ libstdc++'s std::sort, on the device
// The reference orders every list by decreasing block length with std::sort (HTS:1368, 1110-1113).  Above 16 items that
// is introsort -- median-of-3 quicksort to depth 2 lg n, heapsort below that, one insertion pass at the end (GCC
// bits/stl_algo.h, bits/stl_heap.h) -- which is not stable, so the order of equal block lengths is whatever exactly that
// algorithm makes of the incoming (target) order.  It is deterministic, so one thread per list runs exactly that
// algorithm.  Elements are u32: (block length + 1) << 25 | index in the list; only the top 7 bits are compared.
// The recursion of __introsort_loop on the right part becomes an explicit stack: the parts are disjoint, so the order in
// which they are finished does not matter.  Restated on the CPU in oracle/hts_oracle.c (ss_*), which tests pin against
// the real std::sort; tests/test_gpu_parity.py compares this one with it through cora_hip_test_list_sort.
constexpr int DS_KEY_SHIFT = 25;
__device__ __forceinline__ bool ds_less(u32 x, u32 y) { return (x >> DS_KEY_SHIFT) > (y >> DS_KEY_SHIFT); }
__device__ __forceinline__ void ds_swap(u32 *e, long a, long b) { const u32 t = e[a]; e[a] = e[b]; e[b] = t; }

__device__ void ds_adjust_heap(u32 *e, long hole, long len, u32 value)       // __adjust_heap + __push_heap
{
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (ds_less(e[second], e[second - 1])) second--;
        e[hole] = e[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        e[hole] = e[second - 1];
        hole = second - 1;
    }
    long parent = (hole - 1) / 2;
    while (hole > top && ds_less(e[parent], value)) {
        e[hole] = e[parent];
        hole = parent; parent = (hole - 1) / 2;
    }
    e[hole] = value;
}

__device__ void ds_heap_sort(u32 *e, long len)                                // __partial_sort(first, last, last)
{
    if (len >= 2) {
        for (long parent = (len - 2) / 2;; parent--) {
            ds_adjust_heap(e, parent, len, e[parent]);
            if (parent == 0) break;
        }
    }
    for (long last = len; last > 1;) {
        --last;
        const u32 value = e[last];
        e[last] = e[0];
        ds_adjust_heap(e, 0, last, value);
    }
}

__device__ __forceinline__ void ds_unguarded_linear_insert(u32 *e, long last)
{
    const u32 val = e[last];
    long next = last - 1;
    while (ds_less(val, e[next])) { e[last] = e[next]; last = next; --next; }
    e[last] = val;
}

__device__ void ds_insertion_sort(u32 *e, long first, long last)
{
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (ds_less(e[i], e[first])) {
            const u32 val = e[i];
            for (long k = i; k > first; k--) e[k] = e[k - 1];
            e[first] = val;
        } else ds_unguarded_linear_insert(e, i);
    }
}

__device__ void ds_std_sort(u32 *e, long n, long forced_depth)
{
    if (n <= 0) return;
    long lg = 0;
    for (unsigned long v = (unsigned long)n; v >>= 1;) lg++;
    struct Part { long first, last, depth; } stack[72];                      // one pending right part per quicksort level (<= 2 lg n)
    int sp = 0;
    stack[sp++] = {0, n, forced_depth >= 0 ? forced_depth : 2 * lg};
    while (sp) {
        const Part pt = stack[--sp];
        long first = pt.first, last = pt.last, depth = pt.depth;
        while (last - first > 16) {
            if (depth == 0) { ds_heap_sort(e + first, last - first); break; }
            --depth;
            const long mid = first + (last - first) / 2, a = first + 1, c = last - 1;
            if (ds_less(e[a], e[mid])) {                                      // __move_median_to_first(first, first+1, mid, last-1)
                if (ds_less(e[mid], e[c])) ds_swap(e, first, mid);
                else if (ds_less(e[a], e[c])) ds_swap(e, first, c);
                else ds_swap(e, first, a);
            } else if (ds_less(e[a], e[c])) ds_swap(e, first, a);
            else if (ds_less(e[mid], e[c])) ds_swap(e, first, c);
            else ds_swap(e, first, mid);
            long lo = first + 1, hi = last;                                   // __unguarded_partition(first+1, last, first)
            for (;;) {
                while (ds_less(e[lo], e[first])) ++lo;
                --hi;
                while (ds_less(e[first], e[hi])) --hi;
                if (!(lo < hi)) break;
                ds_swap(e, lo, hi);
                ++lo;
            }
            stack[sp++] = {lo, last, depth};
            last = lo;
        }
    }
    if (n > 16) {                                                             // __final_insertion_sort
        ds_insertion_sort(e, 0, 16);
        for (long i = 16; i != n; ++i) ds_unguarded_linear_insert(e, i);
    } else ds_insertion_sort(e, 0, n);
}

__global__ void k_test_list_sort(u32 *e, u64 n, long forced_depth)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ds_std_sort(e, (long)n, forced_depth);
}
void launch_test_list_sort(hipStream_t s, u32 *e, u64 n, long forced_depth) { hipLaunchKernelGGL(k_test_list_sort, dim3(1), dim3(64), 0, s, e, n, forced_depth); }

// ------------------------------------------------------------------ <Inexact> bytes on the device
// Lists (nodes) are written by the GPU: a list of at most 16 items goes through libstdc++'s std::sort as a plain
// insertion sort, i.e. stably, so "by decreasing block length, ties in target order" is the reference's order
// (HTS:1368); a longer one is ordered by the device replica of that std::sort above.  Only the lists of a run of
// consecutive positions that holds tied targets (the block lengths themselves then depend on the reference's push
// order) are left to the host, which receives just those lists.

// per list: does it hold a tied target; does it open a run of consecutive source positions (HTS:1212)
__global__ __launch_bounds__(256) void k_node_class(DevParams p, const u64 *__restrict__ item_key, const u32 *__restrict__ node_start,
                                                    u32 n_nodes, u32 *__restrict__ run_head, u32 *__restrict__ node_tie)
{
    const u32 nd = blockIdx.x * 256 + threadIdx.x;
    if (nd >= n_nodes) return;
    const u32 a = node_start[nd], b = node_start[nd + 1];
    bool tie = false;
    u64 prev = item_key[a];
    for (u32 k = a + 1; k < b; k++) { const u64 x = item_key[k]; tie |= (x == prev); prev = x; }
    const u64 src = item_key[a] >> 32;
    bool head = true;
    if (nd > 0 && src > 0 && (item_key[node_start[nd - 1]] >> 32) == src - 1 && same_window(p, src - 1, src)) head = false;
    run_head[nd] = head ? 1u : 0u;
    node_tie[nd] = tie ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_run_dirty(const u32 *__restrict__ node_tie, const u64 *__restrict__ run_excl,
                                                   const u32 *__restrict__ run_head, u32 n_nodes, u32 *__restrict__ run_dirty)
{
    const u32 nd = blockIdx.x * 256 + threadIdx.x;
    if (nd < n_nodes && node_tie[nd]) run_dirty[run_excl[nd] + run_head[nd] - 1] = 1u;
}

// bytes of every device-written list (0 for host lists and for lists whose items were all absorbed).  A block takes 256
// consecutive lists: their items are one contiguous range, which the whole block reads coalesced; an item that survived the
// block merge finds its list by a search over the 257 list starts in LDS.  (One thread walking each list took 7.2 ms at
// 3.6e7 lists: a wave waited for its longest list, and lists are 1 to thousands of items long.)
__global__ __launch_bounds__(256) void k_node_size(DevParams p, int cw, const u32 *__restrict__ node_start, const int16_t *__restrict__ blen,
                                                   const u64 *__restrict__ run_excl, const u32 *__restrict__ run_head,
                                                   const u32 *__restrict__ run_dirty, u32 n_nodes, u32 *__restrict__ size,
                                                   u32 *__restrict__ hflag, u32 *__restrict__ hcnt, u32 *__restrict__ blk_counts /* [2 * blocks] */)
{
    __shared__ u32 s_start[257], s_alive[256];
    __shared__ u32 s_emit[4], s_big[4];
    const u32 n0 = blockIdx.x * 256;
    const u32 nd = n0 + threadIdx.x;
    const u32 n_here = n_nodes - n0 < 256u ? n_nodes - n0 : 256u;
    // the dependent look-up of this list's run first, so that it is in flight beside the item loop
    bool dirty = false;
    if (nd < n_nodes) dirty = run_dirty[run_excl[nd] + run_head[nd] - 1] != 0;
    s_alive[threadIdx.x] = 0;
    if (threadIdx.x <= n_here) s_start[threadIdx.x] = node_start[n0 + threadIdx.x];
    if (threadIdx.x == 0 && n_here == 256u) s_start[256] = node_start[n0 + 256];
    __syncthreads();
    const u32 A = s_start[0], B = s_start[n_here];
    for (u32 k = A + threadIdx.x; k < B; k += 256) {
        if (blen[k] == -1) continue;
        u32 lo = 0, hi = n_here;                               // the list j with s_start[j] <= k < s_start[j + 1]
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_start[mid] <= k) lo = mid; else hi = mid; }
        atomicAdd(&s_alive[lo], 1u);
    }
    __syncthreads();
    bool emitted = false, big = false;
    if (nd < n_nodes) {
        const u32 cnt = s_start[threadIdx.x + 1] - s_start[threadIdx.x];
        const bool host = dirty || cnt >= (1u << DS_KEY_SHIFT);
        big = cnt > 16;
        u32 sz = 0;
        if (!host) {
            const u32 alive = s_alive[threadIdx.x];
            if (alive) sz = (u32)(cw + 4) + alive * (u32)(1 + cw + 4 + p.d + 2) + 1u;
            emitted = alive != 0;
        }
        size[nd] = sz;
        hflag[nd] = host ? 1u : 0u;
        hcnt[nd] = host ? cnt : 0u;
    }
    const u64 bal = __ballot(emitted), bbig = __ballot(big);
    if ((threadIdx.x & 63) == 0) { s_emit[threadIdx.x >> 6] = (u32)__popcll(bal); s_big[threadIdx.x >> 6] = (u32)__popcll(bbig); }
    __syncthreads();
    if (threadIdx.x == 0) {                                    // per-block partials: two same-address atomics per block were half of this kernel's 7.2 ms
        blk_counts[blockIdx.x] = s_emit[0] + s_emit[1] + s_emit[2] + s_emit[3];       // (also what k_node_emit_list's offsets are scanned from)
        blk_counts[gridDim.x + blockIdx.x] = s_big[0] + s_big[1] + s_big[2] + s_big[3];
    }
}

// The lists the device writes (size != 0), listed in list order: block b's go to list[blk_off[b] ..] (blk_off = the scanned per-block
// counts of k_node_size).  Three lists in four have had all their items absorbed by the block merge and are not written: k_node_write
// over ALL lists ran with a quarter of its lanes busy, and its time is latency (contig searches, byte stores), i.e. per wave.
__global__ __launch_bounds__(256) void k_node_emit_list(const u32 *__restrict__ size, u32 n_nodes, const u64 *__restrict__ blk_off, u32 *__restrict__ list)
{
    __shared__ u32 s_w[4];
    const u32 nd = blockIdx.x * 256 + threadIdx.x;
    const bool e = nd < n_nodes && size[nd] != 0;
    const u64 bal = __ballot(e);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_w[wv] = (u32)__popcll(bal);
    __syncthreads();
    if (!e) return;
    u32 pre = 0;
    for (int w = 0; w < wv; w++) pre += s_w[w];
    list[blk_off[blockIdx.x] + pre + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = nd;
}

// emitted_nodes / big_nodes += the per-block pairs of k_node_size (one atomic per field and workgroup of a 64-workgroup grid)
__global__ __launch_bounds__(256) void k_node_counts(const u32 *__restrict__ blk_counts, u32 n_blocks, DevCounters *ctr)
{
    __shared__ unsigned long long s_e[4], s_b[4];
    unsigned long long e = 0, b = 0;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n_blocks; i += gridDim.x * 256) { e += blk_counts[i]; b += blk_counts[n_blocks + i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = e; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        e = s_e[0] + s_e[1] + s_e[2] + s_e[3]; b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        if (e) atomicAdd(&ctr->emitted_nodes, e);
        if (b) atomicAdd(&ctr->big_nodes, b);
    }
}

__device__ __forceinline__ u8 *put_chr(u8 *w, int cw, u32 c) { *w++ = (u8)c; if (cw == 2) *w++ = (u8)(c >> 8); return w; }
__device__ __forceinline__ u8 *put_u32(u8 *w, u32 x) { w[0] = (u8)x; w[1] = (u8)(x >> 8); w[2] = (u8)(x >> 16); w[3] = (u8)(x >> 24); return w + 4; }

// F2 bytes of the device lists (HTS:1353-1432); host lists: descriptor + their items, compacted
__global__ __launch_bounds__(256) void k_node_write(DevParams p, int cw, const u64 *__restrict__ item_key, const u32 *__restrict__ item_meta,
                                                    const int16_t *__restrict__ blen, const u32 *__restrict__ node_start, u32 n_nodes,
                                                    const u32 *__restrict__ size, const u64 *__restrict__ off, const u32 *__restrict__ hflag,
                                                    const u64 *__restrict__ hidx, const u64 *__restrict__ hoff, const u32 *__restrict__ node_tie,
                                                    u8 *__restrict__ stream, HostNode *__restrict__ hnodes, u64 *__restrict__ h_key,
                                                    u32 *__restrict__ h_meta, int16_t *__restrict__ h_blen, u32 *__restrict__ ord, u32 node_lo,
                                                    int which /* 1: the lists the host finishes, 2: the others, 3: all */,
                                                    const u32 *__restrict__ elist /* null, or k_node_emit_list's: node_lo / n_nodes are then places in it */)
{
    u32 nd = node_lo + blockIdx.x * 256 + threadIdx.x;                 // lists node_lo .. n_nodes-1
    if (nd >= n_nodes) return;
    if (elist) nd = elist[nd];
    const bool for_host = hflag[nd] != 0;
    if (!(which & (for_host ? 1 : 2))) return;
    const u32 a = node_start[nd], b = node_start[nd + 1];
    if (for_host) {
        HostNode h;
        h.node = nd; h.count = b - a; h.tie = node_tie[nd]; h.reserved = 0;
        h.stream_off = off[nd]; h.item_off = hoff[nd];
        hnodes[hidx[nd]] = h;
        for (u32 k = a; k < b; k++) {
            const u64 o = h.item_off + (k - a);
            h_key[o] = item_key[k]; h_meta[o] = item_meta[k]; h_blen[o] = blen[k];
        }
        return;
    }
    if (!size[nd]) return;
    u8 *w = stream + off[nd];
    const u64 src = item_key[a] >> 32;
    const u32 sc = find_contig(p, src);
    w = put_chr(w, cw, sc + 1);
    w = put_u32(w, (u32)(src - p.ctg_gstart[sc]) + 1);
    if (b - a > 16) {                                    // std::sort's unstable regime: run the same algorithm on (length, index)
        u32 *e = ord + a;
        const u32 cnt = b - a;
        for (u32 k = 0; k < cnt; k++) e[k] = ((u32)(blen[a + k] + 1) << DS_KEY_SHIFT) | k;
        ds_std_sort(e, (long)cnt, -1);
        for (u32 k = 0; k < cnt; k++) {
            const u32 x = e[k];
            if ((x >> DS_KEY_SHIFT) == 0) break;           // absorbed items (length -1) sort to the end (HTS:1376)
            const u32 it = a + (x & ((1u << DS_KEY_SHIFT) - 1u));
            const int L = (int)(x >> DS_KEY_SHIFT) - 1;
            const u32 m = item_meta[it];
            const u64 tgt = item_key[it] & 0xFFFFFFFFull;
            const u32 tc = find_contig(p, tgt);
            *w++ = (u8)(m & 1u);
            w = put_chr(w, cw, tc + 1);
            w = put_u32(w, (u32)(tgt - p.ctg_gstart[tc]) + 1);
            *w++ = (u8)meta_o1(m);
            if (p.d > 1) *w++ = (u8)meta_o2(m);
            if (p.d > 2) *w++ = (u8)meta_o3(m);
            *w++ = (u8)L; *w++ = (u8)((u32)L >> 8);
        }
        *w++ = 0xFF;
        return;
    }
    // by decreasing block length, ties in list (= target) order: pick the successor of the last one written.  The list's
    // (at most 16) block lengths are fetched once, all loads in flight together; every pick then runs over registers
    // (re-reading them from memory per pick was a third of this kernel)
    int Ls[16];
#pragma unroll
    for (int j = 0; j < 16; j++) Ls[j] = (a + (u32)j < b) ? (int)blen[a + j] : -1;
    int last_len = 32767 + 1;
    u32 last_k = 0;
    for (;;) {
        int best_len = -2;
        u32 best_k = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int L = Ls[j];
            const u32 k = a + (u32)j;
            const bool after = (L < last_len) || (L == last_len && k > last_k);
            if (L != -1 && after && L > best_len) { best_len = L; best_k = k; }      // first k wins among equal lengths
        }
        if (best_len == -2) break;
        const u32 m = item_meta[best_k];
        const u64 tgt = item_key[best_k] & 0xFFFFFFFFull;
        const u32 tc = find_contig(p, tgt);
        *w++ = (u8)(m & 1u);
        w = put_chr(w, cw, tc + 1);
        w = put_u32(w, (u32)(tgt - p.ctg_gstart[tc]) + 1);
        *w++ = (u8)meta_o1(m);
        if (p.d > 1) *w++ = (u8)meta_o2(m);
        if (p.d > 2) *w++ = (u8)meta_o3(m);
        *w++ = (u8)best_len; *w++ = (u8)((u32)best_len >> 8);
        last_len = best_len; last_k = best_k;
    }
    *w++ = 0xFF;
}


// ------------------------------------------------------------------ <Exact> bytes on the device
// F1 (HTS:572-635): per block of merged classes, the first class only:  chr pos { dir chr pos }*  0xFF  blockSize(u32)
__global__ __launch_bounds__(256) void k_exact_size(int cw, const u32 *__restrict__ cls_entries, const u8 *__restrict__ merge_prev, u32 n_cls,
                                                    u32 *__restrict__ size, DevCounters *ctr)
{
    __shared__ u32 s_emit[4];
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    bool head = false;
    if (c < n_cls) {
        head = !merge_prev[c];
        const u32 ne = cls_entries[c];
        size[c] = head ? ne * (u32)(cw + 4) + (ne - 1) + 5u : 0u;
    }
    const u64 bal = __ballot(head);
    if ((threadIdx.x & 63) == 0) s_emit[threadIdx.x >> 6] = (u32)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = s_emit[0] + s_emit[1] + s_emit[2] + s_emit[3];
        if (t) atomicAdd(&ctr->emitted_exact, (unsigned long long)t);
    }
}

__global__ __launch_bounds__(256) void k_exact_write(DevParams p, int cw, const u32 *__restrict__ cls_entries, const u64 *__restrict__ ent_off,
                                                     const u32 *__restrict__ ent, const u8 *__restrict__ ent_dir,
                                                     const u8 *__restrict__ merge_prev, u32 n_cls, const u32 *__restrict__ size,
                                                     const u64 *__restrict__ off, u8 *__restrict__ stream, DevCounters *ctr)
{
    __shared__ u32 s_big[256];
    __shared__ u32 s_nbig;
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c < n_cls && size[c]) {
        const u32 ne = cls_entries[c];
        if (ne > HUGE_MIN && note_huge(&ctr->huge_write, c)) {
            // left to k_exact_write_huge
        } else if (ne > COOP_MIN) {
            s_big[atomicAdd(&s_nbig, 1u)] = c;
        } else {
            u8 *w = stream + off[c];
            const u64 o = ent_off[c];
            for (u32 k = 0; k < ne; k++) {
                const u64 gp = ent[o + k];
                const u32 ci = find_contig(p, gp);
                if (k) *w++ = ent_dir[o + k];
                w = put_chr(w, cw, ci + 1);
                w = put_u32(w, (u32)(gp - p.ctg_gstart[ci]) + 1);
            }
            *w++ = 0xFF;
            u32 e = c + 1;
            while (e < n_cls && merge_prev[e]) e++;
            w = put_u32(w, e - c);
        }
    }
    __syncthreads();
    const u32 nbig = s_nbig;
    for (u32 b = 0; b < nbig; b++) {                             // a long record: entry k sits at a computable offset
        const u32 cb = s_big[b];
        u8 *w0 = stream + off[cb];
        const u64 o = ent_off[cb];
        const u32 ne = cls_entries[cb];
        for (u32 k = threadIdx.x; k < ne; k += 256) {
            const u64 gp = ent[o + k];
            const u32 ci = find_contig(p, gp);
            u8 *w = k ? w0 + (cw + 4) + (u64)(k - 1) * (cw + 5) : w0;
            if (k) *w++ = ent_dir[o + k];
            w = put_chr(w, cw, ci + 1);
            put_u32(w, (u32)(gp - p.ctg_gstart[ci]) + 1);
        }
        if (threadIdx.x == 0) {
            u8 *w = w0 + (cw + 4) + (u64)(ne - 1) * (cw + 5);
            *w++ = 0xFF;
            u32 e = cb + 1;
            while (e < n_cls && merge_prev[e]) e++;
            put_u32(w, e - cb);
        }
    }
}

__global__ __launch_bounds__(256) void k_exact_write_huge(DevParams p, int cw, const u32 *__restrict__ cls_entries, const u64 *__restrict__ ent_off,
                                                          const u32 *__restrict__ ent, const u8 *__restrict__ ent_dir,
                                                          const u8 *__restrict__ merge_prev, u32 n_cls, const u64 *__restrict__ off,
                                                          u8 *__restrict__ stream, const DevCounters *ctr)
{
    const u32 nh = ctr->huge_write.n < 31u ? ctr->huge_write.n : 31u;
    for (u32 h = 0; h < nh; h++) {
        const u32 cb = ctr->huge_write.id[h];
        u8 *w0 = stream + off[cb];
        const u64 o = ent_off[cb];
        const u32 ne = cls_entries[cb];
        for (u32 k = blockIdx.x * 256 + threadIdx.x; k < ne; k += gridDim.x * 256) {
            const u64 gp = ent[o + k];
            const u32 ci = find_contig(p, gp);
            u8 *w = k ? w0 + (cw + 4) + (u64)(k - 1) * (cw + 5) : w0;
            if (k) *w++ = ent_dir[o + k];
            w = put_chr(w, cw, ci + 1);
            put_u32(w, (u32)(gp - p.ctg_gstart[ci]) + 1);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            u8 *w = w0 + (cw + 4) + (u64)(ne - 1) * (cw + 5);
            *w++ = 0xFF;
            u32 e = cb + 1;
            while (e < n_cls && merge_prev[e]) e++;
            put_u32(w, e - cb);
        }
    }
}

// ------------------------------------------------------------------ launch wrappers
static inline dim3 grid_for(u64 n) { return dim3((unsigned)((n + 255) / 256)); }

void launch_exact_size(hipStream_t s, int cw, const u32 *cls_entries, const u8 *merge_prev, u32 n_cls, u32 *size, DevCounters *ctr)
{
    if (n_cls) hipLaunchKernelGGL(k_exact_size, grid_for(n_cls), dim3(256), 0, s, cw, cls_entries, merge_prev, n_cls, size, ctr);
}
void launch_exact_write(hipStream_t s, const DevParams &p, int cw, const u32 *cls_entries, const u64 *ent_off, const u32 *ent, const u8 *ent_dir,
                        const u8 *merge_prev, u32 n_cls, const u32 *size, const u64 *off, u8 *stream, DevCounters *ctr)
{
    if (!n_cls) return;
    (void)hipMemsetAsync(&ctr->huge_write, 0, sizeof(DevCounters::Huge), s);
    hipLaunchKernelGGL(k_exact_write, grid_for(n_cls), dim3(256), 0, s, p, cw, cls_entries, ent_off, ent, ent_dir, merge_prev, n_cls,
                       size, off, stream, ctr);
    hipLaunchKernelGGL(k_exact_write_huge, dim3(HUGE_GRID), dim3(256), 0, s, p, cw, cls_entries, ent_off, ent, ent_dir, merge_prev, n_cls,
                       off, stream, ctr);
}
void launch_node_class(hipStream_t s, const DevParams &p, const u64 *item_key, const u32 *node_start, u32 n_nodes, u32 *run_head, u32 *node_tie)
{
    if (n_nodes) hipLaunchKernelGGL(k_node_class, grid_for(n_nodes), dim3(256), 0, s, p, item_key, node_start, n_nodes, run_head, node_tie);
}
void launch_run_dirty(hipStream_t s, const u32 *node_tie, const u64 *run_excl, const u32 *run_head, u32 n_nodes, u32 *run_dirty)
{
    if (n_nodes) hipLaunchKernelGGL(k_run_dirty, grid_for(n_nodes), dim3(256), 0, s, node_tie, run_excl, run_head, n_nodes, run_dirty);
}
void launch_node_size(hipStream_t s, const DevParams &p, int cw, const u32 *node_start, const int16_t *blen, const u64 *run_excl,
                      const u32 *run_head, const u32 *run_dirty, u32 n_nodes, u32 *size, u32 *hflag, u32 *hcnt, u32 *blk_counts, DevCounters *ctr)
{
    if (!n_nodes) return;
    const dim3 g = grid_for(n_nodes);
    hipLaunchKernelGGL(k_node_size, g, dim3(256), 0, s, p, cw, node_start, blen, run_excl, run_head, run_dirty, n_nodes, size, hflag, hcnt, blk_counts);
    hipLaunchKernelGGL(k_node_counts, dim3(g.x < 64 ? g.x : 64), dim3(256), 0, s, blk_counts, g.x, ctr);
}
void launch_node_write(hipStream_t s, const DevParams &p, int cw, const u64 *item_key, const u32 *item_meta, const int16_t *blen,
                       const u32 *node_start, u32 node_lo, u32 node_hi, const u32 *size, const u64 *off, const u32 *hflag, const u64 *hidx,
                       const u64 *hoff, const u32 *node_tie, u8 *stream, HostNode *hnodes, u64 *h_key, u32 *h_meta, int16_t *h_blen, u32 *ord,
                       int which, const u32 *elist)
{
    if (node_hi > node_lo) hipLaunchKernelGGL(k_node_write, grid_for(node_hi - node_lo), dim3(256), 0, s, p, cw, item_key, item_meta, blen, node_start,
                                              node_hi, size, off, hflag, hidx, hoff, node_tie, stream, hnodes, h_key, h_meta, h_blen, ord, node_lo, which, elist);
}
void launch_node_emit_list(hipStream_t s, const u32 *size, u32 n_nodes, const u64 *blk_off, u32 *list)
{
    if (n_nodes) hipLaunchKernelGGL(k_node_emit_list, grid_for(n_nodes), dim3(256), 0, s, size, n_nodes, blk_off, list);
}

void launch_pack(hipStream_t s, const DevParams &p, const u8 *raw, u64 *pk, u32 *bad, DevCounters *ctr)
{
    hipLaunchKernelGGL(k_pack, grid_for(p.n_words + 4), dim3(256), 0, s, p, raw, pk, bad, ctr);
}
void launch_kmer_count(hipStream_t s, const DevParams &p, const u32 *bad, u32 *block_counts, u32 n_blocks)
{
    hipLaunchKernelGGL(k_kmer_count, grid_for((u64)n_blocks * 8), dim3(256), 0, s, p, bad, block_counts, (u32 *)nullptr, n_blocks);
}
void launch_kmer_emit(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *bad, const u64 *block_off, u32 n_blocks,
                      u64 *key_hi, u64 *key_lo, u32 *gpos)
{
    hipLaunchKernelGGL(k_kmer_emit, dim3(n_blocks), dim3(256), 0, s, p, pk, bad, block_off, key_hi, key_lo, gpos);
}
void launch_contig_check(hipStream_t s, const DevParams &p, const u32 *gpos_by_position, u64 n, DevCounters *ctr)
{
    hipLaunchKernelGGL(k_contig_check, grid_for(p.n_contigs), dim3(256), 0, s, p, gpos_by_position, n, ctr);
}
void launch_contig_scan(hipStream_t s, const DevParams &p, const u32 *bad, DevCounters *ctr)
{
    if (p.n_contigs) hipLaunchKernelGGL(k_contig_scan, dim3(p.n_contigs), dim3(256), 0, s, p, bad, ctr);
}
void launch_same_detect(hipStream_t s, const DevParams &p, const u64 *pk, const u64 *rec, const u32 *gpos, u64 n, int hb,
                        u64 *head_mask, u64 *mixed_mask)
{
    if (n) hipLaunchKernelGGL(k_same_detect, grid_for((n + 3) / 4), dim3(256), 0, s, p, pk, rec, gpos, n, hb, head_mask, mixed_mask);      // 1024 records per block
}
void launch_mixed_fix(hipStream_t s, const DevParams &p, const u64 *pk, u64 *rec, u32 *gpos, u64 n, int hb, u64 *head_mask, const u32 *mixed_list,
                      u32 n_mixed, u32 *claim, u64 *long_runs, u32 long_cap, DevCounters *ctr)
{
    if (n_mixed) hipLaunchKernelGGL(k_mixed_fix, dim3((n_mixed + 63) / 64), dim3(64), 0, s, p, pk, rec, gpos, n, hb, mixed_list, n_mixed, head_mask,
                                    claim, long_runs, long_cap, ctr);
}
void launch_mixed_fix_long(hipStream_t s, const DevParams &p, const u64 *pk, u64 *rec, u32 *gpos, u64 *rec_copy, u32 *gpos_copy,
                           const u64 *long_runs, u32 n_long, u64 *head_mask, DevCounters *ctr)
{
    if (n_long) hipLaunchKernelGGL(k_mixed_fix_long, dim3(n_long), dim3(256), 0, s, p, pk, rec, gpos, rec_copy, gpos_copy, long_runs, head_mask, ctr);
}
void launch_heads_from_keys(hipStream_t s, const DevParams &p, const u64 *pk, const u64 *hi, const u64 *lo, const u32 *gpos, u64 n,
                            u64 *head_mask, u64 *rec)
{
    if (n) hipLaunchKernelGGL(k_heads_from_keys, grid_for(n), dim3(256), 0, s, p, pk, hi, lo, gpos, n, head_mask, rec);
}
void launch_class_scan(hipStream_t s, const u64 *rec, const u32 *gpos, const u64 *head_mask, u64 n, u32 *blk_nons,
                       unsigned long long *blk_min, u64 *nons_mask, u32 *nonrep, DevCounters *ctr)
{
    if (!n) return;
    const dim3 g = grid_for(n);
    hipLaunchKernelGGL(k_class_scan, grid_for((n + 3) / 4), dim3(256), 0, s, rec, gpos, head_mask, n, blk_nons, blk_min, nons_mask, nonrep);     // 1024 records per block
    hipLaunchKernelGGL(k_min_u64, dim3(g.x < 64 ? g.x : 64), dim3(256), 0, s, blk_min, (u64)g.x, &ctr->first_singleton_key);
}
void launch_class_emit(hipStream_t s, const u32 *gpos, const u64 *head_mask, u64 n, const u64 *nons_mask, u64 *cls, u32 *cls_m,
                       const u64 *nons_off)
{
    if (n) hipLaunchKernelGGL(k_class_emit, grid_for((n + 63) / 64), dim3(256), 0, s, gpos, head_mask, n, nons_mask, cls, cls_m, nons_off);
}
void launch_rep_mask_words(hipStream_t s, const u32 *valid, const u32 *nonrep, u32 *rep, u64 n_words)
{
    if (n_words) hipLaunchKernelGGL(k_rep_mask_words, grid_for(n_words), dim3(256), 0, s, valid, nonrep, rep, n_words);
}
void launch_rep_mask(hipStream_t s, const DevParams &p, const u32 *bad, const u32 *nonrep, u32 *rep)
{
    hipLaunchKernelGGL(k_rep_mask, grid_for(p.n_words), dim3(256), 0, s, p, bad, nonrep, rep);
}
void launch_mask_count(hipStream_t s, const u64 *mask, u64 n_words, u32 *blk_count)
{
    if (n_words) hipLaunchKernelGGL(k_mask_count, grid_for(n_words), dim3(256), 0, s, mask, n_words, blk_count);
}
void launch_mask_emit(hipStream_t s, const u64 *mask, u64 n_words, const u64 *blk_off, u32 *pos)
{
    if (n_words) hipLaunchKernelGGL(k_mask_emit, grid_for(n_words), dim3(256), 0, s, mask, n_words, blk_off, pos);
}
void launch_bitmap_count(hipStream_t s, const u32 *bits, u64 n_words, u32 *block_counts, u32 n_blocks)
{
    hipLaunchKernelGGL(k_bitmap_count, dim3(n_blocks), dim3(256), 0, s, bits, n_words, block_counts);
}
void launch_bitmap_emit(hipStream_t s, const u32 *bits, u64 n_words, const u64 *block_off, u32 n_blocks, u32 *out, u32 base)
{
    hipLaunchKernelGGL(k_bitmap_emit, dim3(n_blocks), dim3(256), 0, s, bits, n_words, block_off, out, base);
}
void launch_class_info(hipStream_t s, const u64 *rec, const u32 *cls_m, const u64 *cls, u32 n_cls, const DevCounters *ctr,
                       u32 *cls_flags, u32 *cls_entries)
{
    if (n_cls) hipLaunchKernelGGL(k_class_info, grid_for(n_cls), dim3(256), 0, s, rec, cls_m, cls, n_cls, ctr, cls_flags, cls_entries);
}
void launch_class_entries(hipStream_t s, const u64 *rec, const u32 *gpos, const u32 *cls_m, const u64 *cls, u32 n_cls,
                          const u32 *cls_flags, const u64 *ent_off, u32 *ent, u8 *ent_dir, DevCounters *ctr)
{
    if (!n_cls) return;
    (void)hipMemsetAsync(&ctr->huge_entries, 0, sizeof(DevCounters::Huge), s);
    hipLaunchKernelGGL(k_class_entries, grid_for(n_cls), dim3(256), 0, s, rec, gpos, cls_m, cls, n_cls, cls_flags, ent_off, ent, ent_dir, ctr);
    hipLaunchKernelGGL(k_class_entries_huge, dim3(HUGE_GRID), dim3(256), 0, s, rec, gpos, cls_m, cls, cls_flags, ent_off, ent, ent_dir, ctr);
}
void launch_class_merge(hipStream_t s, const u64 *cls, u32 n_cls, const u32 *cls_entries, const u64 *ent_off, const u32 *ent,
                        const u8 *ent_dir, u8 *merge_prev)
{
    if (n_cls) hipLaunchKernelGGL(k_class_merge, grid_for(n_cls), dim3(256), 0, s, cls, n_cls, cls_entries, ent_off, ent, ent_dir, merge_prev);
}
void launch_seed_codes(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *reps, u64 n_reps, int seed, u64 *out)
{
    if (n_reps) hipLaunchKernelGGL(k_seed_codes, grid_for(n_reps), dim3(256), 0, s, p, pk, reps, n_reps, seed, out);
}
void launch_seed_codes_all(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *reps, u64 n_reps, u64 *const out[4], u32 *const hist[4],
                           const SeedSketches &sk, u32 tile_records, int only, int hist_shift)
{
    if (!n_reps) return;
    const dim3 g((unsigned)((n_reps + tile_records - 1) / tile_records));
    const int per_thread = (int)(tile_records / 256);
    switch (p.nseeds) {
    case 2: hipLaunchKernelGGL(k_seed_codes_all<2>, g, dim3(256), 0, s, p, pk, reps, n_reps, out[0], out[1], out[2], out[3], hist[0], hist[1], hist[2], hist[3], sk, per_thread, only, hist_shift); break;
    case 3: hipLaunchKernelGGL(k_seed_codes_all<3>, g, dim3(256), 0, s, p, pk, reps, n_reps, out[0], out[1], out[2], out[3], hist[0], hist[1], hist[2], hist[3], sk, per_thread, only, hist_shift); break;
    default: hipLaunchKernelGGL(k_seed_codes_all<4>, g, dim3(256), 0, s, p, pk, reps, n_reps, out[0], out[1], out[2], out[3], hist[0], hist[1], hist[2], hist[3], sk, per_thread, only, hist_shift); break;
    }
}
u64 verify_blocks(u64 n) { return (n + VT_TILE - 1) / VT_TILE; }

// The candidate passes that have something to do in FEW tiles get those tiles as a list instead of a grid over all tiles
// (1.3e6 workgroups that look at a flag and leave cost 1.1-1.7 ms each, three such launches per seed).  which = 0: the tiles
// the balanced kernel left to the walk; 1 / 2: the balanced kernel's / the walk's tiles whose slots the write pass has to
// produce (some found, not all kept by the counting pass).  Order within the list does not matter: every tile writes to
// places of its own.  *count must be zero.
__global__ __launch_bounds__(256) void k_tile_list(u32 n_tiles, const u8 *__restrict__ blk_mode, const u32 *__restrict__ blk_slots,
                                                   const u8 *__restrict__ blk_staged, int which, u32 *__restrict__ list, u32 *__restrict__ count)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    bool want = false;
    if (t < n_tiles) {
        const bool walk = blk_mode[t] == 0;
        if (which == 0) want = walk;
        else want = (walk == (which == 2)) && blk_slots[t] != 0 && !blk_staged[t];
    }
    const u64 bal = __ballot(want);
    if (!bal) return;
    const int lane = threadIdx.x & 63;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(count, (u32)__popcll(bal));
    base = __shfl(base, 0, 64);
    if (want) list[base + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = t;
}
void launch_tile_list(hipStream_t s, u64 n_tiles, const u8 *blk_mode, const u32 *blk_slots, const u8 *blk_staged, int which, u32 *list, u32 *count)
{
    if (n_tiles) hipLaunchKernelGGL(k_tile_list, grid_for(n_tiles), dim3(256), 0, s, (u32)n_tiles, blk_mode, blk_slots, blk_staged, which, list, count);
}

// pairs: the balanced kernel (k_candidates_pairs) or the walk (k_candidates); tiles / n_listed: the tiles to do (null: all)
template <bool WRITE>
static void launch_candidates(hipStream_t s, bool pairs, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                              const u64 *blk_off, PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode,
                              const u32 *tiles, u32 n_listed)
{
    if (!n || (tiles && !n_listed)) return;
    const dim3 g(tiles ? n_listed : (unsigned)verify_blocks(n)), b(256);
#define CAND_ARGS p, codes, n, cnt_slots, blk_slots, blk_cand, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles
    if (pairs) {
        if (sk_b && sk_c) hipLaunchKernelGGL((k_candidates_pairs<WRITE, true, true>), g, b, 0, s, CAND_ARGS);
        else if (sk_b) hipLaunchKernelGGL((k_candidates_pairs<WRITE, true, false>), g, b, 0, s, CAND_ARGS);
        else if (sk_c) hipLaunchKernelGGL((k_candidates_pairs<WRITE, false, true>), g, b, 0, s, CAND_ARGS);
        else hipLaunchKernelGGL((k_candidates_pairs<WRITE, false, false>), g, b, 0, s, CAND_ARGS);
    } else {
        if (sk_b && sk_c) hipLaunchKernelGGL((k_candidates<WRITE, true, true>), g, b, 0, s, CAND_ARGS);
        else if (sk_b) hipLaunchKernelGGL((k_candidates<WRITE, true, false>), g, b, 0, s, CAND_ARGS);
        else if (sk_c) hipLaunchKernelGGL((k_candidates<WRITE, false, true>), g, b, 0, s, CAND_ARGS);
        else hipLaunchKernelGGL((k_candidates<WRITE, false, false>), g, b, 0, s, CAND_ARGS);
    }
#undef CAND_ARGS
}
u64 verify_staging_elems(u64 n) { return verify_blocks(n) * VT_QUOTA; }
// counting pass, all tiles: the balanced kernel (blk_mode != null; it says which tiles it left to the walk) or the walk
void launch_candidates_count(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                             const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode)
{
    launch_candidates<false>(s, blk_mode != nullptr, p, codes, n, cnt_slots, blk_slots, blk_cand, nullptr, nullptr, sk_b, sk_c, staged, blk_staged, blk_mode,
                             nullptr, 0);
}
// counting pass of the walk over the listed tiles (those the balanced kernel left)
void launch_candidates_walk_count(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                  const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode, const u32 *tiles, u32 n_listed)
{
    launch_candidates<false>(s, false, p, codes, n, cnt_slots, blk_slots, blk_cand, nullptr, nullptr, sk_b, sk_c, staged, blk_staged, blk_mode, tiles, n_listed);
}
// write pass: the kept slots of every tile, then the listed tiles of either kernel (blk_mode == null: the walk over all tiles)
void launch_candidates_write(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, const u64 *blk_off,
                             PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode,
                             const u32 *tiles_pairs, u32 n_pairs, const u32 *tiles_walk, u32 n_walk)
{
    if (!n) return;
    hipLaunchKernelGGL(k_staged_slots, dim3((unsigned)verify_blocks(n)), dim3(VT_QUOTA), 0, s, staged, blk_slots, blk_staged, blk_off, slots);
    if (!blk_mode) {
        launch_candidates<true>(s, false, p, codes, n, cnt_slots, blk_slots, nullptr, blk_off, slots, sk_b, sk_c, staged, blk_staged, nullptr, nullptr, 0);
        return;
    }
    launch_candidates<true>(s, true, p, codes, n, cnt_slots, blk_slots, nullptr, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles_pairs, n_pairs);
    launch_candidates<true>(s, false, p, codes, n, cnt_slots, blk_slots, nullptr, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles_walk, n_walk);
}
// ---- PS mode (records sorted on the top 24 code bits; k_candidates_pairs_ps / k_candidates_ps): per-tile arrays hold 2 x
// verify_blocks_ps(n) entries, the balanced kernel's at [tile], the walk's at [tiles + tile] ----
u64 verify_blocks_ps(u64 n) { return (n + PS_TILE - 1) / PS_TILE; }
u64 verify_staging_elems_ps(u64 n) { return 2 * verify_blocks_ps(n) * VT_QUOTA; }

// which = 0: the tiles the walk's counting pass has to visit (left whole, or holding palindromic codes); 1 / 2: the tiles whose
// slots the write pass of the balanced kernel / of the walk has to produce
__global__ __launch_bounds__(256) void k_tile_list_ps(u32 n_tiles, const u8 *__restrict__ blk_mode, const u32 *__restrict__ blk_slots,
                                                      const u8 *__restrict__ blk_staged, int which, u32 *__restrict__ list, u32 *__restrict__ count)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    bool want = false;
    if (t < n_tiles) {
        const u32 mode = blk_mode[t];
        if (which == 0) want = mode != 1;
        else if (which == 1) want = (mode & 1) && blk_slots[t] != 0 && !blk_staged[t];
        else want = mode != 1 && blk_slots[n_tiles + t] != 0 && !blk_staged[n_tiles + t];
    }
    const u64 bal = __ballot(want);
    if (!bal) return;
    const int lane = threadIdx.x & 63;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(count, (u32)__popcll(bal));
    base = __shfl(base, 0, 64);
    if (want) list[base + (u32)__popcll(bal & ((1ull << lane) - 1ull))] = t;
}
void launch_tile_list_ps(hipStream_t s, u64 n_tiles, const u8 *blk_mode, const u32 *blk_slots, const u8 *blk_staged, int which, u32 *list, u32 *count)
{
    if (n_tiles) hipLaunchKernelGGL(k_tile_list_ps, grid_for(n_tiles), dim3(256), 0, s, (u32)n_tiles, blk_mode, blk_slots, blk_staged, which, list, count);
}

template <bool WRITE>
static void launch_pairs_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand, const u64 *blk_off,
                            PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode, const u32 *tiles, u32 n_listed)
{
    if (!n || (tiles && !n_listed)) return;
    const dim3 g(tiles ? n_listed : (unsigned)verify_blocks_ps(n)), b(256);
#define CAND_ARGS p, codes, n, cnt_slots, blk_slots, blk_cand, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles
    if (sk_b && sk_c) hipLaunchKernelGGL((k_candidates_pairs_ps<WRITE, true, true>), g, b, 0, s, CAND_ARGS);
    else if (sk_b) hipLaunchKernelGGL((k_candidates_pairs_ps<WRITE, true, false>), g, b, 0, s, CAND_ARGS);
    else if (sk_c) hipLaunchKernelGGL((k_candidates_pairs_ps<WRITE, false, true>), g, b, 0, s, CAND_ARGS);
    else hipLaunchKernelGGL((k_candidates_pairs_ps<WRITE, false, false>), g, b, 0, s, CAND_ARGS);
#undef CAND_ARGS
}
template <bool WRITE>
static void launch_walk_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand, const u64 *blk_off,
                           PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, const u8 *blk_mode, const u32 *tiles, u32 n_listed)
{
    if (!n || !n_listed) return;
    const dim3 g(n_listed), b(256);
    const u32 nt = (u32)verify_blocks_ps(n);
#define CAND_ARGS p, codes, n, nt, cnt_slots, blk_slots, blk_cand, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles
    if (sk_b && sk_c) hipLaunchKernelGGL((k_candidates_ps<WRITE, true, true>), g, b, 0, s, CAND_ARGS);
    else if (sk_b) hipLaunchKernelGGL((k_candidates_ps<WRITE, true, false>), g, b, 0, s, CAND_ARGS);
    else if (sk_c) hipLaunchKernelGGL((k_candidates_ps<WRITE, false, true>), g, b, 0, s, CAND_ARGS);
    else hipLaunchKernelGGL((k_candidates_ps<WRITE, false, false>), g, b, 0, s, CAND_ARGS);
#undef CAND_ARGS
}
void launch_candidates_count_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode)
{
    launch_pairs_ps<false>(s, p, codes, n, cnt_slots, blk_slots, blk_cand, nullptr, nullptr, sk_b, sk_c, staged, blk_staged, blk_mode, nullptr, 0);
}
void launch_candidates_walk_count_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, u64 *blk_cand,
                                     const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, const u8 *blk_mode, const u32 *tiles, u32 n_listed)
{
    launch_walk_ps<false>(s, p, codes, n, cnt_slots, blk_slots, blk_cand, nullptr, nullptr, sk_b, sk_c, staged, blk_staged, blk_mode, tiles, n_listed);
}
void launch_candidates_write_ps(hipStream_t s, const DevParams &p, const u64 *codes, u64 n, u32 *cnt_slots, u32 *blk_slots, const u64 *blk_off,
                                PairRec *slots, const u64 *sk_b, const u32 *sk_c, u64 *staged, u8 *blk_staged, u8 *blk_mode,
                                const u32 *tiles_pairs, u32 n_pairs, const u32 *tiles_walk, u32 n_walk)
{
    if (!n) return;
    hipLaunchKernelGGL(k_staged_slots, dim3((unsigned)(2 * verify_blocks_ps(n))), dim3(VT_QUOTA), 0, s, staged, blk_slots, blk_staged, blk_off, slots);
    launch_pairs_ps<true>(s, p, codes, n, cnt_slots, blk_slots, nullptr, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles_pairs, n_pairs);
    launch_walk_ps<true>(s, p, codes, n, cnt_slots, blk_slots, nullptr, blk_off, slots, sk_b, sk_c, staged, blk_staged, blk_mode, tiles_walk, n_walk);
}
void launch_judge(hipStream_t s, const DevParams &p, const u64 *pk, PairRec *slots, u64 n_slots, int seed, u32 *blk_valid)
{
    if (n_slots) hipLaunchKernelGGL(k_judge, grid_for(n_slots), dim3(256), 0, s, p, pk, slots, n_slots, seed, blk_valid);
}
void launch_slot_compact(hipStream_t s, const PairRec *slots, u64 n_slots, const u64 *blk_off, PairRec *pairs)
{
    if (n_slots) hipLaunchKernelGGL(k_slot_compact, grid_for(n_slots), dim3(256), 0, s, slots, n_slots, blk_off, pairs);
}
void launch_sum_u64(hipStream_t s, const u64 *in, u64 n, u64 *out)
{
    if (!n) return;
    const u64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_sum_u64, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s, in, n, (unsigned long long *)out);
}
void launch_items_expand(hipStream_t s, const DevParams &p, const PairRec *pairs, u64 n_pairs, u64 *item_key, u32 *item_meta)
{
    if (n_pairs) hipLaunchKernelGGL(k_items_expand, grid_for(n_pairs), dim3(256), 0, s, p, pairs, n_pairs, item_key, item_meta);
}
void launch_node_flags(hipStream_t s, const u64 *item_key, u64 m, u64 *head_mask, u32 *word_heads, DevCounters *ctr)
{
    if (m) hipLaunchKernelGGL(k_node_flags, grid_for((m + 3) / 4), dim3(256), 0, s, item_key, m, head_mask, word_heads, ctr);      // 1024 items per block
}
void launch_node_starts(hipStream_t s, const u64 *head_mask, const u64 *word_first, u64 m, u32 *node_start)
{
    if (m) hipLaunchKernelGGL(k_node_starts, grid_for((m + 63) / 64), dim3(256), 0, s, head_mask, word_first, (m + 63) / 64, node_start);
}
void launch_chain_link(hipStream_t s, const DevParams &p, const u64 *item_key, const u32 *item_meta, const u64 *head_mask,
                       const u64 *word_first, const u32 *node_start, u32 n_nodes, u64 m, u32 *succ, int16_t *blen)
{
    if (!m) return;
    (void)hipMemsetAsync(blen, 0, m * sizeof(int16_t), s);
    hipLaunchKernelGGL(k_chain_link, grid_for(m), dim3(256), 0, s, p, item_key, item_meta, head_mask, word_first, node_start, n_nodes, m, succ, blen);
}
void launch_alive_mask(hipStream_t s, const int16_t *blen, u64 m, u64 *mask)
{
    if (m) hipLaunchKernelGGL(k_alive_mask, grid_for((m + 3) / 4), dim3(256), 0, s, blen, m, mask);
}
void launch_chain_walk(hipStream_t s, const u32 *succ, const u32 *alive, u64 n_alive, int16_t *blen)
{
    if (n_alive) hipLaunchKernelGGL(k_chain_walk, grid_for(n_alive), dim3(256), 0, s, succ, alive, n_alive, blen);
}

}  // namespace cora

// ------------------------------------------------------------------ helpers of the staged / sharded flow
namespace cora {

// full canonical keys of a list of window positions (full-key fallback of the exact stage)
__global__ __launch_bounds__(256) void k_keys_from_gpos(DevParams p, const u64 *__restrict__ pk, const u32 *__restrict__ gpos, u64 n,
                                                        u64 *__restrict__ hi, u64 *__restrict__ lo)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 h, l;
    canonical_at(p, pk, gpos[i], h, l);
    hi[i] = h; lo[i] = l;
}

// key[k] = (cls[k] & rep bits) | k : lets a sort by representative carry the index into the unsorted parts
__global__ __launch_bounds__(256) void k_rekey(const u64 *__restrict__ cls, u32 n, u64 *__restrict__ key)
{
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) key[k] = (cls[k] & 0xFFFFFFFF00000000ull) | k;
}

__global__ __launch_bounds__(256) void k_gather_parts(const u64 *__restrict__ key_sorted, u32 n, const u32 *__restrict__ ce,
                                                      const u64 *__restrict__ eoff, u32 *__restrict__ ce_s, u64 *__restrict__ eoff_s)
{
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const u32 idx = (u32)key_sorted[k];
    ce_s[k] = ce[idx]; eoff_s[k] = eoff[idx];
}

// records per destination rank (the items' partition; k-mer and seed records are partitioned by radix_partition, whose
// digit is the owner: range of the top 32 bits for the uniform hashes, a multiplicative hash of them for the canonical seed
// codes, which are skewed towards small values)
__global__ __launch_bounds__(256) void k_count_dest(const u64 *__restrict__ dest, u64 n, u32 world, unsigned long long *__restrict__ counts)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) atomicAdd(&h[(u32)dest[i] & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < world && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// Multi-GPU item routing.  Lists are finished by the rank that owns their source position; the owner intervals are
// cut where a position holds no list, so that no run of consecutive positions (the unit of block merging) is split.
// wc[b * win + j] = number of items whose source is position bnd[b] - 1 + j   (bnd[b] = nominal start of rank b + 1)
__global__ __launch_bounds__(256) void k_window_counts(const u64 *__restrict__ item_key, u64 m, const u32 *__restrict__ bnd, u32 n_bnd,
                                                       u32 win, u32 *__restrict__ wc)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const u64 src = item_key[i] >> 32;
    for (u32 b = 0; b < n_bnd; b++) {
        const u64 lo = (u64)bnd[b] - 1;                                   // bnd[b] >= 1 (checked by the host)
        if (src >= lo && src < lo + win) atomicAdd(&wc[(u64)b * win + (src - lo)], 1u);
    }
}

// dest = number of cuts <= source position (cuts ascending)
__global__ __launch_bounds__(256) void k_dest_by_cuts(const u64 *__restrict__ item_key, u64 m, const u32 *__restrict__ cuts, u32 n_cuts,
                                                      u64 *__restrict__ dest)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const u32 src = (u32)(item_key[i] >> 32);
    u32 lo = 0, hi = n_cuts;                                              // first cut > src
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (cuts[mid] <= src) lo = mid + 1; else hi = mid; }
    dest[i] = lo;
}

void launch_window_counts(hipStream_t s, const u64 *item_key, u64 m, const u32 *bnd, u32 n_bnd, u32 win, u32 *wc)
{
    if (m && n_bnd) hipLaunchKernelGGL(k_window_counts, grid_for(m), dim3(256), 0, s, item_key, m, bnd, n_bnd, win, wc);
}
void launch_dest_by_cuts(hipStream_t s, const u64 *item_key, u64 m, const u32 *cuts, u32 n_cuts, u64 *dest)
{
    if (m) hipLaunchKernelGGL(k_dest_by_cuts, grid_for(m), dim3(256), 0, s, item_key, m, cuts, n_cuts, dest);
}
void launch_keys_from_gpos(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *gpos, u64 n, u64 *hi, u64 *lo)
{
    if (n) hipLaunchKernelGGL(k_keys_from_gpos, grid_for(n), dim3(256), 0, s, p, pk, gpos, n, hi, lo);
}
void launch_rekey(hipStream_t s, const u64 *cls, u32 n, u64 *key)
{
    if (n) hipLaunchKernelGGL(k_rekey, grid_for(n), dim3(256), 0, s, cls, n, key);
}
void launch_gather_parts(hipStream_t s, const u64 *key_sorted, u32 n, const u32 *ce, const u64 *eoff, u32 *ce_s, u64 *eoff_s)
{
    if (n) hipLaunchKernelGGL(k_gather_parts, grid_for(n), dim3(256), 0, s, key_sorted, n, ce, eoff, ce_s, eoff_s);
}
void launch_count_dest(hipStream_t s, const u64 *dest, u64 n, u32 world, unsigned long long *counts)
{
    if (!n) return;
    const u64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_count_dest, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, s, dest, n, world, counts);
}

}  // namespace cora

namespace cora {
void launch_kmer_count_range(hipStream_t s, const DevParams &p, const u32 *bad, u32 *block_counts, u32 block_lo, u32 n_blocks, u32 *valid)
{
    if (!n_blocks) return;
    DevParams q = p;
    q.pos_base = (u64)block_lo * 256;
    hipLaunchKernelGGL(k_kmer_count, grid_for((u64)n_blocks * 8), dim3(256), 0, s, q, bad, block_counts, valid, n_blocks);
}
void launch_test_window_bits(hipStream_t s, const DevParams &p, const u32 *bad, u32 *fast, u32 *slow)
{
    hipLaunchKernelGGL(k_test_window_bits, grid_for(p.n_words), dim3(256), 0, s, p, bad, fast, slow);
}
void launch_kmer_emit_hashed_range(hipStream_t s, const DevParams &p, const u64 *pk, const u32 *valid, const u64 *block_off, u32 block_lo,
                                   u32 n_blocks, u64 *rec, u32 *gpos, unsigned char *first_digit, int fd_shift)
{
    if (!n_blocks) return;
    DevParams q = p;
    q.pos_base = (u64)block_lo * 256;
    hipLaunchKernelGGL(k_kmer_emit_hashed, dim3(n_blocks), dim3(256), 0, s, q, pk, valid, block_off, rec, gpos, first_digit, fd_shift);
}
}  // namespace cora

// ---- element-wise combines of the in-process multi-GPU run (multi.hip): a peer's array folded into this rank's
namespace cora {
__global__ __launch_bounds__(256) void k_or_u32(u32 *__restrict__ acc, const u32 *__restrict__ other, u64 n)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] |= other[i];
}
__global__ __launch_bounds__(256) void k_add_u32(u32 *__restrict__ acc, const u32 *__restrict__ other, u64 n)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] += other[i];
}
void launch_or_u32(hipStream_t s, u32 *acc, const u32 *other, u64 n) { if (n) hipLaunchKernelGGL(k_or_u32, grid_for(n), dim3(256), 0, s, acc, other, n); }
void launch_add_u32(hipStream_t s, u32 *acc, const u32 *other, u64 n) { if (n) hipLaunchKernelGGL(k_add_u32, grid_for(n), dim3(256), 0, s, acc, other, n); }
}  // namespace cora
