// tables.hip -- the consumer side of the two homology tables: <Exact>/<Inexact> bytes -> HBM-resident CSR lists,
// reverse lookups and dense position hubs (include/cora_hip.h, "the consumer side of the tables").
// MI = /root/reference/mappingInference.cpp; restated on the CPU in oracle/loader_oracle.c.
//
// Parsing.  Both files are sequences of variable-length records with no index: a record's length is known only after
// walking its items up to the 0xFF that stands where the next item's dir byte would (MI:590-605, 686-703), and 0xFF also
// occurs inside positions.  The only bytes that decide anything are the "decision bytes" (dir or 0xFF), and the distance
// from one to the next is `item` bytes after an item and 1 + tail + hdr bytes after a terminator.  So the parser state at
// a chunk boundary is just "how many bytes until the next decision byte", < 16 values.  Each thread takes a 64-byte
// chunk and computes, for every possible incoming state, the outgoing one (a 16-nibble map in one u64); maps compose
// associatively, so a scan over them gives every chunk its true incoming state; then the chunks are walked for real:
// count, scan, emit.  Four small kernels per table, no sequential pass over the bytes.
//
// Reverse lookups (MI:638-650).  maxRevLookup[L] = L - i, i being the first node from a moving cursor whose longest
// block reaches L.  The cursor makes this sequential, but it restarts at every node no earlier node reaches (block
// lengths are < 128, so that is a look-back over < 127 positions): one thread walks each such run of nodes.
//
// Hubs (MI:1093-1152).  equivHub[p] ends as the LAST class that covers p and classes are visited in ascending order, so
// it is an atomic max of the class index.  homHub[member + j] = homHub[representative + j] copies the value held AT
// THAT MOMENT; in every table a builder writes, no position is both copied to and copied from and all copies onto one
// position carry one value, so order does not matter and the copies run in parallel.  Both conditions are checked on
// the device; if one fails (hand-made tables in tests/golden/loader_cases) homHub is refilled by ONE thread in the
// reference's order -- still on the GPU, slow, and exact.
#include "job_internal.hpp"

#include <hip/hip_runtime.h>

typedef uint8_t u8;
typedef uint16_t u16;

namespace {

constexpr u32 CHUNK = 64;          // bytes per parser thread
constexpr u32 PT = 256;            // parser threads per workgroup

struct Fmt { u32 hdr, item, tail, cb, d; };   // header bytes, item bytes (dir included), bytes between 0xFF and the next header, contig id bytes

struct LoadCounters {
    u64 recs[2], items[2];         // totals of the two parses
    u32 end_state[2];              // bytes from end of file to the next decision byte: must equal hdr
    u32 err_order, err_empty, err_range, hazard;
    u64 first_bad;                 // index of a node / record an error was seen at (any one)
};

__device__ __forceinline__ u32 nib(u64 m, u32 x) { return (u32)(m >> (4 * x)) & 15u; }
// state map of "a, then b"
__device__ __forceinline__ u64 then(u64 a, u64 b)
{
    u64 r = 0;
#pragma unroll
    for (u32 x = 0; x < 16; x++) r |= (u64)nib(b, nib(a, x)) << (4 * x);
    return r;
}
constexpr u64 IDENT = 0xFEDCBA9876543210ull;

__device__ __forceinline__ u32 jump_after(const u8 *b, u64 p, const Fmt f) { return b[p] == 0xFF ? 1 + f.tail + f.hdr : f.item; }

// pass 1: per-chunk state maps, scanned inside the workgroup.  pre[t] = map of the chunks before t in its workgroup.
__global__ __launch_bounds__(PT) void k_parse_maps(const u8 *__restrict__ b, u64 n, Fmt f, u64 *__restrict__ pre, u64 *__restrict__ group_map)
{
    __shared__ u64 sm[PT];
    const u64 t = (u64)blockIdx.x * PT + threadIdx.x;
    const u64 lo = t * CHUNK;
    u64 m = IDENT;
    if (lo < n) {
        const u32 len = n - lo < CHUNK ? (u32)(n - lo) : CHUNK;
        const u32 S = f.item > 1 + f.tail + f.hdr ? f.item : 1 + f.tail + f.hdr;
        m = 0;
        for (u32 s = 0; s < 16; s++) {
            u32 p = s;
            if (s < S) { while (p < len) p += jump_after(b, lo + p, f); p -= len; }
            m |= (u64)(p & 15u) << (4 * s);
        }
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (u32 off = 1; off < PT; off <<= 1) {             // inclusive scan, earlier chunks first
        u64 mine = sm[threadIdx.x], before = threadIdx.x >= off ? sm[threadIdx.x - off] : IDENT;
        __syncthreads();
        sm[threadIdx.x] = then(before, mine);
        __syncthreads();
    }
    pre[t] = threadIdx.x ? sm[threadIdx.x - 1] : IDENT;
    if (threadIdx.x == PT - 1) group_map[blockIdx.x] = sm[PT - 1];
}

// pass 2: exclusive scan of the workgroup maps, one workgroup, carried map
__global__ __launch_bounds__(PT) void k_parse_groupscan(u64 *group_map, u64 n_groups)
{
    __shared__ u64 sm[PT];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = IDENT;
    __syncthreads();
    for (u64 base = 0; base < n_groups; base += PT) {
        const u64 g = base + threadIdx.x;
        const u64 own = g < n_groups ? group_map[g] : IDENT;
        sm[threadIdx.x] = own;
        __syncthreads();
        for (u32 off = 1; off < PT; off <<= 1) {
            u64 mine = sm[threadIdx.x], before = threadIdx.x >= off ? sm[threadIdx.x - off] : IDENT;
            __syncthreads();
            sm[threadIdx.x] = then(before, mine);
            __syncthreads();
        }
        const u64 c = carry;
        if (g < n_groups) group_map[g] = then(c, threadIdx.x ? sm[threadIdx.x - 1] : IDENT);
        __syncthreads();
        if (threadIdx.x == PT - 1) carry = then(c, sm[PT - 1]);
        __syncthreads();
    }
}

// pass 3: every chunk's true incoming state; terminators and items per chunk
__global__ __launch_bounds__(PT) void k_parse_count(const u8 *__restrict__ b, u64 n, Fmt f, const u64 *__restrict__ pre, const u64 *__restrict__ group_map,
                                                    u8 *__restrict__ state, u32 *__restrict__ n_term, u32 *__restrict__ n_item, LoadCounters *ctr, int which)
{
    const u64 t = (u64)blockIdx.x * PT + threadIdx.x;
    const u64 lo = t * CHUNK;
    u32 terms = 0, items = 0, s = 0;
    if (lo < n) {
        const u32 len = n - lo < CHUNK ? (u32)(n - lo) : CHUNK;
        s = nib(pre[t], nib(group_map[blockIdx.x], f.hdr));    // the file starts hdr bytes before its first decision byte
        u32 p = s;
        while (p < len) { if (b[lo + p] == 0xFF) { terms++; p += 1 + f.tail + f.hdr; } else { items++; p += f.item; } }
        if (lo + len == n) ctr->end_state[which] = p - len;
    }
    state[t] = (u8)s; n_term[t] = terms; n_item[t] = items;
}

__device__ __forceinline__ u32 rd32(const u8 *p) { return p[0] | p[1] << 8 | p[2] << 16 | (u32)p[3] << 24; }
__device__ __forceinline__ u16 rdchr(const u8 *p, u32 cb) { return cb == 2 ? (u16)(p[0] | p[1] << 8) : (u16)p[0]; }

// pass 4: the walk again, now writing.  Only launched for bytes that parsed to exactly the announced record count
// ending on a record boundary, so every field read below lies inside the file.
// Exact (MI:680-721): chr pos {dir chr pos}* 0xFF blockLen.   Inexact (MI:581-633): chr pos {dir chr pos off*d blockLen}* 0xFF.
template <bool INEXACT>
__global__ __launch_bounds__(PT) void k_parse_emit(const u8 *__restrict__ b, u64 n, Fmt f, const u8 *__restrict__ state,
                                                   const u64 *__restrict__ term_at, const u64 *__restrict__ item_at, u64 n_recs,
                                                   u16 *__restrict__ r_chr, u32 *__restrict__ r_pos, u32 *__restrict__ r_block, u64 *__restrict__ r_off,
                                                   u8 *__restrict__ i_dir, u16 *__restrict__ i_chr, u32 *__restrict__ i_pos,
                                                   u8 *__restrict__ i_block, u8 *__restrict__ i_offs)
{
    const u64 t = (u64)blockIdx.x * PT + threadIdx.x;
    const u64 lo = t * CHUNK;
    if (lo >= n) return;
    const u32 len = n - lo < CHUNK ? (u32)(n - lo) : CHUNK;
    u64 r = term_at[t], k = item_at[t];
    if (t == 0) { r_chr[0] = rdchr(b, f.cb); r_pos[0] = rd32(b + f.cb); r_off[0] = 0; }
    for (u32 p = state[t]; p < len;) {
        const u8 *q = b + lo + p;
        if (q[0] == 0xFF) {
            r_off[r + 1] = k;
            if (!INEXACT) r_block[r] = rd32(q + 1);
            const u8 *h = q + 1 + f.tail;
            if (r + 1 < n_recs) { r_chr[r + 1] = rdchr(h, f.cb); r_pos[r + 1] = rd32(h + f.cb); }
            r++;
            p += 1 + f.tail + f.hdr;
        } else {
            i_chr[k] = rdchr(q + 1, f.cb);
            i_pos[k] = rd32(q + 1 + f.cb);
            if (INEXACT) {
                i_dir[k] = q[0] & 1;                                        // 1-bit field (MI:480, 607)
                for (u32 e = 0; e < f.d; e++) i_offs[k * f.d + e] = q[1 + f.cb + 4 + e];
                const u8 *bl = q + 1 + f.cb + 4 + f.d;
                i_block[k] = (u8)((bl[0] | bl[1] << 8) & 127);              // short read into a 7-bit field (MI:479, 617-618)
            } else {
                i_dir[k] = q[0];
            }
            k++;
            p += f.item;
        }
    }
}

// ---- reverse lookups ----

// reach[L] = last position node L's longest block covers (its first item is the longest, MI:643); run_head[L] = no
// earlier node of the contig reaches L.  Also the checks the walk relies on: nodes ascending, none empty.
__global__ void k_node_reach(const u16 *__restrict__ chr, const u32 *__restrict__ pos, const u64 *__restrict__ off, const u8 *__restrict__ i_block,
                             u64 n, u32 *__restrict__ reach, LoadCounters *ctr)
{
    const u64 L = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n) return;
    u32 bl = 0;
    if (off[L + 1] > off[L]) bl = i_block[off[L]];
    if (bl == 0) { atomicAdd(&ctr->err_empty, 1u); ctr->first_bad = L; }
    if (L && !(chr[L - 1] < chr[L] || (chr[L - 1] == chr[L] && pos[L - 1] < pos[L]))) { atomicAdd(&ctr->err_order, 1u); ctr->first_bad = L; }
    reach[L] = pos[L] + bl - 1;
}

__global__ void k_run_heads(const u16 *__restrict__ chr, const u32 *__restrict__ pos, const u32 *__restrict__ reach, u64 n, u8 *__restrict__ head)
{
    const u64 L = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n) return;
    bool reached = false;
    for (u64 i = L; i-- > 0;) {
        if (chr[i] != chr[L] || (u64)pos[i] + 126 < pos[L]) break;      // blocks are < 128 long
        if (reach[i] >= pos[L]) { reached = true; break; }
    }
    head[L] = !reached;
}

// one thread per run: the reference's moving cursor (MI:639-650), restricted to the run it cannot leave
__global__ void k_rev_lookup(const u32 *__restrict__ pos, const u32 *__restrict__ reach, const u8 *__restrict__ head, u64 n, u8 *__restrict__ maxrev)
{
    const u64 L = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n || !head[L]) return;
    maxrev[L] = 0;
    u64 i = L;
    for (u64 look = L + 1; look < n && !head[look]; look++) {
        while (reach[i] < pos[look]) i++;                                // ends at i == look at the latest: reach >= pos
        maxrev[look] = (u8)(look - i);
    }
}

// ---- hubs ----

struct Contigs { const u64 *start; const u32 *len; u32 n; };

__device__ __forceinline__ bool hub_index(const Contigs c, u32 chr, u32 pos, u64 &g)
{
    if (chr < 1 || chr > c.n || pos < 1 || pos > c.len[chr - 1]) return false;
    g = c.start[chr - 1] + pos - 1;
    return true;
}

__global__ void k_hub_nodes(const u16 *__restrict__ chr, const u32 *__restrict__ pos, u64 n, Contigs c, u32 *__restrict__ hom_hub, LoadCounters *ctr)
{
    const u64 L = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= n) return;
    u64 g;
    if (!hub_index(c, chr[L], pos[L], g)) { atomicAdd(&ctr->err_range, 1u); return; }
    hom_hub[g] = (u32)(L + 1);                                           // homHub[chr][pos] = i (MI:1096-1100); positions are distinct
}

constexpr u32 SPAN_LANES = 8;
struct ExactView { const u16 *chr; const u32 *pos; const u32 *block; const u64 *off; const u8 *i_dir; const u16 *i_chr; const u32 *i_pos; u64 n, m; };

// Which record an entry belongs to: rec_of[k] for k < m, filled from the offsets.
__global__ void k_entry_owner(const u64 *__restrict__ off, u64 n, u32 *__restrict__ rec_of)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (u64 k = off[i]; k < off[i + 1]; k++) rec_of[k] = (u32)i;
}

// SPAN_LANES lanes per "span" (most blocks are a few positions long; a long one is strided over by its lanes):
// spans [0, n) are the representatives' ranges, spans [n, n + m) the members'.
// phase 0: equivHub (atomic max of the 1-based record index) and the bitmap of positions copies write to.
// phase 1: is any source also a target?  then the copies.      phase 2: did every copy keep its value?
template <int PHASE>
__global__ __launch_bounds__(256) void k_hub_spans(ExactView x, const u32 *__restrict__ rec_of, Contigs c, int exact_only,
                                                   u32 *__restrict__ equiv_hub, u32 *__restrict__ hom_hub, u32 *__restrict__ target_mask, LoadCounters *ctr)
{
    const u64 span = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / SPAN_LANES;
    const u32 lane = threadIdx.x % SPAN_LANES;
    const u64 n_spans = exact_only ? x.n : x.n + x.m;
    if (span >= n_spans) return;
    if (span < x.n) {
        if (PHASE != 0) return;
        const u32 rc = x.chr[span], rp = x.pos[span], bl = x.block[span];
        for (u32 j = lane; j < bl; j += SPAN_LANES) {
            u64 g;
            if (!hub_index(c, rc, rp + j, g)) { atomicAdd(&ctr->err_range, 1u); continue; }
            atomicMax(&equiv_hub[g], (u32)(span + 1));                   // MI:1108-1111
        }
        return;
    }
    const u64 k = span - x.n;
    const u32 i = rec_of[k];
    const u32 rc = x.chr[i], rp = x.pos[i], bl = x.block[i];
    const u32 mc = x.i_chr[k], mp = x.i_pos[k];
    const bool fwd = x.i_dir[k] == 0;                                    // MI:1116
    for (u32 j = lane; j < bl; j += SPAN_LANES) {
        u64 t, s;
        if (!hub_index(c, mc, fwd ? mp + j : mp - j, t) || !hub_index(c, rc, rp + j, s)) { if (PHASE == 0) atomicAdd(&ctr->err_range, 1u); continue; }
        if (PHASE == 0) {
            atomicMax(&equiv_hub[t], i + 1);                             // MI:1120, 1130
            if (t != s) atomicOr(&target_mask[t >> 5], 1u << (t & 31));
        } else if (PHASE == 1) {
            if ((target_mask[s >> 5] >> (s & 31)) & 1) atomicOr(&ctr->hazard, 1u);
            else if (t != s) hom_hub[t] = hom_hub[s];                    // MI:1123, 1133
        } else {
            if (hom_hub[t] != hom_hub[s]) atomicOr(&ctr->hazard, 1u);
        }
    }
}

// the reference's own order, one thread (only for tables that failed the checks above)
__global__ void k_hub_replay(ExactView x, Contigs c, u32 *__restrict__ hom_hub)
{
    if (blockIdx.x || threadIdx.x) return;
    for (u64 i = 0; i < x.n; i++) {
        const u32 rc = x.chr[i], rp = x.pos[i], bl = x.block[i];
        for (u64 k = x.off[i]; k < x.off[i + 1]; k++) {
            const u32 mc = x.i_chr[k], mp = x.i_pos[k];
            const bool fwd = x.i_dir[k] == 0;
            for (u32 j = 0; j < bl; j++) {
                u64 t, s;
                if (!hub_index(c, mc, fwd ? mp + j : mp - j, t) || !hub_index(c, rc, rp + j, s)) continue;
                hom_hub[t] = hom_hub[s];
            }
        }
    }
}

// RetrieveOffsetAndDirInfo (MI:6618-6674) for a batch of positions, one thread each: the hub names the class, then the
// representative's range and the members' ranges are tried in list order, in the reference's unsigned arithmetic.
// result: class index (1-based, 0 = no class), block offset, direction from the position to the representative.
__global__ __launch_bounds__(256) void k_lookup_equiv(ExactView x, Contigs c, const u32 *__restrict__ equiv_hub,
                                                      const u32 *__restrict__ q_chr, const u32 *__restrict__ q_pos, u64 n,
                                                      u32 *__restrict__ r_idx, int16_t *__restrict__ r_off, u8 *__restrict__ r_dir, LoadCounters *ctr)
{
    const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u32 chr = q_chr[q], pos = q_pos[q];
    u32 idx = 0; int16_t off = 0; u8 dir = 0;
    u64 g;
    if (hub_index(c, chr, pos, g)) idx = equiv_hub[g];               // a key the maps cannot hold is "not found" (MI:6621-6630)
    if (idx) {
        const u32 i = idx - 1, bl = x.block[i], rp = x.pos[i];
        bool found = false;
        if (chr == x.chr[i] && pos >= rp && pos < rp + bl) { off = (int16_t)(pos - rp); found = true; }
        for (u64 k = x.off[i]; !found && k < x.off[i + 1]; k++) {
            if (chr != x.i_chr[k]) continue;
            const u32 hp = x.i_pos[k];
            if (x.i_dir[k] == 0) { if (pos >= hp && pos < hp + bl) { off = (int16_t)(pos - hp); found = true; } }
            else if (pos <= hp && pos > hp - bl) { dir = 1; off = (int16_t)(hp - pos); found = true; }
        }
        if (!found) { atomicAdd(&ctr->err_range, 1u); idx = 0xFFFFFFFFu; }   // the reference asserts here (MI:6673)
    }
    r_idx[q] = idx; r_off[q] = off; r_dir[q] = dir;
}

u32 grid_for(u64 n, u32 per) { return (u32)((n + per - 1) / per ? (n + per - 1) / per : 1); }

}  // namespace

struct cora_hip_tables {
    int d = 0, cb = 1, exact_only = 0;
    hipStream_t stream = nullptr;
    u64 n_eq = 0, m_eq = 0, n_hm = 0, m_hm = 0, G = 0;
    std::vector<u32> len;
    DBuf<u64> d_start; DBuf<u32> d_len;
    DBuf<u16> eq_chr; DBuf<u32> eq_pos, eq_block; DBuf<u64> eq_off; DBuf<u8> eqi_dir; DBuf<u16> eqi_chr; DBuf<u32> eqi_pos;
    DBuf<u16> hm_chr; DBuf<u32> hm_pos; DBuf<u64> hm_off; DBuf<u8> hm_maxrev;
    DBuf<u8> hmi_dir; DBuf<u16> hmi_chr; DBuf<u32> hmi_pos; DBuf<u8> hmi_block, hmi_offs;
    DBuf<u32> equiv_hub, hom_hub;
    cora_hip_tables_info info;
    ~cora_hip_tables() { if (stream) (void)hipStreamDestroy(stream); }
};

namespace {

struct Parsed { DBuf<u8> state; DBuf<u64> term_at, item_at; u64 recs = 0, items = 0; u32 end_state = 0; };

// passes 1-3 and the two scans; leaves what pass 4 needs in `ps`
int parse_count(cora_hip_tables *t, const u8 *bytes, u64 n, const Fmt &f, int which, LoadCounters *ctr, Parsed &ps)
{
    const u64 chunks = (n + CHUNK - 1) / CHUNK;
    const u32 groups = grid_for(chunks, PT);
    const u64 slots = (u64)groups * PT;
    DBuf<u64> pre, gmap, scan_tmp;
    DBuf<u32> n_term, n_item;
    RCQ(pre.alloc(slots)); RCQ(gmap.alloc(groups)); RCQ(n_term.alloc(slots)); RCQ(n_item.alloc(slots));
    RCQ(ps.state.alloc(slots)); RCQ(ps.term_at.alloc(slots)); RCQ(ps.item_at.alloc(slots));
    RCQ(scan_tmp.alloc(scan_scratch_elems(slots)));
    hipStream_t s = t->stream;
    hipLaunchKernelGGL(k_parse_maps, dim3(groups), dim3(PT), 0, s, bytes, n, f, pre.p, gmap.p);
    hipLaunchKernelGGL(k_parse_groupscan, dim3(1), dim3(PT), 0, s, gmap.p, (u64)groups);
    hipLaunchKernelGGL(k_parse_count, dim3(groups), dim3(PT), 0, s, bytes, n, f, pre.p, gmap.p, ps.state.p, n_term.p, n_item.p, ctr, which);
    RCQ(exclusive_scan_u32_to_u64(s, n_term.p, ps.term_at.p, slots, scan_tmp.p, &ctr->recs[which]));
    RCQ(exclusive_scan_u32_to_u64(s, n_item.p, ps.item_at.p, slots, scan_tmp.p, &ctr->items[which]));
    LoadCounters h;
    HIPQ(hipMemcpyAsync(&h, ctr, sizeof h, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    ps.recs = h.recs[which]; ps.items = h.items[which]; ps.end_state = n ? h.end_state[which] : f.hdr;
    return 0;
}

int check_parse(const Parsed &ps, const Fmt &f, u64 claimed, u64 n, const char *name)
{
    if (ps.recs != claimed || ps.end_state != f.hdr)
        return fail(91, "%s: %llu bytes parse to %llu records%s, %s.itemCount says %llu", name, (unsigned long long)n, (unsigned long long)ps.recs,
                    ps.end_state != f.hdr ? " and end inside a record" : "", name, (unsigned long long)claimed);
    return 0;
}

int load_impl(cora_hip_tables *t, const u8 *exact, size_t exact_bytes, u64 exact_count, const u8 *inexact, size_t inexact_bytes, u64 inexact_count)
{
    HIPQ(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    hipStream_t s = t->stream;
    const u32 cb = (u32)t->cb, d = (u32)t->d;
    const Fmt fx = {cb + 4, 1 + cb + 4, 4, cb, d};
    const Fmt fi = {cb + 4, 1 + cb + 4 + d + 2, 0, cb, d};
    const bool inex = !t->exact_only;
    if (exact_count >= 0xFFFFFFFFull || inexact_count >= 0xFFFFFFFFull) return fail(91, "more than 2^32 - 2 records: the reference's list indices are unsigned int");

    // contigs
    const u32 nc = (u32)t->len.size();
    std::vector<u64> start(nc + 1, 0);
    for (u32 c = 0; c < nc; c++) start[c + 1] = start[c] + t->len[c];
    t->G = start[nc];
    RCQ(t->d_start.alloc(nc + 1)); RCQ(t->d_len.alloc(nc));
    HIPQ(hipMemcpyAsync(t->d_start.p, start.data(), (nc + 1) * 8, hipMemcpyHostToDevice, s));
    if (nc) HIPQ(hipMemcpyAsync(t->d_len.p, t->len.data(), nc * 4, hipMemcpyHostToDevice, s));
    const Contigs ctg = {t->d_start.p, t->d_len.p, nc};

    // table bytes to HBM (not part of the timed stages: a caller that built the tables on this GPU has them there)
    DBuf<u8> bx, bi;
    RCQ(bx.alloc(exact_bytes + 16)); RCQ(bi.alloc((inex ? inexact_bytes : 0) + 16));
    if (exact_bytes) HIPQ(hipMemcpyAsync(bx.p, exact, exact_bytes, hipMemcpyHostToDevice, s));
    if (inex && inexact_bytes) HIPQ(hipMemcpyAsync(bi.p, inexact, inexact_bytes, hipMemcpyHostToDevice, s));
    DBuf<LoadCounters> ctr;
    RCQ(ctr.alloc(1));
    HIPQ(hipMemsetAsync(ctr.p, 0, sizeof(LoadCounters), s));

    hipEvent_t ev[4];
    for (auto &e : ev) HIPQ(hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 4; i++) (void)hipEventDestroy(e[i]); } } guard{ev};
    HIPQ(hipEventRecord(ev[0], s));

    // ---- parse ----
    {
        Parsed px, pi;
        RCQ(parse_count(t, bx.p, exact_bytes, fx, 0, ctr.p, px));
        RCQ(check_parse(px, fx, exact_count, exact_bytes, "<Exact>"));
        if (inex) {
            RCQ(parse_count(t, bi.p, inexact_bytes, fi, 1, ctr.p, pi));
            RCQ(check_parse(pi, fi, inexact_count, inexact_bytes, "<Inexact>"));
        }
        t->n_eq = px.recs; t->m_eq = px.items; t->n_hm = pi.recs; t->m_hm = pi.items;
        RCQ(t->eq_chr.alloc(t->n_eq)); RCQ(t->eq_pos.alloc(t->n_eq)); RCQ(t->eq_block.alloc(t->n_eq)); RCQ(t->eq_off.alloc(t->n_eq + 1));
        RCQ(t->eqi_dir.alloc(t->m_eq)); RCQ(t->eqi_chr.alloc(t->m_eq)); RCQ(t->eqi_pos.alloc(t->m_eq));
        RCQ(t->hm_chr.alloc(t->n_hm)); RCQ(t->hm_pos.alloc(t->n_hm)); RCQ(t->hm_off.alloc(t->n_hm + 1)); RCQ(t->hm_maxrev.alloc(t->n_hm));
        RCQ(t->hmi_dir.alloc(t->m_hm)); RCQ(t->hmi_chr.alloc(t->m_hm)); RCQ(t->hmi_pos.alloc(t->m_hm));
        RCQ(t->hmi_block.alloc(t->m_hm)); RCQ(t->hmi_offs.alloc(t->m_hm * d));
        HIPQ(hipMemsetAsync(t->eq_off.p, 0, 8, s)); HIPQ(hipMemsetAsync(t->hm_off.p, 0, 8, s));
        if (t->n_eq)
            hipLaunchKernelGGL((k_parse_emit<false>), dim3(grid_for((exact_bytes + CHUNK - 1) / CHUNK, PT)), dim3(PT), 0, s, bx.p, (u64)exact_bytes, fx,
                               px.state.p, px.term_at.p, px.item_at.p, t->n_eq, t->eq_chr.p, t->eq_pos.p, t->eq_block.p, t->eq_off.p,
                               t->eqi_dir.p, t->eqi_chr.p, t->eqi_pos.p, (u8 *)nullptr, (u8 *)nullptr);
        if (t->n_hm)
            hipLaunchKernelGGL((k_parse_emit<true>), dim3(grid_for((inexact_bytes + CHUNK - 1) / CHUNK, PT)), dim3(PT), 0, s, bi.p, (u64)inexact_bytes, fi,
                               pi.state.p, pi.term_at.p, pi.item_at.p, t->n_hm, t->hm_chr.p, t->hm_pos.p, (u32 *)nullptr, t->hm_off.p,
                               t->hmi_dir.p, t->hmi_chr.p, t->hmi_pos.p, t->hmi_block.p, t->hmi_offs.p);
        HIPQ(hipEventRecord(ev[1], s));
        HIPQ(hipStreamSynchronize(s));       // pass 4 reads px / pi, which go out of scope here
    }

    // ---- reverse lookups ----
    if (t->n_hm) {
        DBuf<u32> reach; DBuf<u8> head;
        RCQ(reach.alloc(t->n_hm)); RCQ(head.alloc(t->n_hm));
        const u32 g = grid_for(t->n_hm, 256);
        hipLaunchKernelGGL(k_node_reach, dim3(g), dim3(256), 0, s, t->hm_chr.p, t->hm_pos.p, t->hm_off.p, t->hmi_block.p, t->n_hm, reach.p, ctr.p);
        LoadCounters h;
        HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        if (h.err_order) return fail(92, "<Inexact>: %u nodes are not in ascending (contig, position) order (one of them: node %llu)", h.err_order, (unsigned long long)h.first_bad + 1);
        if (h.err_empty) return fail(92, "<Inexact>: %u nodes have no item or a first item of block length 0 (one of them: node %llu); the reference reads list[0] (mappingInference.cpp:643)", h.err_empty, (unsigned long long)h.first_bad + 1);
        hipLaunchKernelGGL(k_run_heads, dim3(g), dim3(256), 0, s, t->hm_chr.p, t->hm_pos.p, reach.p, t->n_hm, head.p);
        hipLaunchKernelGGL(k_rev_lookup, dim3(g), dim3(256), 0, s, t->hm_pos.p, reach.p, head.p, t->n_hm, t->hm_maxrev.p);
        HIPQ(hipEventRecord(ev[2], s));
        HIPQ(hipStreamSynchronize(s));
    } else {
        HIPQ(hipEventRecord(ev[2], s));
    }

    // ---- hubs ----
    RCQ(t->equiv_hub.alloc(t->G)); RCQ(t->hom_hub.alloc(t->G));
    HIPQ(hipMemsetAsync(t->equiv_hub.p, 0, (t->G ? t->G : 1) * 4, s));
    HIPQ(hipMemsetAsync(t->hom_hub.p, 0, (t->G ? t->G : 1) * 4, s));
    if (t->n_hm) hipLaunchKernelGGL(k_hub_nodes, dim3(grid_for(t->n_hm, 256)), dim3(256), 0, s, t->hm_chr.p, t->hm_pos.p, t->n_hm, ctg, t->hom_hub.p, ctr.p);
    t->info.serial_hub_replay = 0;
    if (t->n_eq) {
        const ExactView x = {t->eq_chr.p, t->eq_pos.p, t->eq_block.p, t->eq_off.p, t->eqi_dir.p, t->eqi_chr.p, t->eqi_pos.p, t->n_eq, t->m_eq};
        DBuf<u32> rec_of, tmask;
        RCQ(rec_of.alloc(t->m_eq)); RCQ(tmask.alloc(t->G / 32 + 1));
        HIPQ(hipMemsetAsync(tmask.p, 0, (t->G / 32 + 1) * 4, s));
        hipLaunchKernelGGL(k_entry_owner, dim3(grid_for(t->n_eq, 256)), dim3(256), 0, s, t->eq_off.p, t->n_eq, rec_of.p);
        const u64 spans = t->exact_only ? t->n_eq : t->n_eq + t->m_eq;
        const u32 g = grid_for(spans, 256 / SPAN_LANES);
        hipLaunchKernelGGL((k_hub_spans<0>), dim3(g), dim3(256), 0, s, x, rec_of.p, ctg, t->exact_only, t->equiv_hub.p, t->hom_hub.p, tmask.p, ctr.p);
        if (!t->exact_only && t->m_eq) {
            hipLaunchKernelGGL((k_hub_spans<1>), dim3(g), dim3(256), 0, s, x, rec_of.p, ctg, 0, t->equiv_hub.p, t->hom_hub.p, tmask.p, ctr.p);
            hipLaunchKernelGGL((k_hub_spans<2>), dim3(g), dim3(256), 0, s, x, rec_of.p, ctg, 0, t->equiv_hub.p, t->hom_hub.p, tmask.p, ctr.p);
        }
        LoadCounters h;
        HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        if (h.err_range) return fail(93, "the tables name %u positions outside the %u contigs given", h.err_range, nc);
        if (h.hazard) {
            t->info.serial_hub_replay = 1;
            HIPQ(hipMemsetAsync(t->hom_hub.p, 0, (t->G ? t->G : 1) * 4, s));
            if (t->n_hm) hipLaunchKernelGGL(k_hub_nodes, dim3(grid_for(t->n_hm, 256)), dim3(256), 0, s, t->hm_chr.p, t->hm_pos.p, t->n_hm, ctg, t->hom_hub.p, ctr.p);
            hipLaunchKernelGGL(k_hub_replay, dim3(1), dim3(64), 0, s, x, ctg, t->hom_hub.p);
        }
        HIPQ(hipEventRecord(ev[3], s));
        HIPQ(hipStreamSynchronize(s));
    } else {
        LoadCounters h;
        HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPQ(hipEventRecord(ev[3], s));
        HIPQ(hipStreamSynchronize(s));
        if (h.err_range) return fail(93, "the tables name %u positions outside the %u contigs given", h.err_range, nc);
    }
    HIPQ(hipGetLastError());

    cora_hip_tables_info &I = t->info;
    I.exact_records = t->n_eq; I.exact_entries = t->m_eq; I.inexact_nodes = t->n_hm; I.inexact_items = t->m_hm;
    I.hub_positions = t->G; I.d = t->d; I.chr_short = t->cb == 2;
    I.device_bytes = t->n_eq * 10 + (t->n_eq + 1) * 8 + t->m_eq * 7 + t->n_hm * 7 + (t->n_hm + 1) * 8 + t->m_hm * (8 + d) + t->G * 8;
    HIPQ(hipEventElapsedTime(&I.ms_parse, ev[0], ev[1]));
    HIPQ(hipEventElapsedTime(&I.ms_lookups, ev[1], ev[2]));
    HIPQ(hipEventElapsedTime(&I.ms_hubs, ev[2], ev[3]));
    HIPQ(hipEventElapsedTime(&I.ms_total, ev[0], ev[3]));
    return 0;
}

}  // namespace

extern "C" {

int cora_hip_tables_load(const uint8_t *exact, size_t exact_bytes, uint64_t exact_count,
                         const uint8_t *inexact, size_t inexact_bytes, uint64_t inexact_count,
                         const uint32_t *contig_len, int32_t n_contigs, int32_t d, int32_t chr_short, int32_t exact_only,
                         cora_hip_tables **out)
{
    g_err.clear();
    if (!out) return fail(1, "cora_hip_tables_load: out is NULL");
    *out = nullptr;
    if (g_device < 0) return fail(2, "cora_hip_init has not been called: no GPU, no loader (there is no CPU path)");
    if (d < 1 || d > 3) return fail(1, "d = %d: the tables hold 1 to 3 offsets per item", d);
    if (n_contigs < 0 || (n_contigs && !contig_len)) return fail(1, "contig lengths missing");
    if (n_contigs > (chr_short ? 65530 : 250)) return fail(1, "%d contigs: more than the %s-byte contig ids hold (mappingInference.cpp:42-46)", n_contigs, chr_short ? "2" : "1");
    if ((exact_bytes && !exact) || (!exact_only && inexact_bytes && !inexact)) return fail(1, "table bytes missing");
    HIPQ(hipSetDevice(g_device));
    cora_hip_tables *t = new cora_hip_tables();
    memset(&t->info, 0, sizeof t->info);
    t->d = d; t->cb = chr_short ? 2 : 1; t->exact_only = exact_only ? 1 : 0;
    t->len.assign(contig_len, contig_len + n_contigs);
    PoolScope no_pool(nullptr);
    int rc = load_impl(t, exact, exact_bytes, exact_count, inexact, exact_only ? 0 : inexact_bytes, exact_only ? 0 : inexact_count);
    if (rc) { (void)hipDeviceSynchronize(); delete t; return rc; }
    *out = t;
    return 0;
}

int cora_hip_tables_from_job(const cora_hip_job *job, int32_t exact_only, cora_hip_tables **out)
{
    g_err.clear();
    if (!job || !out) return fail(1, "cora_hip_tables_from_job: NULL argument");
    if (!job->stats.ms_total) return fail(1, "the job has not run yet");
    return cora_hip_tables_load(job->exact_bytes.data(), job->exact_bytes.size(), job->stats.exact_nodes,
                                job->inexact_bytes.data(), job->inexact_bytes.size(), job->stats.inexact_nodes,
                                job->hg.len.data(), (int32_t)job->hg.len.size(), job->prm.d, job->prm.chr_short, exact_only, out);
}

int cora_hip_tables_get_info(const cora_hip_tables *t, cora_hip_tables_info *out)
{
    if (!t || !out) return fail(1, "cora_hip_tables_get_info: NULL argument");
    *out = t->info;
    return 0;
}

int cora_hip_tables_array_ptr(const cora_hip_tables *t, int which, const void **device_ptr, uint64_t *n_elems, uint32_t *elem_bytes)
{
    if (!t) return fail(1, "cora_hip_tables_array_ptr: NULL tables");
    const void *p = nullptr; u64 n = 0; u32 eb = 0;
    switch (which) {
    case CORA_TA_EQ_CHR: p = t->eq_chr.p; n = t->n_eq; eb = 2; break;
    case CORA_TA_EQ_POS: p = t->eq_pos.p; n = t->n_eq; eb = 4; break;
    case CORA_TA_EQ_BLOCK: p = t->eq_block.p; n = t->n_eq; eb = 4; break;
    case CORA_TA_EQ_OFF: p = t->eq_off.p; n = t->n_eq + 1; eb = 8; break;
    case CORA_TA_EQI_DIR: p = t->eqi_dir.p; n = t->m_eq; eb = 1; break;
    case CORA_TA_EQI_CHR: p = t->eqi_chr.p; n = t->m_eq; eb = 2; break;
    case CORA_TA_EQI_POS: p = t->eqi_pos.p; n = t->m_eq; eb = 4; break;
    case CORA_TA_HM_CHR: p = t->hm_chr.p; n = t->n_hm; eb = 2; break;
    case CORA_TA_HM_POS: p = t->hm_pos.p; n = t->n_hm; eb = 4; break;
    case CORA_TA_HM_OFF: p = t->hm_off.p; n = t->n_hm + 1; eb = 8; break;
    case CORA_TA_HM_MAXREV: p = t->hm_maxrev.p; n = t->n_hm; eb = 1; break;
    case CORA_TA_HMI_DIR: p = t->hmi_dir.p; n = t->m_hm; eb = 1; break;
    case CORA_TA_HMI_CHR: p = t->hmi_chr.p; n = t->m_hm; eb = 2; break;
    case CORA_TA_HMI_POS: p = t->hmi_pos.p; n = t->m_hm; eb = 4; break;
    case CORA_TA_HMI_BLOCK: p = t->hmi_block.p; n = t->m_hm; eb = 1; break;
    case CORA_TA_HMI_OFFS: p = t->hmi_offs.p; n = t->m_hm * (u64)t->d; eb = 1; break;
    case CORA_TA_EQUIV_HUB: p = t->equiv_hub.p; n = t->G; eb = 4; break;
    case CORA_TA_HOM_HUB: p = t->hom_hub.p; n = t->G; eb = 4; break;
    default: return fail(1, "no such array: %d", which);
    }
    if (device_ptr) *device_ptr = p;
    if (n_elems) *n_elems = n;
    if (elem_bytes) *elem_bytes = eb;
    return 0;
}

int cora_hip_tables_export(const cora_hip_tables *t, int which, void *dst_host, uint64_t dst_bytes)
{
    const void *p; u64 n; u32 eb;
    RCQ(cora_hip_tables_array_ptr(t, which, &p, &n, &eb));
    if (dst_bytes < n * eb) return fail(1, "array %d needs %llu bytes, the buffer has %llu", which, (unsigned long long)(n * eb), (unsigned long long)dst_bytes);
    if (n && !dst_host) return fail(1, "cora_hip_tables_export: NULL buffer");
    if (n) {
        HIPQ(hipSetDevice(g_device));
        HIPQ(hipMemcpy(dst_host, p, n * eb, hipMemcpyDeviceToHost));
    }
    return 0;
}

int cora_hip_tables_lookup_equiv(const cora_hip_tables *t, const uint32_t *chr, const uint32_t *pos, uint64_t n, int on_device,
                                 uint32_t *class_index, int16_t *block_offset, uint8_t *dir)
{
    g_err.clear();
    if (!t) return fail(1, "cora_hip_tables_lookup_equiv: NULL tables");
    if (!n) return 0;
    if (!chr || !pos || !class_index || !block_offset || !dir) return fail(1, "cora_hip_tables_lookup_equiv: NULL array");
    HIPQ(hipSetDevice(g_device));
    PoolScope no_pool(nullptr);
    hipStream_t s = t->stream;
    DBuf<u32> dc, dp, di; DBuf<int16_t> dof; DBuf<u8> dd; DBuf<LoadCounters> ctr;
    const u32 *qc = chr, *qp = pos;
    u32 *ri = class_index; int16_t *ro = block_offset; u8 *rd = dir;
    if (!on_device) {
        RCQ(dc.alloc(n)); RCQ(dp.alloc(n)); RCQ(di.alloc(n)); RCQ(dof.alloc(n)); RCQ(dd.alloc(n));
        HIPQ(hipMemcpyAsync(dc.p, chr, n * 4, hipMemcpyHostToDevice, s));
        HIPQ(hipMemcpyAsync(dp.p, pos, n * 4, hipMemcpyHostToDevice, s));
        qc = dc.p; qp = dp.p; ri = di.p; ro = dof.p; rd = dd.p;
    }
    RCQ(ctr.alloc(1));
    HIPQ(hipMemsetAsync(ctr.p, 0, sizeof(LoadCounters), s));
    const ExactView x = {t->eq_chr.p, t->eq_pos.p, t->eq_block.p, t->eq_off.p, t->eqi_dir.p, t->eqi_chr.p, t->eqi_pos.p, t->n_eq, t->m_eq};
    const Contigs ctg = {t->d_start.p, t->d_len.p, (u32)t->len.size()};
    hipLaunchKernelGGL(k_lookup_equiv, dim3(grid_for(n, 256)), dim3(256), 0, s, x, ctg, t->equiv_hub.p, qc, qp, n, ri, ro, rd, ctr.p);
    LoadCounters h;
    HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
    if (!on_device) {
        HIPQ(hipMemcpyAsync(class_index, di.p, n * 4, hipMemcpyDeviceToHost, s));
        HIPQ(hipMemcpyAsync(block_offset, dof.p, n * 2, hipMemcpyDeviceToHost, s));
        HIPQ(hipMemcpyAsync(dir, dd.p, n, hipMemcpyDeviceToHost, s));
    }
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    if (h.err_range) return fail(94, "%u positions are in the hub of a class but in none of its ranges (the reference asserts, mappingInference.cpp:6673)", h.err_range);
    return 0;
}

void cora_hip_tables_destroy(cora_hip_tables *t)
{
    if (!t) return;
    (void)hipSetDevice(g_device);
    (void)hipDeviceSynchronize();
    delete t;
}

}  // extern "C"
