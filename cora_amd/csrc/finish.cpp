// finish.cpp -- host side of the table build (see finish.hpp).
// HTS = /root/reference/homTable_setup.cpp.
#include "finish.hpp"
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include "kernels.hpp"

#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <tuple>

namespace cora {

uint32_t HostGenome::contig_of(uint64_t g) const
{
    // gstart has n_contigs+1 entries; the last one is G
    auto it = std::upper_bound(gstart.begin(), gstart.end() - 1, g);
    return (uint32_t)(it - gstart.begin()) - 1;
}

// HTS:2015-2066.  n = d+1 seeds; long read-mers: positions i < 8n carry seed 1 + i mod n and are
// mirrored at K-1-i; short ones: q = K/(2n) bases per half, the remainder in the middle unassigned.
void make_seed_template(int K, int d, char *tmpl, int *seed_q)
{
    const int n = d + 1;
    memset(tmpl, 0, (size_t)K);
    int span;
    if (K >= 16 * n) { span = 8 * n; *seed_q = 8; }
    else { span = K / 2 - (K % (2 * n)) / 2; *seed_q = K / (2 * n); }
    for (int i = 0; i < span; i++) { tmpl[i] = (char)(1 + i % n); tmpl[K - 1 - i] = (char)(1 + i % n); }
}

// HTS:2069-2116: first three positions of the seed from the left (so, ts1, ts2) and from the right
// (eo, te1, te2).
void seed_signal_offsets(int K, const char *tmpl, int seed, int off[6])
{
    int left[3] = {-1, -1, -1}, right[3] = {-1, -1, -1}, nl = 0, nr = 0;
    for (int i = 0; i < K && nl < 3; i++) if (tmpl[i] == seed) left[nl++] = i;
    for (int i = K - 1; i > left[2] && nr < 3; i--) if (tmpl[i] == seed) right[nr++] = i;
    off[0] = left[0]; off[1] = right[0]; off[2] = left[1]; off[3] = right[1]; off[4] = left[2]; off[5] = right[2];
}

// HTS:727-765 builds "split runs" of whole contigs (about total/(min(nchr,250)+2) bases each, the last
// contig never opens a run of its own); HTS:1467-1474 then tests only the LAST contig of a run
// against the 50 Mbp limit ('=' where '+=' was meant); if it is longer, every contig of the run that
// is longer than the limit is compacted window by window (HTS:1479,1528-1581).  d == 3 only (HTS:2852-2856).
void plan_windows(int d, const std::vector<uint32_t> &len, std::vector<uint32_t> &window)
{
    const int nchr = (int)len.size();
    window.assign(nchr, 0);
    if (d != 3 || nchr == 0) return;
    const uint32_t limit = 50000000u;
    int nruns = std::min(nchr, 250) + 2;
    std::vector<int> run_end(1, 0);                       // 1-based contig ids; run r covers (run_end[r], run_end[r+1]]
    long long total = 0;
    for (uint32_t l : len) total += l;
    const long long aimed = total / nruns;
    long long sofar = 0;
    for (int c = 1; c < nchr; c++) {                      // the loop stops at the last contig without adding it
        sofar += len[c - 1];
        if (sofar >= aimed) { run_end.push_back(c); sofar = 0; }
    }
    run_end.push_back(nchr);
    for (size_t r = 0; r + 1 < run_end.size(); r++) {
        const int a = run_end[r] + 1, b = run_end[r + 1];
        if (b < a) continue;
        if (len[b - 1] > limit)
            for (int c = a; c <= b; c++) if (len[c - 1] > limit) window[c - 1] = limit;
    }
}

// A table of a human-sized reference is written once, 0.8 GB of fresh memory: with 4 KB pages that is 2e5 page faults under
// the copying threads (80 of the 105 ms the one-shot executable spent assembling <Inexact>); large buffers are mapped
// with huge pages asked for.
static void bytebuf_release(ByteBuf &b)
{
    if (b.mapped) munmap(b.p, b.mapped); else free(b.p);
    b.p = nullptr; b.mapped = 0; b.cap = 0;
}
bool ByteBuf::reserve_uninit(size_t c)
{
    if (c <= cap && p) return true;
    if (fd >= 0) {                           // the file grows and is mapped anew; what it holds stays
        const size_t len = ((c ? c : 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (ftruncate(fd, (off_t)len) != 0) return false;
        void *m = p ? mremap(p, mapped, len, MREMAP_MAYMOVE) : mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (uint8_t *)m; mapped = len; cap = c;
        return true;
    }
    bytebuf_release(*this);
    if (c >= ((size_t)32 << 20)) {
        const size_t len = (c + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            (void)madvise(m, len, MADV_HUGEPAGE);
            p = (uint8_t *)m; mapped = len; cap = c;
            return true;
        }
    }
    p = (uint8_t *)malloc(c ? c : 1);
    cap = p ? c : 0;
    if (!p) n = 0;
    return p != nullptr;
}
bool ByteBuf::grow_keep(size_t c)
{
    if (c <= cap && p) return true;
    if (fd >= 0 || !p || !n) return reserve_uninit(c);
    ByteBuf grown;
    if (!grown.reserve_uninit(c)) return false;
    memcpy(grown.p, p, n);
    std::swap(p, grown.p); std::swap(cap, grown.cap); std::swap(mapped, grown.mapped);
    return true;
}
bool ByteBuf::open_file(const char *path)
{
    if (fd >= 0) { if (mapped) munmap(p, mapped); close(fd); fd = -1; p = nullptr; mapped = 0; cap = 0; }
    else bytebuf_release(*this);
    n = 0;
    fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    return fd >= 0;
}
bool ByteBuf::close_file()
{
    if (fd < 0) return false;
    bool ok = true;
    if (mapped && munmap(p, mapped) != 0) ok = false;
    if (ftruncate(fd, (off_t)n) != 0) ok = false;
    if (close(fd) != 0) ok = false;
    fd = -1; p = nullptr; mapped = 0; cap = 0;
    return ok;
}
ByteBuf::~ByteBuf()
{
    if (fd >= 0) { if (mapped) munmap(p, mapped); close(fd); fd = -1; p = nullptr; mapped = 0; }
    else bytebuf_release(*this);
}


namespace {

struct Writer {                                           // raw cursor into a buffer sized for the worst case
    uint8_t *at;
    int cw;                                               // contig id width: 1 or 2 bytes (HTS:210-214)
    void u8(uint32_t x) { *at++ = (uint8_t)x; }
    void chr(uint32_t c) { *at++ = (uint8_t)c; if (cw == 2) *at++ = (uint8_t)(c >> 8); }
    void u32(uint32_t x) { memcpy(at, &x, 4); at += 4; }  // little-endian host, as the reference's fwrite
    void i16(int x) { const int16_t v = (int16_t)x; memcpy(at, &v, 2); at += 2; }
};

}  // namespace

namespace {

struct DItem {                                            // one directed item in a list that needs the exact order
    uint32_t tgt;
    uint8_t dir, o1, o2, o3;
    int16_t blen;
    std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t> ord;
};

inline uint32_t base_of(const View<uint64_t> &pk, uint64_t gp)
{
    return (uint32_t)(pk[gp >> 5] >> (62 - 2 * (gp & 31))) & 3u;
}

// HTS:1159-1191: one candidate only -- walk forward from the cursor to the first target >= wanted,
// back to the last target <= wanted, then test that single item.
int search_one(const std::vector<DItem> &v, int last, uint32_t tgt, int dir, int o1, int o2, int o3)
{
    const int n = (int)v.size();
    int cur = last;
    while (cur < n && v[cur].tgt < tgt) cur++;
    if (cur >= n) cur = n - 1;
    while (cur >= 0 && v[cur].tgt > tgt) cur--;
    if (cur < 0) cur = 0;
    const DItem &x = v[cur];
    if (x.tgt != tgt || x.dir != dir || x.o1 != o1 || x.o2 != o2 || x.o3 != o3) return -1;
    return cur;
}

void emit_item(Writer &w, const HostGenome &g, int d, uint32_t tgt, uint32_t dir, uint32_t o1, uint32_t o2, uint32_t o3, int blen)
{
    const uint32_t ci = g.contig_of(tgt);
    w.u8(dir);
    w.chr(ci + 1);
    w.u32((uint32_t)(tgt - g.gstart[ci]) + 1);
    w.u8(o1);
    if (d > 1) w.u8(o2);
    if (d > 2) w.u8(o3);
    w.i16(blen);
}

}  // namespace

// F2 (HTS:1353-1432): per source position with a surviving item: chr,pos { dir chr pos off.. blockLen }* 0xFF,
// items by decreasing block length through std::sort (HTS:1368; unstable above 16 items, which is why the
// very same library routine is applied to the very same pre-order here).
static bool serialize_inexact_range(const HostGenome &g, const InexactDeviceResult &r, const View<uint64_t> &packed,
                                    size_t node_begin, size_t node_end, ByteBuf &out, FinishStats &st, size_t *marks)
{
    // marks[nd - node_begin] = offset in `out` where list nd's bytes start (lists are written in order)
    const int cw = g.chr_short ? 2 : 1;
    const size_t n_items = (size_t)r.node_start[node_end] - r.node_start[node_begin];
    if (!out.reserve_uninit(n_items * (size_t)(1 + cw + 4 + g.d + 2) + (node_end - node_begin) * (size_t)(cw + 5) + 16)) return false;
    Writer w{out.p, cw};
    st.inexact_nodes = st.dirty_nodes = st.big_nodes = 0;
    const size_t n_nodes = node_end;
    auto src_of = [&](size_t nd) { return (uint32_t)(r.key[r.node_start[nd]] >> 32); };
    auto same_win = [&](uint32_t a, uint32_t b, uint32_t ci) {
        const uint32_t wl = g.window[ci];
        if (!wl) return true;
        return (a - g.gstart[ci]) / wl == (b - g.gstart[ci]) / wl;
    };
    int sig_off[4][6];
    for (int s = 1; s <= g.d + 1; s++) seed_signal_offsets(g.K, g.tmpl, s, sig_off[s - 1]);

    std::vector<std::pair<int16_t, uint32_t>> order;
    size_t i = node_begin;
    while (i < n_nodes) {
        // maximal run of consecutive source positions inside one compaction window
        const uint32_t s0 = src_of(i);
        const uint32_t ci = g.contig_of(s0);
        size_t j = i + 1;
        while (j < n_nodes && src_of(j) == src_of(j - 1) + 1 && same_win(src_of(j - 1), src_of(j), ci)) j++;
        bool dirty = false;
        for (size_t nd = i; nd < j && !dirty; nd++)
            for (uint32_t k = r.node_start[nd] + 1; k < r.node_start[nd + 1]; k++)
                if (r.key[k] == r.key[k - 1]) { dirty = true; break; }
        if (!dirty) {
            for (size_t nd = i; nd < j; nd++) {
                const uint32_t a = r.node_start[nd], b = r.node_start[nd + 1];
                if (marks) marks[nd - node_begin] = (size_t)(w.at - out.p);
                if (b - a == 1) {                                    // the common case: one item, nothing to order
                    if (r.blen[a] != -1) {
                        const uint32_t sp = src_of(nd), m = r.meta[a];
                        w.chr(ci + 1); w.u32((uint32_t)(sp - g.gstart[ci]) + 1);
                        emit_item(w, g, g.d, (uint32_t)r.key[a], m & 1u, meta_o1(m), meta_o2(m), meta_o3(m), r.blen[a]);
                        w.u8(0xFF); st.inexact_nodes++;
                    }
                    continue;
                }
                order.clear();
                for (uint32_t k = a; k < b; k++) order.emplace_back(r.blen[k], k);
                if (b - a > 16) st.big_nodes++;
                std::sort(order.begin(), order.end(),
                          [](const std::pair<int16_t, uint32_t> &x, const std::pair<int16_t, uint32_t> &y) { return x.first > y.first; });
                bool header = false;
                for (auto &e : order) {
                    if (e.first == -1) continue;
                    if (!header) { const uint32_t sp = src_of(nd); w.chr(ci + 1); w.u32((uint32_t)(sp - g.gstart[ci]) + 1); header = true; }
                    const uint32_t m = r.meta[e.second];
                    emit_item(w, g, g.d, (uint32_t)r.key[e.second], m & 1u, meta_o1(m), meta_o2(m), meta_o3(m), e.first);
                }
                if (header) { w.u8(0xFF); st.inexact_nodes++; }
            }
            i = j;
            continue;
        }
        // ---- a list of the run has tied targets: redo the run in the reference's own order ----
        std::vector<std::vector<DItem>> L(j - i);
        for (size_t nd = i; nd < j; nd++) {
            std::vector<DItem> &lst = L[nd - i];
            const uint32_t a = r.node_start[nd], b = r.node_start[nd + 1];
            bool tie = false;
            for (uint32_t k = a; k < b; k++) {
                const uint32_t src = (uint32_t)(r.key[k] >> 32), tgt = (uint32_t)r.key[k], m = r.meta[k];
                if (k > a && r.key[k] == r.key[k - 1]) tie = true;
                DItem it;
                it.tgt = tgt; it.dir = (uint8_t)(m & 1u); it.o1 = (uint8_t)meta_o1(m); it.o2 = (uint8_t)meta_o2(m); it.o3 = (uint8_t)meta_o3(m);
                it.blen = 1;
                // position of the pair record in the reference's pair files (HTS:2671-2800) and of this
                // item among the pushes (HTS:1591-1619): file t, group g, seed s, scan position of k-mer 1,
                // pass (normal / palindrome), chain walk = later insertions first, duplicate, push side.
                const uint32_t p1 = std::max(src, tgt), p2 = std::min(src, tgt);
                const int *so = sig_off[meta_seed0(m)];
                const uint32_t grp = g.sig2[base_of(packed, (uint64_t)p1 + so[0]) * 4 + base_of(packed, (uint64_t)p1 + so[1])];
                const uint32_t thr = g.sig4[base_of(packed, (uint64_t)p1 + so[2]) * 64 + base_of(packed, (uint64_t)p1 + so[4]) * 16 +
                                            base_of(packed, (uint64_t)p1 + so[5]) * 4 + base_of(packed, (uint64_t)p1 + so[3])];
                const uint32_t side = (src == p1) ? 0u : 1u;
                it.ord = std::make_tuple(thr, grp, meta_seed0(m), p1, meta_pass(m), ~p2, meta_dup(m) * 2 + side);
                lst.push_back(it);
            }
            if (tie) st.dirty_nodes++;
            if (b - a > 16) st.big_nodes++;
            std::sort(lst.begin(), lst.end(), [](const DItem &x, const DItem &y) { return x.ord < y.ord; });      // push order
            std::sort(lst.begin(), lst.end(), [](const DItem &x, const DItem &y) { return x.tgt < y.tgt; });      // HTS:1141
        }
        const size_t nl = L.size();
        for (size_t k = 0; k < nl; k++) {
            std::vector<DItem> &base = L[k];
            for (size_t dist = 1; k + dist < nl; dist++) {                                                        // HTS:1203-1270
                std::vector<DItem> &comp = L[k + dist];
                int last = 0;
                bool live = false;                         // an item can only grow past dist if it reached dist: else nothing is left to do
                for (const DItem &b : base) if (b.blen == (int)dist) { live = true; break; }
                if (!live) break;
                for (DItem &b : base) {
                    if (b.blen != (int)dist) continue;
                    const uint32_t tc = g.contig_of(b.tgt);
                    const long long pos1 = (long long)(b.tgt - g.gstart[tc]) + 1;
                    const long long want = b.dir == 0 ? pos1 + (long long)dist : pos1 - (long long)dist;
                    if (want <= 0 || want > (long long)g.len[tc]) continue;
                    const int o1 = (int)b.o1 - (int)dist, o2 = b.o2 ? (int)b.o2 - (int)dist : 0, o3 = b.o3 ? (int)b.o3 - (int)dist : 0;
                    if (o1 <= 0 || o2 < 0 || o3 < 0) continue;
                    const uint32_t want_g = b.dir == 0 ? b.tgt + (uint32_t)dist : b.tgt - (uint32_t)dist;
                    const int at = search_one(comp, last, want_g, b.dir, o1, o2, o3);
                    if (at != -1) { last = at; b.blen++; comp[at].blen = -1; }
                }
            }
            std::sort(base.begin(), base.end(), [](const DItem &x, const DItem &y) { return x.blen > y.blen; });  // HTS:1368
            if (marks) marks[i + k - node_begin] = (size_t)(w.at - out.p);
            bool header = false;
            for (const DItem &b : base) {
                if (b.blen == -1) continue;
                if (!header) { const uint32_t sp = s0 + (uint32_t)k; w.chr(ci + 1); w.u32((uint32_t)(sp - g.gstart[ci]) + 1); header = true; }
                emit_item(w, g, g.d, b.tgt, b.dir, b.o1, b.o2, b.o3, b.blen);
            }
            if (header) { w.u8(0xFF); st.inexact_nodes++; }
        }
        i = j;
    }
    out.n = (size_t)(w.at - out.p);
    return true;
}

bool serialize_inexact(const HostGenome &g, const InexactDeviceResult &r, View<uint64_t> packed,
                       ByteBuf &out, FinishStats &st, int n_threads, std::vector<size_t> *marks)
{
    if (marks) marks->assign(r.node_start.empty() ? 1 : r.node_start.size(), 0);
    const size_t n_nodes = r.node_start.empty() ? 0 : r.node_start.size() - 1;
    st.inexact_nodes = st.dirty_nodes = st.big_nodes = 0;
    out.n = 0;
    if (n_nodes == 0) return true;
    auto src_of = [&](size_t nd) { return (uint32_t)(r.key[r.node_start[nd]] >> 32); };
    // a thread's range must start where a run of consecutive source positions starts (runs never
    // interact: HTS:1212 stops at the first position without a list)
    auto run_starts_at = [&](size_t nd) {
        if (nd == 0 || src_of(nd) != src_of(nd - 1) + 1) return true;
        const uint32_t ci = g.contig_of(src_of(nd));
        const uint32_t wl = g.window[ci];
        return wl != 0 && (src_of(nd - 1) - g.gstart[ci]) / wl != (src_of(nd) - g.gstart[ci]) / wl;
    };
    // one thread per 16 384 items (starting a thread costs about what serialising a few thousand items does: 16 threads on the
    // 10 MB table of a 64 Mbp reference made the whole build 1.5 ms slower than 4)
    int T = n_threads < 1 ? 1 : n_threads;
    T = (int)std::min<size_t>((size_t)T, r.key.size() / 16384 + 1);
    std::vector<size_t> cut(1, 0);
    for (int t = 1; t < T; t++) {
        // split by items, not nodes, to balance
        const uint64_t want_item = (uint64_t)r.key.size() * t / T;
        size_t nd = (size_t)(std::upper_bound(r.node_start.p, r.node_start.p + n_nodes, (uint32_t)want_item) - r.node_start.p);
        if (nd > 0) nd--;
        while (nd < n_nodes && !run_starts_at(nd)) nd++;
        if (nd > cut.back() && nd < n_nodes) cut.push_back(nd);
    }
    cut.push_back(n_nodes);
    const size_t parts = cut.size() - 1;
    std::vector<ByteBuf> bufs(parts);
    std::vector<FinishStats> sts(parts);
    std::vector<std::thread> th;
    std::vector<char> ok(parts, 1);
    std::vector<double> part_ms(parts, 0.0);
    auto one = [&](size_t k) {
        const auto t0 = std::chrono::steady_clock::now();
        ok[k] = serialize_inexact_range(g, r, packed, cut[k], cut[k + 1], bufs[k], sts[k], marks ? marks->data() + cut[k] : nullptr);
        part_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (size_t k = 1; k < parts; k++) th.emplace_back(one, k);
    one(0);
    for (auto &x : th) x.join();
    if (getenv("CORA_TRACE")) {
        fprintf(stderr, "[cora]   host lists, per thread (ms / items / lists finished in the reference's order):");
        for (size_t k = 0; k < parts; k++)
            fprintf(stderr, " %.2f/%u/%llu", part_ms[k], (unsigned)(r.node_start[cut[k + 1]] - r.node_start[cut[k]]), (unsigned long long)sts[k].dirty_nodes);
        fprintf(stderr, "\n");
    }
    for (size_t k = 0; k < parts; k++) if (!ok[k]) return false;
    std::vector<size_t> off(parts + 1, 0);
    for (size_t k = 0; k < parts; k++) {
        off[k + 1] = off[k] + bufs[k].size();
        if (marks) for (size_t nd = cut[k]; nd < cut[k + 1]; nd++) (*marks)[nd] += off[k];
        st.inexact_nodes += sts[k].inexact_nodes; st.dirty_nodes += sts[k].dirty_nodes; st.big_nodes += sts[k].big_nodes;
    }
    if (!out.reserve_uninit(off[parts] + 16)) return false;
    out.n = off[parts];
    th.clear();
    for (size_t k = 1; k < parts; k++)
        th.emplace_back([&, k]() { if (bufs[k].size()) memcpy(out.p + off[k], bufs[k].data(), bufs[k].size()); });
    if (bufs[0].size()) memcpy(out.p, bufs[0].data(), bufs[0].size());
    for (auto &x : th) x.join();
    if (marks) (*marks)[n_nodes] = off[parts];
    return true;
}

// The device wrote the bytes of its lists back to back (`stream`); the host lists (in list order, each with the
// stream offset its bytes belong at) are finished here and spliced in.
bool assemble_inexact(const HostGenome &g, View<uint8_t> stream, View<HostNode> hnodes, const InexactDeviceResult &host_items,
                      View<uint64_t> packed, uint64_t device_nodes, ByteBuf &out, FinishStats &st, int n_threads,
                      const StreamArrival *arrival)
{
    ByteBuf hb;
    std::vector<size_t> marks;
    FinishStats hst;
    const bool trace = getenv("CORA_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[cora]   assemble: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (hnodes.size() && !serialize_inexact(g, host_items, packed, hb, hst, n_threads, &marks)) return false;
    mark("host lists");
    st.inexact_nodes = device_nodes + hst.inexact_nodes;
    st.dirty_nodes = hst.dirty_nodes; st.big_nodes = hst.big_nodes;
    const size_t total = stream.size() + hb.size();
    if (!out.reserve_uninit(total + 16)) return false;
    out.n = total;
    mark("table buffer");
    // the stream arrives in steps; a piece is copied once the stream bytes it reads are there (need), host pieces at once
    std::vector<size_t> upto;
    if (arrival) upto = arrival->upto;
    if (upto.empty() || upto.back() != stream.size()) upto.push_back(stream.size());
    struct Piece { const uint8_t *src; size_t dst, n, need; };
    std::vector<Piece> pieces;
    const size_t chunk = 1u << 18;
    size_t step = 0;
    auto add_host = [&](const uint8_t *src, size_t dst, size_t n) {
        for (size_t o = 0; o < n; o += chunk) pieces.push_back({src + o, dst + o, std::min(chunk, n - o), 0});
    };
    auto add_stream = [&](size_t from, size_t dst, size_t n) {           // stream bytes from .. from+n, cut at the arrival steps
        while (n) {
            while (from >= upto[step]) step++;
            const size_t m = std::min(std::min(chunk, n), upto[step] - from);
            pieces.push_back({stream.p + from, dst, m, from + m});
            from += m; dst += m; n -= m;
        }
    };
    size_t s_at = 0, d_at = 0;
    for (size_t i = 0; i < hnodes.size(); i++) {
        const size_t so = (size_t)hnodes[i].stream_off;
        if (so > s_at) { add_stream(s_at, d_at, so - s_at); d_at += so - s_at; s_at = so; }
        const size_t hn = marks[i + 1] - marks[i];
        if (hn) { add_host(hb.p + marks[i], d_at, hn); d_at += hn; }
    }
    if (stream.size() > s_at) add_stream(s_at, d_at, stream.size() - s_at);
    int T = n_threads < 1 ? 1 : n_threads;
    T = (int)std::min<size_t>((size_t)T, total / ((size_t)4 << 20) + 1);      // one thread per 4 MB to move
    mark("piece list");
    size_t lo = 0;
    for (size_t c = 0; c < upto.size(); c++) {
        if (arrival && arrival->wait && !arrival->wait(c)) return false;
        size_t hi = lo;
        while (hi < pieces.size() && pieces[hi].need <= upto[c]) hi++;
        const int Tc = (hi - lo) < 8 ? 1 : T;
        auto work = [&](int t) { for (size_t i = lo + (size_t)t; i < hi; i += (size_t)Tc) memcpy(out.p + pieces[i].dst, pieces[i].src, pieces[i].n); };
        std::vector<std::thread> th;
        for (int t = 1; t < Tc; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        lo = hi;
        mark("stream part moved");
    }
    return true;
}

}  // namespace cora
