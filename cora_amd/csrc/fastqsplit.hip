// fastqsplit.hip -- CORA's fastqSplit on gfx950.
// FS = /root/reference/fastqSplit.cpp.  Every read-mer of the input (reads, mates, splits in file order) goes to the
// fragment file of the group its signal belongs to (first and last base, or first two and last two, FS:131-139), as a
// record of 8 bytes (2 x running read-mer index, FS:157-173) + its characters, in input order inside each file.
// Here: sequence lines located as in collapse (reads_common.hpp), one pass that classifies every read-mer and counts
// the groups, one stable 8-bit radix pass on the group number (the order of each file), and a copy kernel in which every
// lane produces 16 consecutive output bytes.  Byte work, HBM-bound: the FASTQ image is read about three times (line
// search twice, bodies once) and the records written once.
#include "job_internal.hpp"
#include "reads_common.hpp"

namespace {

struct FsCounters { unsigned long long bad_signal, short_lines; unsigned long long hist[256]; };

struct FsParams {
    u32 R, splits, sig_size, slip;     // slip: the reference's THREEWAY_PAIRED body of mate 1's third record (FS:579-580)
    u8 group_of[628];
};

__device__ __forceinline__ u32 sym_of(u32 c) { return c == 'N' ? 0u : c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : c == 'T' ? 4u : 5u; }

__global__ __launch_bounds__(CT) void k_fs_classify(const u8 *__restrict__ rd, const u64 *__restrict__ seq_at, const u64 *__restrict__ seq_end,
                                                    u64 n_items, FsParams prm, u64 *__restrict__ key, u32 *__restrict__ val, FsCounters *ctr)
{
    __shared__ u32 hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const u64 idx = (u64)blockIdx.x * CT + threadIdx.x;
    if (idx < n_items) {
        const u64 line = idx / prm.splits;
        const u32 s = (u32)(idx - line * prm.splits);
        const u64 at = seq_at[line];
        if (s == 0 && seq_end[line] - at < (u64)prm.R * prm.splits) atomicAdd(&ctr->short_lines, 1ull);
        const u8 *p = rd + at + (u64)s * prm.R;
        u32 g;
        if (prm.sig_size == 2) {
            const u32 a = sym_of(p[0]), b = sym_of(p[prm.R - 1]);
            if (a > 4 || b > 4) { atomicAdd(&ctr->bad_signal, 1ull); g = 0; }
            else g = prm.group_of[a * 5 + b];
        } else {
            const u32 a = sym_of(p[0]), b = sym_of(p[1]), c = sym_of(p[prm.R - 2]), d = sym_of(p[prm.R - 1]);
            if (a > 4 || b > 4 || c > 4 || d > 4) { atomicAdd(&ctr->bad_signal, 1ull); g = 0; }
            else g = prm.group_of[((a * 5 + b) * 5 + c) * 5 + d];
        }
        key[idx] = g; val[idx] = (u32)idx;
        atomicAdd(&hist[g], 1u);
    }
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&ctr->hist[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// 16 consecutive bytes of the concatenated fragment files per lane; record j of that stream is read-mer order[j]
__global__ __launch_bounds__(CT) void k_fs_write(const u8 *__restrict__ rd, const u64 *__restrict__ seq_at, const u32 *__restrict__ order,
                                                 u64 n_items, FsParams prm, u8 *__restrict__ out)
{
    const u64 t = (u64)blockIdx.x * CT + threadIdx.x;
    const u32 RS = prm.R + 8;
    const u64 total = n_items * RS, b0 = t * 16;
    if (b0 >= total) return;
    u64 j = b0 / RS;
    u32 o = (u32)(b0 - j * RS);
    u64 wlo = 0, whi = 0;                                    // the 16 bytes (no run-time register indexing)
    const u32 n = (u32)min((u64)16, total - b0);
    u32 i = 0;
    while (i < n) {                                          // at most two records meet in 16 bytes (a record is 10 bytes or more)
        const u64 src = order ? (u64)order[j] : j;
        const u64 line = src / prm.splits;
        u32 s = (u32)(src - line * prm.splits);
        if (prm.slip && s == 2 && !(line & 1)) s = 1;
        const u64 code = 2 * src;
        for (; o < 8 && i < n; o++, i++) {
            const u64 c = (code >> (8 * o)) & 0xFFull;
            if (i < 8) wlo |= c << (8 * i); else whi |= c << (8 * (i - 8));
        }
        if (i < n) {
            const u32 len = min(n - i, RS - o);
            ByteRun run(rd + seq_at[line] + (u64)s * prm.R + (o - 8), len);
            while (run.more()) {
                u32 cnt;
                u32 four = run.next(cnt);
                for (u32 q = 0; q < cnt; q++, i++, four >>= 8) {
                    const u64 c = four & 0xFFu;
                    if (i < 8) wlo |= c << (8 * i); else whi |= c << (8 * (i - 8));
                }
            }
            o += len;
        }
        if (o == RS) { o = 0; j++; }
    }
    *reinterpret_cast<uint4 *>(out + b0) = make_uint4((u32)wlo, (u32)(wlo >> 32), (u32)whi, (u32)(whi >> 32));
}

}  // namespace

struct cora_hip_fqsplit {
    cora_hip_fqsplit_params prm;
    hipStream_t stream = nullptr;
    std::vector<u8> h_rd;
    std::vector<ReadFile> files;
    HBuf<u8> out;
    DBuf<u8> d_out;                           // the fragment files, one behind the other, kept in HBM until the next run
    u64 group_start[257] = {};                // in records
    u32 RS = 0;
    bool device_only = false;                 // leave the fragment files in HBM (cora_hip_fqsplit_group_device), no copy to the host
    SortScratch sort_scratch;
    DevPool pool;
    cora_hip_fqsplit_stats st;
    ~cora_hip_fqsplit()
    {
        sort_scratch.release();
        d_out.release();
        pool.destroy();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int fs_run(cora_hip_fqsplit *c)
{
    hipStream_t s = c->stream;
    const cora_hip_fqsplit_params &prm = c->prm;
    memset(&c->st, 0, sizeof c->st);
    const auto t_begin = std::chrono::steady_clock::now();
    hipEvent_t ev[5];
    for (auto &e : ev) HIPQ(hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 5; i++) (void)hipEventDestroy(e[i]); } } guard{ev};

    FsParams fp;
    memset(&fp, 0, sizeof fp);
    fp.splits = (u32)prm.splits; fp.R = (u32)(prm.read_len / prm.splits); fp.sig_size = (u32)prm.signal_size;
    fp.slip = (prm.signal_size == 2 && prm.paired && prm.splits == 3) ? 1u : 0u;
    memcpy(fp.group_of, prm.group_of, sizeof fp.group_of);
    const u32 RS = fp.R + 8;
    c->RS = RS;

    DBuf<u8> rd;
    DBuf<FsCounters> ctr;
    DBuf<u64> tot, seq_at, seq_end;
    RCQ(rd.alloc(c->h_rd.size() + 1024)); RCQ(ctr.alloc(1)); RCQ(tot.alloc(4));
    HIPQ(hipMemsetAsync(rd.p + c->h_rd.size(), 0, 1024, s));
    HIPQ(hipMemsetAsync(ctr.p, 0, sizeof(FsCounters), s));
    if (!c->h_rd.empty()) HIPQ(hipMemcpyAsync(rd.p, c->h_rd.data(), c->h_rd.size(), hipMemcpyHostToDevice, s));
    HIPQ(hipEventRecord(ev[0], s));
    u64 n_rd = 0;
    RCQ(locate_sequence_lines(s, rd.p, c->h_rd, c->files, prm.paired != 0, tot.p, seq_at, n_rd, &seq_end));
    const u64 n_items = n_rd * fp.splits;
    if (n_items >= 0xFFFFFFFFull) return fail(19, "%llu read-mers: 32-bit record numbers do not hold that", (unsigned long long)n_items);
    HIPQ(hipEventRecord(ev[1], s));

    DBuf<u64> key, tkey;
    DBuf<u32> val, tval;
    RCQ(key.alloc(n_items + 1)); RCQ(tkey.alloc(n_items + 1)); RCQ(val.alloc(n_items + 1)); RCQ(tval.alloc(n_items + 1));
    if (n_items) hipLaunchKernelGGL(k_fs_classify, dim3(grid_of(n_items, CT)), dim3(CT), 0, s, rd.p, seq_at.p, seq_end.p, n_items, fp, key.p, val.p, ctr.p);
    FsCounters hc;
    HIPQ(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    if (hc.short_lines) return fail(78, "%llu sequence lines are shorter than %u bases (the reference would copy across the line end)", hc.short_lines, fp.R * fp.splits);
    if (hc.bad_signal) return fail(78, "%llu read-mers have a signal byte outside A C G T N (the reference's tables are indexed by it, FS:16-17)", hc.bad_signal);
    c->group_start[0] = 0;
    for (int g = 0; g < 256; g++) c->group_start[g + 1] = c->group_start[g] + hc.hist[g];
    for (int g = prm.n_groups; g < 256; g++) if (hc.hist[g]) return fail(19, "group table names group %d of %d", g, prm.n_groups);
    HIPQ(hipEventRecord(ev[2], s));

    // ---- the order of the records inside each file: a stable pass on the group number ----
    const u32 *order = nullptr;
    SortArrays ka, kt;
    ka.a = key.p; ka.c = val.p; kt.a = tkey.p; kt.c = tval.p;
    if (prm.n_groups > 1 && n_items) {
        if (radix_sort(s, ka, kt, n_items, 0, 8, c->sort_scratch, nullptr)) return fail(100, "radix sort failed");
        order = ka.c;
    }
    HIPQ(hipEventRecord(ev[3], s));

    const u64 total = n_items * RS;
    DBuf<u8> &d_out = c->d_out;
    RCQ(d_out.alloc(total + 32));
    if (total) hipLaunchKernelGGL(k_fs_write, dim3(grid_of((total + 15) / 16, CT)), dim3(CT), 0, s, rd.p, seq_at.p, order, n_items, fp, d_out.p);
    HIPQ(hipEventRecord(ev[4], s));
    if (!c->device_only) {
        RCQ(c->out.reserve(total + 1));
        if (total) HIPQ(hipMemcpyAsync(c->out.p, d_out.p, total, hipMemcpyDeviceToHost, s));
    }
    HIPQ(hipStreamSynchronize(s));
    HIPQ(hipGetLastError());
    cora_hip_fqsplit_stats &st = c->st;
    st.reads = n_rd; st.read_mers = n_items; st.record_bytes = RS; st.output_bytes = total; st.image_bytes = c->h_rd.size();
    HIPQ(hipEventElapsedTime(&st.ms_parse, ev[0], ev[1]));
    HIPQ(hipEventElapsedTime(&st.ms_classify, ev[1], ev[2]));
    HIPQ(hipEventElapsedTime(&st.ms_order, ev[2], ev[3]));
    HIPQ(hipEventElapsedTime(&st.ms_write, ev[3], ev[4]));
    HIPQ(hipEventElapsedTime(&st.ms_device, ev[0], ev[4]));
    st.ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

}  // namespace

extern "C" {

// FS:54-129: "<lines> <signal size>", then one blank-separated token per group: signals of <size> characters, each followed
// by one separator character; a token that starts with N also takes every signal that holds an N (and loses its first two bytes).
int cora_hip_fqsplit_parse_signals(const char *text, size_t bytes, cora_hip_fqsplit_params *out)
{
    if (!text || !out) return fail(10, "null argument");
    auto sym = [](unsigned char ch) { return ch == 'N' ? 0 : ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'T' ? 4 : -1; };
    size_t i = 0;
    auto token = [&](std::string &t) {
        while (i < bytes && isspace((unsigned char)text[i])) i++;
        t.clear();
        while (i < bytes && !isspace((unsigned char)text[i])) t.push_back(text[i++]);
        return !t.empty();
    };
    std::string t;
    if (!token(t)) return fail(7, "signals file: no header");
    if (!token(t)) return fail(7, "signals file: no signal size");
    const int size = atoi(t.c_str());
    if (size != 2 && size != 4) return fail(7, "fragsignalsize should be either 2 or 4 (size: %d)", size);      // FS:123-128
    memset(out->group_of, 0, sizeof out->group_of);
    int groups = 0;
    while (token(t)) {
        if (groups >= 255) return fail(7, "signals file: more than 255 groups (FS:12)");
        if (t[0] == 'N') {
            for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
                if (size == 2) { if (a == 0 || b == 0) out->group_of[a * 5 + b] = (uint8_t)groups; }
                else for (int c = 0; c < 5; c++) for (int d = 0; d < 5; d++)
                    if (a == 0 || b == 0 || c == 0 || d == 0) out->group_of[((a * 5 + b) * 5 + c) * 5 + d] = (uint8_t)groups;
            }
            t = t.size() >= 2 ? t.substr(2) : std::string();
        }
        const size_t step = (size_t)size + 1;
        if (t.size() % step) return fail(7, "signals file: group %d is not a whole number of signals", groups);   // the reference asserts (FS:79, 109)
        for (size_t k = 0; k < t.size() / step; k++) {
            int v = 0;
            for (int q = 0; q < size; q++) {
                const int sy = sym((unsigned char)t[k * step + q]);
                if (sy < 0) return fail(7, "signals file: byte 0x%02x in a signal of group %d", (unsigned char)t[k * step + q], groups);
                v = v * 5 + sy;
            }
            out->group_of[v] = (uint8_t)groups;
        }
        groups++;
    }
    out->signal_size = size;
    out->n_groups = groups;
    return 0;
}

int cora_hip_fqsplit_create(const cora_hip_fqsplit_params *p, cora_hip_fqsplit **out)
{
    if (!p || !out) return fail(10, "null argument");
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    if (p->splits < 1 || p->splits > 3) return fail(5, "ERROR: Mapping mode undefined... ");                      // FS:600-604
    if (p->signal_size != 2 && p->signal_size != 4) return fail(7, "fragsignalsize should be either 2 or 4");
    if (p->n_groups < 1 || p->n_groups > 255) return fail(19, "%d groups outside 1..255 (FS:12)", p->n_groups);
    if (p->read_len / p->splits < p->signal_size || p->read_len > 255) return fail(19, "read length %d outside %d..255 (FS:11)", p->read_len, p->signal_size * p->splits);
    cora_hip_fqsplit *c = new cora_hip_fqsplit();
    c->prm = *p;
    memset(&c->st, 0, sizeof c->st);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(100, "hipStreamCreate failed"); }
    *out = c;
    return 0;
}

int cora_hip_fqsplit_add_reads(cora_hip_fqsplit *c, const uint8_t *fastq, size_t bytes)
{
    if (!c || (!fastq && bytes)) return fail(10, "null argument");
    ReadFile f;
    f.off = (c->h_rd.size() + 15) & ~(size_t)15;
    f.bytes = bytes;
    c->h_rd.resize(f.off + bytes);
    if (bytes) memcpy(c->h_rd.data() + f.off, fastq, bytes);
    c->files.push_back(f);
    return 0;
}

int cora_hip_fqsplit_set_device_only(cora_hip_fqsplit *c, int32_t on)
{
    if (!c) return fail(10, "null argument");
    c->device_only = on != 0;
    return 0;
}

int cora_hip_fqsplit_run(cora_hip_fqsplit *c)
{
    if (!c) return fail(10, "null argument");
    PoolScope scope(&c->pool);
    return fs_run(c);
}

int cora_hip_fqsplit_group(const cora_hip_fqsplit *c, int32_t group, const uint8_t **bytes, size_t *n)
{
    if (!c || !bytes || !n) return fail(10, "null argument");
    if (group < 0 || group >= c->prm.n_groups) return fail(19, "group %d outside 0..%d", group, c->prm.n_groups - 1);
    if (c->device_only) return fail(19, "the fragment files were left in HBM (cora_hip_fqsplit_set_device_only): ask for cora_hip_fqsplit_group_device");
    *bytes = c->out.p ? c->out.p + c->group_start[group] * c->RS : nullptr;
    *n = (size_t)((c->group_start[group + 1] - c->group_start[group]) * c->RS);
    return 0;
}

int cora_hip_fqsplit_group_device(const cora_hip_fqsplit *c, int32_t group, const void **dev, size_t *n)
{
    if (!c || !dev || !n) return fail(10, "null argument");
    if (group < 0 || group >= c->prm.n_groups) return fail(19, "group %d outside 0..%d", group, c->prm.n_groups - 1);
    *dev = c->d_out.p ? c->d_out.p + c->group_start[group] * c->RS : nullptr;
    *n = (size_t)((c->group_start[group + 1] - c->group_start[group]) * c->RS);
    return 0;
}

int cora_hip_fqsplit_get_stats(const cora_hip_fqsplit *c, cora_hip_fqsplit_stats *out)
{
    if (!c || !out) return fail(10, "null argument");
    *out = c->st;
    return 0;
}

void cora_hip_fqsplit_destroy(cora_hip_fqsplit *c) { delete c; }

}  // extern "C"
