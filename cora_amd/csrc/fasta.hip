// fasta.hip -- FASTA structure scan on gfx950: the contig index that `cora faiGenerate` computes with an awk
// one-liner (/root/reference/cora.cpp:1006-1029), and a reference loader that does not need the fixed line
// width ReadReference assumes (/root/reference/homTable_setup.cpp:96-142, HTS below).  SURVEY.md section 8f.3.
//
// The awk program's view of a line: $1 = its first run of bytes that are neither space nor tab; a line whose $1
// starts with '>' opens a contig named by the rest of $1; every other line adds length($1) to the contig's
// length; the offset of a contig is the byte after its header line; lineBases is length($1) of the first
// sequence line after a header (and keeps its old value when a header is followed by another header).
//
// Kernels (byte work, HBM bound: the image is read twice, 16 B per lane):
//   k_sep_count / k_sep_emit   positions of all separator bytes (newline, space, tab) and of the newlines
//   k_line_info                one thread per line: $1 from the separator list in O(1 + leading blanks)
//   scans                      header rank and sequence-byte prefix per line
//   k_contig_emit/_fill        one record per contig
//   k_conform                  does every contig keep the fixed line width?  (then the image is used as it is)
//   k_compact                  otherwise: the $1 bytes of the sequence lines, contig after contig, one wave per line
#include "job_internal.hpp"
#include <chrono>
#include <thread>

using namespace cora;

namespace {

constexpr int FT = 256;            // threads per workgroup
constexpr int FB = 16;             // bytes per thread
constexpr int FTILE = FT * FB;     // 4096 bytes per workgroup

struct ContigRec {                 // one per header line, device -> host
    u64 name_at;                   // byte after the '>'
    u64 length;                    // sum of length($1) of its sequence lines
    u64 offset;                    // byte after the header line's newline
    u64 first_tok_at;              // where the first sequence line's $1 starts (if has_first)
    u32 name_len;
    u32 first_line_bases;          // length($1) of the line after the header, if that is a sequence line
    u32 has_first;
    u32 pad;
};

struct ScanCounters {
    unsigned long long n_sep, n_nl, n_hdr, n_seq, violations, long_tokens;
};

__device__ __forceinline__ bool is_sep(u32 c) { return c == '\n' || c == ' ' || c == '\t'; }

__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *lds, u32 *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wv] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FT / 64; w++) { u32 s = lds[w]; if (w < wv) base += s; tot += s; }
    if (total) *total = tot;
    __syncthreads();
    return base + x - v;
}

// the 16 bytes of this thread; bytes at or beyond n read as 0 (the buffer is padded)
__device__ __forceinline__ void load16(const u8 *__restrict__ img, u64 at, u64 n, u32 w[4])
{
    if (at < n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(img + at);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if (at + 16 > n)
            for (int i = 0; i < 16; i++) if (at + i >= n) w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
    } else { w[0] = w[1] = w[2] = w[3] = 0; }
}

__global__ __launch_bounds__(FT) void k_sep_count(const u8 *__restrict__ img, u64 n, u32 *__restrict__ blk_sep, u32 *__restrict__ blk_nl)
{
    __shared__ u32 lds[8];
    u32 w[4];
    load16(img, ((u64)blockIdx.x * FT + threadIdx.x) * FB, n, w);
    u32 ns = 0, nn = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32 c = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
        ns += is_sep(c); nn += (c == '\n');
    }
    u32 ts, tn;
    block_excl_scan(ns, lds, &ts);
    block_excl_scan(nn, lds + 4, &tn);
    if (threadIdx.x == 0) { blk_sep[blockIdx.x] = ts; blk_nl[blockIdx.x] = tn; }
}

__global__ __launch_bounds__(FT) void k_sep_emit(const u8 *__restrict__ img, u64 n, const u64 *__restrict__ sep_off,
                                                 const u64 *__restrict__ nl_off, u64 *__restrict__ S, u64 *__restrict__ NL,
                                                 u64 *__restrict__ NLS)
{
    __shared__ u32 lds[8];
    u32 w[4];
    const u64 at = ((u64)blockIdx.x * FT + threadIdx.x) * FB;
    load16(img, at, n, w);
    u32 ns = 0, nn = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32 c = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
        ns += is_sep(c); nn += (c == '\n');
    }
    u64 si = sep_off[blockIdx.x] + block_excl_scan(ns, lds, nullptr);
    u64 ni = nl_off[blockIdx.x] + block_excl_scan(nn, lds + 4, nullptr);
    if (!ns) return;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u32 c = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
        if (is_sep(c)) {
            S[si] = at + i;
            if (c == '\n') { NL[ni] = at + i; NLS[ni] = si; ni++; }
            si++;
        }
    }
}

// One thread per line k = bytes (NL[k-1], NL[k]).  $1 = [t, u): t skips the leading blanks (they are consecutive
// entries of S right at the line start), u = the next separator.
__global__ __launch_bounds__(FT) void k_line_info(const u8 *__restrict__ img, const u64 *__restrict__ S, const u64 *__restrict__ NL,
                                                  const u64 *__restrict__ NLS, u64 n_lines, u64 *__restrict__ tok_at,
                                                  u32 *__restrict__ tok_len, u32 *__restrict__ hdr_flag, u32 *__restrict__ seq_len,
                                                  ScanCounters *ctr)
{
    const u64 k = (u64)blockIdx.x * FT + threadIdx.x;
    if (k > n_lines) return;
    if (k == n_lines) { hdr_flag[k] = 0; seq_len[k] = 0; return; }        // the scans' closing element
    const u64 s = k ? NL[k - 1] + 1 : 0, e = NL[k];
    u64 lo = k ? NLS[k - 1] + 1 : 0;
    const u64 hi = NLS[k];
    u64 t = s;
    while (lo < hi && S[lo] == t) { lo++; t++; }
    const u64 u = lo < hi ? S[lo] : e;
    u64 len = u - t;
    if (len > 0xFFFFFFFFull) { atomicAdd(&ctr->long_tokens, 1ull); len = 0xFFFFFFFFull; }
    const bool hdr = len > 0 && img[t] == '>';
    tok_at[k] = t; tok_len[k] = (u32)len;
    hdr_flag[k] = hdr; seq_len[k] = hdr ? 0u : (u32)len;
}

__global__ __launch_bounds__(FT) void k_contig_emit(const u32 *__restrict__ hdr_flag, const u64 *__restrict__ H, u64 n_lines,
                                                    u64 *__restrict__ ctg_line)
{
    const u64 k = (u64)blockIdx.x * FT + threadIdx.x;
    if (k < n_lines && hdr_flag[k]) ctg_line[H[k]] = k;
}

__global__ __launch_bounds__(FT) void k_contig_fill(const u64 *__restrict__ ctg_line, u64 nc, u64 n_lines, const u64 *__restrict__ NL,
                                                    const u64 *__restrict__ P, const u64 *__restrict__ tok_at,
                                                    const u32 *__restrict__ tok_len, const u32 *__restrict__ hdr_flag,
                                                    ContigRec *__restrict__ out)
{
    const u64 c = (u64)blockIdx.x * FT + threadIdx.x;
    if (c >= nc) return;
    const u64 k = ctg_line[c], k2 = c + 1 < nc ? ctg_line[c + 1] : n_lines;
    ContigRec r;
    r.name_at = tok_at[k] + 1; r.name_len = tok_len[k] - 1;
    r.length = P[k2] - P[k];
    r.offset = NL[k] + 1;
    r.has_first = (k + 1 < n_lines && !hdr_flag[k + 1]) ? 1u : 0u;
    r.first_line_bases = r.has_first ? tok_len[k + 1] : 0u;
    r.first_tok_at = r.has_first ? tok_at[k + 1] : 0;
    r.pad = 0;
    out[c] = r;
}

// Lines that break ReadReference's assumption (HTS:96-142: a header starts its line, every sequence line of a
// contig holds exactly W bases but the last, which holds 1..W; nothing else in the file).
__global__ __launch_bounds__(FT) void k_conform(const u64 *__restrict__ NL, const u64 *__restrict__ H, const u64 *__restrict__ tok_at,
                                                const u32 *__restrict__ tok_len, const u32 *__restrict__ hdr_flag, u64 n_lines, u32 W,
                                                ScanCounters *ctr)
{
    const u64 k = (u64)blockIdx.x * FT + threadIdx.x;
    bool bad = false;
    if (k < n_lines) {
        const u64 s = k ? NL[k - 1] + 1 : 0, e = NL[k];
        const u32 len = tok_len[k];
        if (tok_at[k] != s) bad = true;
        else if (!hdr_flag[k]) {
            if (H[k] == 0 || (u64)len != e - s || len == 0 || len > W) bad = true;
            else if (len != W && k + 1 < n_lines && !hdr_flag[k + 1]) bad = true;
        }
    }
    const u64 b = __ballot(bad);
    if (b && (threadIdx.x & 63) == 0) atomicAdd(&ctr->violations, (unsigned long long)__popcll(b));
}

// One wave per line: copies $1 of every sequence line that belongs to a contig to its place in the dense image
// (contig c starts at P[its header] - P0 + c: one '\n' closes every contig).  Header lines write the '\n' that
// closes the contig before them.
__global__ __launch_bounds__(FT) void k_compact(const u8 *__restrict__ img, const u64 *__restrict__ H, const u64 *__restrict__ P,
                                                const u64 *__restrict__ tok_at, const u32 *__restrict__ tok_len,
                                                const u32 *__restrict__ hdr_flag, u64 n_lines, u64 P0, u8 *__restrict__ dense)
{
    const u64 k = (u64)blockIdx.x * (FT / 64) + (threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63;
    if (k >= n_lines) return;
    const u64 h = H[k];
    if (hdr_flag[k]) {
        if (h >= 1 && lane == 0) dense[P[k] - P0 + (h - 1)] = '\n';
        return;
    }
    if (h == 0) return;                                   // text before the first header belongs to no contig
    const u64 dst = P[k] - P0 + (h - 1), src = tok_at[k];
    const u32 len = tok_len[k];
    for (u32 i = lane; i < len; i += 64) dense[dst + i] = img[src + i];
}

u32 grid_of(u64 n, u32 per) { const u64 g = (n + per - 1) / per; return (u32)(g ? g : 1); }

}  // namespace

struct cora_hip_fasta {
    // Every device array of the scan -- the ones kept below and its temporaries -- comes from this pool, so that nothing is given
    // back to the driver (hipFree synchronises the device, and the dozen releases at the end of a 3.1 GB scan took 11 ms, 163 ms in
    // the first process on a freshly leased GPU box) before cora_hip_fasta_destroy.  Declared first: it outlives the arrays.
    DevPool pool;
    hipStream_t stream = nullptr;
    u64 n = 0, n_scan = 0, n_lines = 0, nc = 0, bases = 0;
    DBuf<u8> img;
    DBuf<u64> NL, H, P, tok_at;
    DBuf<u32> tok_len, hdr_flag;
    std::vector<ContigRec> ctg;
    std::vector<std::string> names;
    std::vector<u64> line_bases;          // the awk program's lineBases per contig (stale values included)
    std::string fai;
    cora_hip_fasta_info info;
    ~cora_hip_fasta()
    {
        if (stream) (void)hipStreamDestroy(stream);
        img.release(); NL.release(); H.release(); P.release(); tok_at.release(); tok_len.release(); hdr_flag.release();
        pool.destroy();
    }
};

namespace {

// A large image in pageable memory (the executables pass an mmap of the file): a plain hipMemcpy moves it at 10-20 GB/s,
// one thread feeding the runtime's own staging buffer (0.15-0.3 s for a 3.1 GB reference, a quarter of the drop-in's wall
// time).  Here a few host threads copy alternate chunks into pinned buffers of their own and send each on its way at once.
int upload_pageable(u8 *dst, const u8 *src, size_t bytes, hipStream_t s)
{
    static const size_t CHUNK = (size_t)(getenv("CORA_UPLOAD_CHUNK_MB") ? atoi(getenv("CORA_UPLOAD_CHUNK_MB")) : 8) << 20;
    // three workers: 75-78 ms for the 3.1 GB image with 2 to 4 of them, 83-90 ms with 6 or 12 (profiles/e2e_variants.py; reading the page
    // cache is not the limit -- 6 threads copy 70-100 GB/s there, profiles/io_probe.cpp -- the copy engines' queue is)
    static const int WORKERS = getenv("CORA_UPLOAD_WORKERS") ? std::max(1, std::min(32, atoi(getenv("CORA_UPLOAD_WORKERS")))) : 3;
    if (bytes < (size_t)64 * CHUNK) { HIPQ(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); return 0; }
    int dev = 0;
    HIPQ(hipGetDevice(&dev));
    const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
    std::vector<int> status(WORKERS, 0);
    std::vector<std::thread> th;
    for (int w = 0; w < WORKERS; w++)
        th.emplace_back([&, w] {
            hipStream_t ws = nullptr;
            u8 *buf[2] = {nullptr, nullptr};
            hipEvent_t ev[2] = {nullptr, nullptr};
            auto bad = [&](hipError_t e) { if (e != hipSuccess && !status[w]) status[w] = (int)e; return e != hipSuccess; };
            if (bad(hipSetDevice(dev)) || bad(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking))) return;
            for (int b = 0; b < 2 && !status[w]; b++) { bad(hipHostMalloc((void **)&buf[b], CHUNK, hipHostMallocDefault)); bad(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
            int b = 0;
            for (size_t c = (size_t)w; c < n_chunks && !status[w]; c += WORKERS, b ^= 1) {
                const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
                if (c >= (size_t)2 * WORKERS && bad(hipEventSynchronize(ev[b]))) break;       // this buffer's last copy has left
                memcpy(buf[b], src + off, len);
                if (bad(hipMemcpyAsync(dst + off, buf[b], len, hipMemcpyHostToDevice, ws)) || bad(hipEventRecord(ev[b], ws))) break;
            }
            if (ws) bad(hipStreamSynchronize(ws));
            for (int k = 0; k < 2; k++) { if (buf[k]) (void)hipHostFree(buf[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
            if (ws) (void)hipStreamDestroy(ws);
        });
    for (auto &t : th) t.join();
    for (int w = 0; w < WORKERS; w++) if (status[w]) return fail(100, "upload of the FASTA image failed: %s", hipGetErrorString((hipError_t)status[w]));
    (void)s;
    return 0;
}

int scan_impl(cora_hip_fasta *f, const u8 *image, size_t bytes)
{
    HIPQ(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    hipStream_t s = f->stream;
    f->n = bytes;
    const bool virt_nl = bytes && image[bytes - 1] != '\n';   // awk treats an unterminated last line as a record
    f->n_scan = bytes + (virt_nl ? 1 : 0);
    static const bool trace = getenv("CORA_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    RCQ(f->img.alloc(f->n_scan + 64));
    if (trace) fprintf(stderr, "[cora] fasta: image buffer of %zu bytes after %.3f ms\n", bytes, since());
    if (bytes) RCQ(upload_pageable(f->img.p, image, bytes, s));          // (returns when a large image has arrived)
    if (trace) fprintf(stderr, "[cora] fasta: image uploaded after %.3f ms\n", since());
    HIPQ(hipMemsetAsync(f->img.p + bytes, 0, f->n_scan + 64 - bytes, s));
    if (virt_nl) HIPQ(hipMemsetAsync(f->img.p + bytes, '\n', 1, s));

    hipEvent_t ev[2];
    for (auto &e : ev) HIPQ(hipEventCreate(&e));
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 2; i++) (void)hipEventDestroy(e[i]); } } guard{ev};
    HIPQ(hipEventRecord(ev[0], s));

    DBuf<ScanCounters> ctr;
    RCQ(ctr.alloc(1));
    HIPQ(hipMemsetAsync(ctr.p, 0, sizeof(ScanCounters), s));
    const u32 blocks = grid_of(f->n_scan, FTILE);
    DBuf<u32> blk_sep, blk_nl;
    DBuf<u64> sep_off, nl_off, scan_tmp;
    RCQ(blk_sep.alloc(blocks)); RCQ(blk_nl.alloc(blocks)); RCQ(sep_off.alloc(blocks)); RCQ(nl_off.alloc(blocks));
    RCQ(scan_tmp.alloc(scan_scratch_elems(blocks)));
    hipLaunchKernelGGL(k_sep_count, dim3(blocks), dim3(FT), 0, s, f->img.p, f->n_scan, blk_sep.p, blk_nl.p);
    RCQ(exclusive_scan_u32_to_u64(s, blk_sep.p, sep_off.p, blocks, scan_tmp.p, (u64 *)&ctr.p->n_sep));
    RCQ(exclusive_scan_u32_to_u64(s, blk_nl.p, nl_off.p, blocks, scan_tmp.p, (u64 *)&ctr.p->n_nl));
    ScanCounters h;
    HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    const u64 n_sep = h.n_sep, n_lines = h.n_nl;
    f->n_lines = n_lines;

    DBuf<u64> S, NLS;
    RCQ(S.alloc(n_sep)); RCQ(NLS.alloc(n_lines)); RCQ(f->NL.alloc(n_lines));
    hipLaunchKernelGGL(k_sep_emit, dim3(blocks), dim3(FT), 0, s, f->img.p, f->n_scan, sep_off.p, nl_off.p, S.p, f->NL.p, NLS.p);

    RCQ(f->tok_at.alloc(n_lines)); RCQ(f->tok_len.alloc(n_lines)); RCQ(f->hdr_flag.alloc(n_lines + 1));
    DBuf<u32> seq_len;
    RCQ(seq_len.alloc(n_lines + 1));
    RCQ(f->H.alloc(n_lines + 1)); RCQ(f->P.alloc(n_lines + 1));
    hipLaunchKernelGGL(k_line_info, dim3(grid_of(n_lines + 1, FT)), dim3(FT), 0, s, f->img.p, S.p, f->NL.p, NLS.p, n_lines, f->tok_at.p,
                       f->tok_len.p, f->hdr_flag.p, seq_len.p, ctr.p);
    DBuf<u64> scan_tmp2;
    RCQ(scan_tmp2.alloc(scan_scratch_elems(n_lines + 1)));
    RCQ(exclusive_scan_u32_to_u64(s, f->hdr_flag.p, f->H.p, n_lines + 1, scan_tmp2.p, (u64 *)&ctr.p->n_hdr));
    RCQ(exclusive_scan_u32_to_u64(s, seq_len.p, f->P.p, n_lines + 1, scan_tmp2.p, (u64 *)&ctr.p->n_seq));
    HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
    HIPQ(hipStreamSynchronize(s));
    if (h.long_tokens) return fail(51, "a FASTA line holds a token of 4 GiB or more");
    const u64 nc = h.n_hdr;
    f->nc = nc;
    if (nc > 0x7FFFFFFFull) return fail(51, "more than 2^31 - 1 contigs");

    f->ctg.resize(nc);
    if (nc) {
        DBuf<u64> ctg_line;
        DBuf<ContigRec> rec;
        RCQ(ctg_line.alloc(nc)); RCQ(rec.alloc(nc));
        hipLaunchKernelGGL(k_contig_emit, dim3(grid_of(n_lines, FT)), dim3(FT), 0, s, f->hdr_flag.p, f->H.p, n_lines, ctg_line.p);
        hipLaunchKernelGGL(k_contig_fill, dim3(grid_of(nc, FT)), dim3(FT), 0, s, ctg_line.p, nc, n_lines, f->NL.p, f->P.p, f->tok_at.p,
                           f->tok_len.p, f->hdr_flag.p, rec.p);
        HIPQ(hipMemcpyAsync(f->ctg.data(), rec.p, nc * sizeof(ContigRec), hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
    }

    // lineBases keeps its value over headers that no sequence line follows (the awk variable is only set by the
    // first sequence line after a header); 0 before it was ever set
    f->line_bases.resize(nc);
    f->names.resize(nc);
    u64 lb = 0, bases = 0;
    for (u64 c = 0; c < nc; c++) {
        const ContigRec &r = f->ctg[c];
        if (r.has_first) lb = r.first_line_bases;
        f->line_bases[c] = lb;
        f->names[c].assign((const char *)image + r.name_at, r.name_len);
        bases += r.length;
    }
    f->bases = bases;

    // is the file what ReadReference can read?  W as the awk program reports it for the first contig
    const u32 W = nc && f->ctg[0].has_first ? f->ctg[0].first_line_bases : 0;
    u64 violations = 1;
    if (nc && W) {
        hipLaunchKernelGGL(k_conform, dim3(grid_of(n_lines, FT)), dim3(FT), 0, s, f->NL.p, f->H.p, f->tok_at.p, f->tok_len.p, f->hdr_flag.p,
                           n_lines, W, ctr.p);
        HIPQ(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        violations = h.violations;
        for (u64 c = 0; c < nc && !violations; c++) if (!f->ctg[c].has_first) violations = 1;     // a contig without bases
    }
    HIPQ(hipEventRecord(ev[1], s));
    HIPQ(hipStreamSynchronize(s));

    // the .fai text (cora.cpp:1023: name, length, offset, lineBases, lineBases + 1; one more line from END when
    // there is no contig at all: empty name, the bytes seen, zeros)
    std::string &t = f->fai;
    char buf[128];
    for (u64 c = 0; c < nc; c++) {
        t += f->names[c];
        snprintf(buf, sizeof buf, "\t%llu\t%llu\t%llu\t%llu\n", (unsigned long long)f->ctg[c].length, (unsigned long long)f->ctg[c].offset,
                 (unsigned long long)f->line_bases[c], (unsigned long long)f->line_bases[c] + 1);
        t += buf;
    }
    if (!nc) {
        snprintf(buf, sizeof buf, "\t%llu\t0\t0\t1\n", (unsigned long long)h.n_seq);
        t += buf;
    }

    cora_hip_fasta_info &in = f->info;
    memset(&in, 0, sizeof in);
    in.image_bytes = bytes; in.lines = n_lines; in.contigs = nc; in.bases = bases;
    in.conforming = violations == 0; in.line_width = (int32_t)W;
    HIPQ(hipEventElapsedTime(&in.ms_scan, ev[0], ev[1]));
    if (trace) fprintf(stderr, "[cora] fasta: scanned after %.3f ms (kernels %.3f ms)\n", since(), in.ms_scan);
    return 0;
}

}  // namespace

extern "C" {

int cora_hip_fasta_scan(const uint8_t *image, size_t image_bytes, cora_hip_fasta **out)
{
    if ((!image && image_bytes) || !out) return fail(51, "null argument");
    if (g_device < 0) return fail(102, "cora_hip_init has not succeeded");
    cora_hip_fasta *f = new cora_hip_fasta();
    PoolScope own_pool(&f->pool);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = scan_impl(f, image, image_bytes);
    if (getenv("CORA_TRACE")) fprintf(stderr, "[cora] fasta: scan and release of its temporaries %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc) { delete f; return rc; }
    *out = f;
    return 0;
}

int cora_hip_fasta_get_info(const cora_hip_fasta *f, cora_hip_fasta_info *out)
{
    if (!f || !out) return fail(51, "null argument");
    *out = f->info;
    return 0;
}

int cora_hip_fasta_fai_text(const cora_hip_fasta *f, const char **text, size_t *bytes)
{
    if (!f || !text || !bytes) return fail(51, "null argument");
    *text = f->fai.data(); *bytes = f->fai.size();
    return 0;
}

int cora_hip_fasta_contig(const cora_hip_fasta *f, int32_t index, const char **name, uint64_t *length, uint64_t *offset, uint64_t *line_bases)
{
    if (!f || index < 0 || (u64)index >= f->nc) return fail(51, "contig index out of range");
    if (name) *name = f->names[index].c_str();
    if (length) *length = f->ctg[index].length;
    if (offset) *offset = f->ctg[index].offset;
    if (line_bases) *line_bases = f->line_bases[index];
    return 0;
}

void cora_hip_fasta_destroy(cora_hip_fasta *f) { delete f; }

int cora_hip_job_set_reference_fasta(cora_hip_job *job, const cora_hip_fasta *f)
{
    if (!job || !f) return fail(51, "bad reference arguments");
    if (f->nc == 0) return fail(51, "Reference file is not properly formatted: no '>' line");
    PoolScope no_pool(nullptr);
    const u64 nc = f->nc;
    std::vector<cora_hip_contig> ctg(nc);
    for (u64 c = 0; c < nc; c++) {
        if (f->ctg[c].length > 0xFFFFFFFFull) return fail(19, "contig %llu is longer than 2^32 - 1 bases", (unsigned long long)c + 1);
        ctg[c].length = (u32)f->ctg[c].length; ctg[c].reserved = 0;
    }
    if (f->info.conforming) {
        for (u64 c = 0; c < nc; c++) ctg[c].file_offset = f->ctg[c].offset;
        return set_reference_impl(job, ctg.data(), (int32_t)nc, (u32)f->info.line_width, f->n, nullptr, f->img.p, nullptr);
    }
    // ragged lines, blank lines, trailing words: gather the sequence bytes, contig after contig, each closed by '\n'
    u64 at = 0;
    for (u64 c = 0; c < nc; c++) { ctg[c].file_offset = at; at += f->ctg[c].length + 1; }
    const u64 dense_bytes = at;
    DBuf<u8> dense;
    RCQ(dense.alloc(dense_bytes + 16));
    cora_hip_fasta *ff = const_cast<cora_hip_fasta *>(f);
    hipStream_t s = ff->stream;
    u64 P0 = 0;                                             // sequence bytes ahead of the first header
    {
        // P of the first header line = bytes that belong to no contig
        u64 total = 0;
        HIPQ(hipMemcpyAsync(&total, f->P.p + f->n_lines, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        P0 = total - f->bases;
    }
    HIPQ(hipMemsetAsync(dense.p, '\n', dense_bytes + 16, s));   // the last contig's closing byte and the tail
    hipLaunchKernelGGL(k_compact, dim3(grid_of(f->n_lines, FT / 64)), dim3(FT), 0, s, f->img.p, f->H.p, f->P.p, f->tok_at.p, f->tok_len.p,
                       f->hdr_flag.p, f->n_lines, P0, dense.p);
    HIPQ(hipStreamSynchronize(s));
    return set_reference_impl(job, ctg.data(), (int32_t)nc, 0x7FFFFFFFu, dense_bytes, nullptr, nullptr, &dense);
}

}  // extern "C"
