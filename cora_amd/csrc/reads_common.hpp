// reads_common.hpp -- what the read-side paths (collapse.hip, fastqsplit.hip) share: finding the sequence line of every
// FASTQ record on the device, and small launch / scan helpers.  Included into each translation unit's own anonymous namespace.
#pragma once
#include "job_internal.hpp"

namespace {

using namespace cora;

constexpr int CT = 256;

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
    for (int w = 0; w < CT / 64; w++) { u32 t = lds[w]; if (w < wv) base += t; tot += t; }
    if (total) *total = tot;
    __syncthreads();
    return base + x - v;
}

// ------------------------------------------------------------------ FASTQ: sequence line of every record
// First pass: every lane looks at 16 bytes and keeps which of them are line ends (16 bits); blocks count them.
__global__ __launch_bounds__(CT) void k_nl_count(const u8 *__restrict__ img, u64 n, u32 *__restrict__ blk, unsigned short *__restrict__ nl_mask)
{
    __shared__ u32 lds[4];
    const u64 t = (u64)blockIdx.x * CT + threadIdx.x, at = t * 16;
    u32 m = 0;
    if (at < n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(img + at);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) m |= (u32)((at + i < n) && ((w[i >> 2] >> (8 * (i & 3))) & 0xFF) == '\n') << i;
        nl_mask[t] = (unsigned short)m;
    }
    u32 tot;
    block_excl_scan((u32)__popc(m), lds, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// Second pass, over the masks alone (an eighth of the image's bytes): newline number j of the file (0-based) with
// j % 4 == 0 ends a header line: the sequence of record j / 4 starts behind it (and ends at newline j + 1: kept when seq_end is given)
__global__ __launch_bounds__(CT) void k_fq_emit(const unsigned short *__restrict__ nl_mask, u64 n, const u64 *__restrict__ blk_off, u64 file_off,
                                                u64 *__restrict__ seq_at, u64 *__restrict__ seq_end, u64 first_read, u32 stride, u64 n_records)
{
    __shared__ u32 lds[4];
    const u64 t = (u64)blockIdx.x * CT + threadIdx.x, at = t * 16;
    u32 m = at < n ? (u32)nl_mask[t] : 0u;
    u64 j = blk_off[blockIdx.x] + block_excl_scan((u32)__popc(m), lds, nullptr);
    while (m) {
        const u32 i = (u32)__ffs((int)m) - 1;
        m &= m - 1;
        if ((j & 3) == 0 && (j >> 2) < n_records) seq_at[(first_read + (j >> 2)) * stride] = file_off + at + i + 1;
        if (seq_end && (j & 3) == 1 && (j >> 2) < n_records) seq_end[(first_read + (j >> 2)) * stride] = file_off + at + i;
        j++;
    }
}

u32 grid_of(u64 n, u32 per) { const u64 g = (n + per - 1) / per; return (u32)(g ? g : 1); }

int scan_u32(hipStream_t s, const u32 *in, u64 *out, u64 n, u64 *total_dev)
{
    DBuf<u64> tmp;
    RCQ(tmp.alloc(scan_scratch_elems(n)));
    if (exclusive_scan_u32_to_u64(s, in, out, n, tmp.p, total_dev)) return fail(100, "scan launch failed");
    HIPQ(hipStreamSynchronize(s));            // tmp is released on return
    return 0;
}


// Bytes q[0 .. len) of an image, four at a time, through aligned 4-byte loads that stay inside the words holding those bytes
// (so nothing is read beyond a buffer's last word): a quarter of the load instructions of a byte-by-byte walk.
struct ByteRun {
    const u32 *wp;
    u32 sh, lo, k, len;
    __device__ __forceinline__ ByteRun(const u8 *q, u32 n)
    {
        sh = (u32)((size_t)q & 3);
        wp = reinterpret_cast<const u32 *>(q - sh);
        lo = n ? wp[0] : 0u;
        k = 0; len = n;
    }
    __device__ __forceinline__ bool more() const { return k < len; }
    // the next cnt (1..4) bytes in the low bytes of the result, first byte lowest
    __device__ __forceinline__ u32 next(u32 &cnt)
    {
        cnt = min(4u, len - k);
        const u32 hi = (sh + cnt > 4) ? wp[k / 4 + 1] : 0u;
        const u32 four = __builtin_amdgcn_alignbyte(hi, lo, sh);
        lo = (sh == 0 && k + 4 < len) ? wp[k / 4 + 1] : hi;
        k += 4;
        return four;
    }
};

struct ReadFile { u64 off, bytes; };          // a FASTQ image inside the uploaded buffer (16-byte aligned start)

// Sequence line of every record of every file: seq_at[line] = byte offset of the line's first base in the buffer, lines
// numbered in the reference's reading order (per sample: record by record, and for pairs mate 1 then mate 2).
// Records are four lines (CL:923-935); a file whose line count is no multiple of four is refused (status 7).
int locate_sequence_lines(hipStream_t s, const u8 *rd_dev, const std::vector<u8> &h_rd, const std::vector<ReadFile> &files, bool paired,
                          u64 *tot_dev, DBuf<u64> &seq_at, u64 &n_rd, DBuf<u64> *seq_end = nullptr)
{
    const u32 stride = paired ? 2u : 1u;
    if (files.size() % stride) return fail(6, "paired input needs two files per sample");
    std::vector<u64> file_records(files.size());
    std::vector<DBuf<u64> *> blk_offs;
    std::vector<DBuf<unsigned short> *> masks;
    struct Guard { std::vector<DBuf<u64> *> &v; std::vector<DBuf<unsigned short> *> &m;
                   ~Guard() { for (auto *p : v) delete p; for (auto *p : m) delete p; } } guard{blk_offs, masks};
    u64 n_reads = 0;                                        // records over all samples (pairs count once here)
    for (size_t f = 0; f < files.size(); f++) {
        const auto &fl = files[f];
        const u32 blocks = grid_of(fl.bytes, CT * 16);
        DBuf<u32> blk;
        RCQ(blk.alloc(blocks));
        DBuf<u64> *bo = new DBuf<u64>();
        blk_offs.push_back(bo);
        RCQ(bo->alloc(blocks));
        DBuf<unsigned short> *mk = new DBuf<unsigned short>();
        masks.push_back(mk);
        RCQ(mk->alloc((u64)blocks * CT));
        hipLaunchKernelGGL(k_nl_count, dim3(blocks), dim3(CT), 0, s, rd_dev + fl.off, fl.bytes, blk.p, mk->p);
        RCQ(scan_u32(s, blk.p, bo->p, blocks, tot_dev));
        u64 n_nl = 0;
        HIPQ(hipMemcpyAsync(&n_nl, tot_dev, 8, hipMemcpyDeviceToHost, s));
        HIPQ(hipStreamSynchronize(s));
        const u64 lines = n_nl + ((fl.bytes && h_rd[fl.off + fl.bytes - 1] != '\n') ? 1 : 0);
        if (lines % 4) return fail(7, "input file %zu holds %llu lines: not a multiple of 4 (FASTQ records are read 4 lines at a time)", f + 1, (unsigned long long)lines);
        file_records[f] = lines / 4;
        if (paired && (f & 1) && file_records[f] != file_records[f - 1])
            return fail(7, "mate files of sample %zu hold %llu and %llu records", f / 2 + 1, (unsigned long long)file_records[f - 1], (unsigned long long)file_records[f]);
        if (!paired || (f & 1)) n_reads += file_records[f];
    }
    n_rd = n_reads * stride;
    RCQ(seq_at.alloc(n_rd + 1));
    if (seq_end) RCQ(seq_end->alloc(n_rd + 1));
    u64 first = 0;
    for (size_t f = 0; f < files.size(); f++) {
        const auto &fl = files[f];
        const u32 mate = paired ? (u32)(f & 1) : 0u;
        hipLaunchKernelGGL(k_fq_emit, dim3(grid_of(fl.bytes, CT * 16)), dim3(CT), 0, s, masks[f]->p, fl.bytes, blk_offs[f]->p, fl.off,
                           seq_at.p + mate, seq_end ? seq_end->p + mate : (u64 *)nullptr, first, stride, file_records[f]);
        if (!paired || (f & 1)) first += file_records[f];
    }
    HIPQ(hipStreamSynchronize(s));                          // the per-file block offsets are released on return
    return 0;
}

}  // namespace
