// radix_sort.hpp -- LDS-staged stable LSD radix sort and prefix scans for gfx950.
// Host-side launch interface; kernels are in radix_sort.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cora {

typedef uint64_t u64;
typedef uint32_t u32;

// A record set in structure-of-arrays form: a (u64, always), b (u64, optional
// low key word), c (u32, optional payload).  The logical key is a:b (128 bit)
// when b is present, else a (64 bit).
struct SortArrays {
    u64 *a = nullptr;
    u64 *b = nullptr;
    u32 *c = nullptr;
    // Optional (keys only): b is the first of b_chain_n arrays of b_stride words each, one behind the other in one allocation.
    // Every pass then writes the second word into the NEXT array of that chain (back to the first after the last) instead of
    // ping-ponging with tmp.b, which is not used: a pass whose second word goes to a higher address than it came from is the
    // faster one (radix_sort.hip, "the second word climbs").
    int b_chain_n = 0;
    size_t b_stride = 0;
};

struct SortTiming {          // accumulated over calls; events are created lazily
    float scatter_ms = 0.f;  // time inside scatter launches
    u32 scatter_launches = 0;
    u64 scatter_bytes = 0;   // algorithmic bytes: every record read once and written once per pass
    bool enabled = false;
};

// Scratch for the per-tile digit histograms (grown on demand, owned by caller).
struct SortScratch {
    u32 *tile_hist = nullptr;    // [tiles][256]
    u32 *chunk_sums = nullptr;   // [chunks][256]
    size_t cap_tiles = 0, cap_chunks = 0;
    u32 xcd_group = 0;           // consecutive tiles handed to one XCD by the scatter (0 = not chosen yet)
    hipEvent_t ev[2 * 16] = {};  // one pair per pass of a sort (created when a sort is first timed)
    bool external = false;       // tile_hist / chunk_sums are the caller's memory: never grown, never freed
    // [records] the NEXT pass's digit of every record, in output order, written by a scatter pass beside the records: the next pass's
    // histogram then reads one byte per record instead of the key word (radix_sort.hip, "the next digit travels ahead")
    unsigned char *next_digit = nullptr;      // what the sort at hand uses: the lease if one is bound and large enough, else the scratch's own array
    unsigned char *own_next_digit = nullptr;  // grown with hipMalloc on demand, up to RS_OWN_NEXT_DIGIT_MAX records (beyond: no digit bytes, key-word histograms)
    size_t cap_own_next_digit = 0;
    unsigned char *lease_next_digit = nullptr;   // caller's memory for the time of some sorts (a job lends pool memory: a separate 3 GB hipMalloc
    size_t cap_lease_next_digit = 0;             // made a process 50 ms slower to end)
    int reserve_next_digit(u64 n);
    int reserve(u64 n);
    void release();
    // Use caller-owned device memory instead of hipMalloc: hist holds tiles * 256 u32, chunks 2 * chunk_count * 256 u32.
    void bind_external(u32 *hist, size_t tiles, u32 *chunks, size_t chunk_count);
};

constexpr int RS_MAX_PASSES = 16;
constexpr int RS_THREADS = 256;                   // smallest workgroup of the sort kernels (see radix_sort())
constexpr int RS_IPT = 16;
constexpr int RS_TILE_MIN = RS_THREADS * RS_IPT;  // 4096 records: the smallest tile
constexpr int RS_CHUNK_TILES = 64;                // tiles per column-scan workgroup
constexpr size_t RS_OWN_NEXT_DIGIT_MAX = (size_t)1 << 28;

// Sorts n records by key bits [bit_lo, bit_hi) (bit 0 = LSB of the logical key,
// bit_lo is rounded down to a multiple of 8).  Stable.  `keys` and `tmp` are
// ping-pong buffers of equal shape; on return `keys` names the sorted data
// (the two structs are swapped after an odd number of passes).
// first_hist (optional): the [tiles of radix_sort_tile_records()][256] histogram of the first pass's digit, already
// computed by whoever produced the records; it is consumed (turned into offsets in place).
u32 radix_sort_tile_records(bool has_b, bool has_c);      // records per tile for that record shape
// One stable pass grouping the records by owning rank (multi-GPU flow): owner = (top 32 bits of keys.a [dmode 2: times the golden
// ratio, mod 2^32]) * world >> 32.  All arrays present in `keys` move together; counts[r] = records of rank r; result in `keys`.
int radix_partition(hipStream_t s, SortArrays &keys, SortArrays &tmp, u64 n, int dmode, u32 world, SortScratch &scratch, SortTiming *timing,
                    uint64_t *counts);
// A sort whose first pass reads ALL n records of `src` and keeps those owned by rank `me` (owner as above, fmode = dmode): the
// multi-GPU flow on few GPUs, where every rank scans the whole reference instead of exchanging records.  First
// radix_sort_count_kept (histogram + offsets of the first digit over the kept records; *n_kept tells the caller how big keys / tmp
// must be), then radix_sort with the same bit range and `from`.  The digit word of the first pass must be word a.
struct SortSource {
    SortArrays src;
    u64 n = 0;
    int fmode = 1;
    u32 world = 1, me = 0;
};
int radix_sort_count_kept(hipStream_t s, const SortSource &from, int bit_lo, int bit_hi, SortScratch &scratch, u64 *n_kept);
// first_digits_ready: the producer of the records has left the first pass's digit of every record in scratch.next_digit (asked
// radix_sort_next_digit_wanted first, reserved with reserve_next_digit): the first histogram reads those bytes.
bool radix_sort_next_digit_wanted(const SortScratch &scratch, u64 n);
int radix_sort(hipStream_t s, SortArrays &keys, SortArrays &tmp, u64 n, int bit_lo, int bit_hi,
               SortScratch &scratch, SortTiming *timing, u32 *first_hist = nullptr, const SortSource *from = nullptr,
               bool first_digits_ready = false);

// out[i] = sum_{j<i} in[j]  (u32 in, u64 out); *total_dev (device u64) gets the grand total.
// scratch_dev must hold at least scan_scratch_elems(n) u64.
size_t scan_scratch_elems(u64 n);
int exclusive_scan_u32_to_u64(hipStream_t s, const u32 *in, u64 *out, u64 n, u64 *scratch_dev, u64 *total_dev);

}  // namespace cora
