// sortbench -- microbenchmark of the radix sort kernels on the record shapes of the path (HIP events,
// per-pass figures), next to a plain copy of the same bytes (measured stream peak) and rocPRIM's
// radix_sort_pairs (a yardstick only; the product does not link rocPRIM).
//   sortbench [n_records=64000000] [reps=5]
#include "../radix_sort.hpp"
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <utility>

using namespace cora;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(u64 *a, u32 *c, u64 n, u64 seed)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        a[i] = z;
        if (c) c[i] = (u32)i;
    }
}
__global__ void k_copy(const u64 *__restrict__ a, const u32 *__restrict__ c, u64 *__restrict__ ao, u32 *__restrict__ co, u64 n)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        ao[i] = a[i];
        if (c) co[i] = c[i];
    }
}

__global__ void k_read(const ulonglong2 *__restrict__ a, u64 n2, u64 *out)
{
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (u64)gridDim.x * blockDim.x) { ulonglong2 v = a[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567) out[0] = acc;
}

// experiments on the per-tile histogram: MODE 0 = LDS atomics, 1 = loads only, 2 = wave-private LDS atomics, 3 = two tiles per block
template <int MODE>
__global__ __launch_bounds__(256) void x_hist(const u64 *__restrict__ w, u64 n, int shift, u32 *__restrict__ tile_hist)
{
    __shared__ u32 h[4][256];
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int q = 0; q < 4; q++) h[q][tid] = 0;
    __syncthreads();
    const int per = MODE == 3 ? 2 : 1;
    for (int rep = 0; rep < per; rep++) {
        const u64 base = ((u64)blockIdx.x * per + rep) * 4096;
        if (base + 4096 > n) break;
        const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w + base);
        ulonglong2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = w2[k * 256 + tid];
        if (MODE == 1) {
            u64 acc = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) acc += v[k].x ^ v[k].y;
            h[0][tid] += (u32)(acc >> shift);
        } else {
            const int q = MODE == 2 ? wv : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                atomicAdd(&h[q][(u32)(v[k].x >> shift) & 255u], 1u);
                atomicAdd(&h[q][(u32)(v[k].y >> shift) & 255u], 1u);
            }
        }
        if (MODE == 3) {
            __syncthreads();
            tile_hist[((size_t)blockIdx.x * per + rep) * 256 + tid] = h[0][tid];
            h[0][tid] = 0;
            __syncthreads();
        }
    }
    if (MODE != 3) {
        __syncthreads();
        tile_hist[(size_t)blockIdx.x * 256 + tid] = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    }
}

// sortbench <n> <reps> gaps: the six... the four arrays of a two-word sort cut from ONE allocation at varying distances
// (whole GB between them), all in one process: how the scatter's pace depends on where the arrays lie relative to each
// other, with nothing else changing.
static int gaps_mode(u64 n, int reps)
{
    const size_t GBy = (size_t)1 << 30;
    const size_t arr = (n * 8 + GBy - 1) / GBy * GBy;              // every array begins at a multiple of 1 GB
    const size_t total = 4 * arr + 24 * GBy;
    char *arena; CK(hipMalloc(&arena, total));
    hipStream_t s; CK(hipStreamCreate(&s));
    SortScratch scr;
    // swap: the sort's input lies where its spare arrays lay before, and the other way round
    struct G { int g1, g2, g3, swap; } gs[] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {1, 2, 3, 0}, {4, 0, 0, 0}, {0, 4, 0, 0}, {0, 0, 5, 0},
                                     {0, 0, 0, 1}, {1, 2, 3, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}, {0, 0, 0, 0}};
    for (const G &g : gs) {
        u64 *a = (u64 *)arena, *ta = (u64 *)(arena + arr + g.g1 * GBy), *b = (u64 *)(arena + 2 * arr + (g.g1 + g.g2) * GBy),
            *tb = (u64 *)(arena + 3 * arr + (g.g1 + g.g2 + g.g3) * GBy);
        if (g.swap == 1) { std::swap(a, ta); std::swap(b, tb); }
        if (g.swap == 2) std::swap(a, ta);
        if (g.swap == 3) std::swap(b, tb);
        float pass_ms[4] = {0, 0, 0, 0};
        for (int r = 0; r < reps; r++) {
            k_fill<<<4096, 256, 0, s>>>(a, nullptr, n, 7 + r);
            k_fill<<<4096, 256, 0, s>>>(b, nullptr, n, 107 + r);
            SortArrays k, t; k.a = a; k.b = b; t.a = ta; t.b = tb;
            SortTiming tm; tm.enabled = true;
            if (radix_sort(s, k, t, n, 96, 128, scr, &tm)) { fprintf(stderr, "sort failed\n"); return 1; }
            for (int q = 0; q < 4; q++) { float ms; CK(hipEventElapsedTime(&ms, scr.ev[2 * q], scr.ev[2 * q + 1])); pass_ms[q] += ms / reps; }
        }
        printf("swap %d  distances in GB  a->spare a %2zu  spare a->b %2zu  b->spare b %2zu  (a->b %2zu, spare a->spare b %2zu):  passes %.2f %.2f %.2f %.2f ms\n",
               g.swap, arr / GBy + g.g1, arr / GBy + g.g2, arr / GBy + g.g3, 2 * arr / GBy + g.g1 + g.g2, 2 * arr / GBy + g.g2 + g.g3, pass_ms[0], pass_ms[1], pass_ms[2], pass_ms[3]);
        fflush(stdout);
    }
    return 0;
}

// sortbench <gb> <reps> regions: one allocation of <gb> GB; every 4 GB piece of it is read as a stream and written the way a
// scatter tile writes (256 runs of 128 bytes per workgroup, evenly spaced over the piece): is every part of an allocation
// equally fast?
__global__ __launch_bounds__(256) void k_piece_write(const u64 *__restrict__ in, u64 *__restrict__ out, u64 region_words)
{
    const u32 tile = blockIdx.x;                           // 256 runs x 16 words; the source is a small buffer read over and over
    const u64 *src = in + (u64)(tile & 4095u) * 4096;
    for (u32 k = 0; k < 16; k++) {
        const u32 i = k * 256 + threadIdx.x, d = i / 16, o = i % 16;
        out[(u64)d * region_words + (u64)tile * 16 + o] = src[i];
    }
}
static int regions_mode(u64 gb, int reps)
{
    const size_t GBy = (size_t)1 << 30, piece = 4 * GBy;
    char *arena; CK(hipMalloc(&arena, gb * GBy));
    u64 *src; CK(hipMalloc(&src, 4096 * 4096 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_fill<<<4096, 256, 0, s>>>(src, nullptr, 4096 * 4096, 3);
    k_fill<<<4096, 256, 0, s>>>((u64 *)arena, nullptr, gb * GBy / 8, 5);
    float ms;
    for (size_t off = 0; off + piece <= gb * GBy; off += piece) {
        u64 *p = (u64 *)(arena + off);
        const u64 words = piece / 8, region_words = words / 256;
        const u32 tiles = (u32)(region_words / 16);
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) k_piece_write<<<tiles, 256, 0, s>>>(src, p, region_words);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double w = (double)piece * reps / ms / 1e6;
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) k_read<<<8192, 256, 0, s>>>((const ulonglong2 *)p, words / 2, src);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("piece at %3zu GB: scattered 128-byte runs written at %5.0f GB/s, stream read at %5.0f GB/s\n", off / GBy, w, (double)piece * reps / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 3 && !strcmp(argv[3], "regions")) return regions_mode(strtoull(argv[1], 0, 10), atoi(argv[2]));
    if (argc > 3 && !strcmp(argv[3], "gaps")) return gaps_mode(strtoull(argv[1], 0, 10), atoi(argv[2]));
    const u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 64000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const bool only_sorts = argc > 3 && !strcmp(argv[3], "sorts");      // skip the ceilings and the rocPRIM yardstick
    u64 *a, *ta; u32 *c, *tc;
    // SB_SHIFT_A / _B / _C: the spare buffer of that array begins this many bytes (a multiple of 16) into its allocation
    const size_t sh_a = getenv("SB_SHIFT_A") ? strtoull(getenv("SB_SHIFT_A"), 0, 0) : 0, sh_b = getenv("SB_SHIFT_B") ? strtoull(getenv("SB_SHIFT_B"), 0, 0) : 0,
                 sh_c = getenv("SB_SHIFT_C") ? strtoull(getenv("SB_SHIFT_C"), 0, 0) : 0;
    const size_t room = 0;
    // SB_ROUND: every allocation's size is rounded up to a multiple of this many bytes (how the driver backs an allocation
    // depends on whether its size is a multiple of 2 MB)
    const size_t rnd = getenv("SB_ROUND") ? strtoull(getenv("SB_ROUND"), 0, 0) : 1;
    auto R = [rnd](size_t x) { return (x + rnd - 1) / rnd * rnd; };
    // SB_ONE: all six arrays are cut from ONE allocation (each beginning at a multiple of SB_ROUND bytes)
    const bool one = getenv("SB_ONE") != nullptr;
    char *arena = nullptr;
    size_t arena_at = 0;
    if (one) CK(hipMalloc(&arena, 2 * R(n * 8 + sh_a + room) + 2 * R(n * 4 + sh_c + room) + 2 * R(n * 8 + sh_b + room)));
    auto take = [&](size_t bytes) { void *p = nullptr; if (one) { p = arena + arena_at; arena_at += bytes; } else CK(hipMalloc(&p, bytes)); return p; };
    a = (u64 *)take(R(n * 8 + room)); ta = (u64 *)take(R(n * 8 + sh_a + room)); c = (u32 *)take(R(n * 4 + room)); tc = (u32 *)take(R(n * 4 + sh_c + room));
    ta = (u64 *)((char *)ta + sh_a); tc = (u32 *)((char *)tc + sh_c);
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    SortScratch scr;
    float ms;
    // copy peak
    for (int withc = 1; withc >= 0 && !only_sorts; withc--) {
        k_fill<<<4096, 256, 0, s>>>(a, c, n, 1);
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) k_copy<<<8192, 256, 0, s>>>(a, withc ? c : nullptr, ta, tc, n);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 2.0 * n * (withc ? 12 : 8);
        printf("copy %2d B records: %.3f ms  %.0f GB/s\n", withc ? 12 : 8, ms / reps, bytes * reps / ms / 1e6);
    }
    for (int blocks = 2048; blocks <= 32768 && !only_sorts; blocks *= 4) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) k_read<<<blocks, 256, 0, s>>>((const ulonglong2 *)a, n / 2, ta);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read-only 8 B records, %5d blocks: %.3f ms  %.0f GB/s\n", blocks, ms / reps, 8.0 * n * reps / ms / 1e6);
    }
    if (!only_sorts) {
        u32 *th; CK(hipMalloc(&th, (n / 4096 + 2) * 1024));
        const unsigned tiles = (unsigned)(n / 4096);
        for (int mode = 0; mode < 4; mode++) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++) {
                if (mode == 0) x_hist<0><<<tiles, 256, 0, s>>>(a, n, 40, th);
                if (mode == 1) x_hist<1><<<tiles, 256, 0, s>>>(a, n, 40, th);
                if (mode == 2) x_hist<2><<<tiles, 256, 0, s>>>(a, n, 40, th);
                if (mode == 3) x_hist<3><<<(tiles + 1) / 2, 256, 0, s>>>(a, n, 40, th);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("x_hist mode %d: %.3f ms  %.0f GB/s\n", mode, ms / reps, 8.0 * n * reps / ms / 1e6);
        }
    }
    u64 *b, *tb;
    b = (u64 *)take(R(n * 8 + room)); tb = (u64 *)take(R(n * 8 + sh_b + room));
    tb = (u64 *)((char *)tb + sh_b);
    struct Case { const char *name; bool hasc, hasb; int lo, hi; } cases[] = {
        {"exact  (u64 hash + u32 pos, bits 16..64)", true, false, 16, 64},
        {"seeds  (u64 code|pos, bits 32..64)", false, false, 32, 64},
        {"seeds with sketches (u64 code|pos + u64 sketch, bits 96..128)", false, true, 96, 128},
    };
    const int only_case = argc > 4 ? atoi(argv[4]) : -1;           // sortbench <n> <reps> sorts <case 0..2>
    for (const Case &cs : cases) {
        if (only_case >= 0 && &cs != &cases[only_case]) continue;
        double tot = 0, sc = 0, scb = 0; u32 launches = 0;
        for (int r = 0; r < reps; r++) {
            k_fill<<<4096, 256, 0, s>>>(a, c, n, 7 + r);
            SortArrays k, t; k.a = a; t.a = ta; if (cs.hasc) { k.c = c; t.c = tc; }
            if (cs.hasb) { k_fill<<<4096, 256, 0, s>>>(b, nullptr, n, 107 + r); k.b = b; t.b = tb; }
            SortTiming tm; tm.enabled = true;
            CK(hipEventRecord(e0, s));
            if (radix_sort(s, k, t, n, cs.lo, cs.hi, scr, &tm)) { fprintf(stderr, "sort failed\n"); return 1; }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms; sc += tm.scatter_ms; scb += (double)tm.scatter_bytes; launches += tm.scatter_launches;
            if (r == 0) {      // check order on a sample
                std::vector<u64> h(1 << 20);
                CK(hipMemcpy(h.data(), k.a, h.size() * 8, hipMemcpyDeviceToHost));
                const int lo = cs.lo & 63, hi = cs.hi - (cs.lo - lo);
                const u64 m = (hi - lo) == 64 ? ~0ull : (((1ull << (hi - lo)) - 1) << lo);
                for (size_t i = 1; i < h.size() && i < n; i++) if ((h[i - 1] & m) > (h[i] & m)) { fprintf(stderr, "NOT SORTED at %zu\n", i); return 1; }
            }
        }
        printf("%s: n=%llu  sort %.3f ms  (%d passes)  scatter avg %.3f ms = %.0f GB/s  whole sort %.0f GB/s algorithmic\n", cs.name,
               (unsigned long long)n, tot / reps, launches / reps, sc / launches, scb / sc / 1e6, scb / tot / 1e6);
    }
    // rocPRIM yardstick
    if (!only_sorts) {
        size_t tb = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb, a, ta, c, tc, n, 16, 64, s));
        void *tmp; CK(hipMalloc(&tmp, tb));
        double tot = 0;
        for (int r = 0; r < reps; r++) {
            k_fill<<<4096, 256, 0, s>>>(a, c, n, 7 + r);
            CK(hipEventRecord(e0, s));
            CK(rocprim::radix_sort_pairs(tmp, tb, a, ta, c, tc, n, 16, 64, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        printf("rocPRIM radix_sort_pairs u64/u32 bits 16..64: %.3f ms\n", tot / reps);
        size_t tb2 = 0;
        CK(rocprim::radix_sort_keys(nullptr, tb2, a, ta, n, 32, 64, s));
        void *tmp2; CK(hipMalloc(&tmp2, tb2));
        tot = 0;
        for (int r = 0; r < reps; r++) {
            k_fill<<<4096, 256, 0, s>>>(a, nullptr, n, 7 + r);
            CK(hipEventRecord(e0, s));
            CK(rocprim::radix_sort_keys(tmp2, tb2, a, ta, n, 32, 64, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        printf("rocPRIM radix_sort_keys u64 bits 32..64: %.3f ms\n", tot / reps);
    }
    return 0;
}
