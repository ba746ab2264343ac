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

int main(int argc, char **argv)
{
    const u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 64000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const bool only_sorts = argc > 3 && !strcmp(argv[3], "sorts");      // skip the ceilings and the rocPRIM yardstick
    u64 *a, *ta; u32 *c, *tc;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&ta, n * 8)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&tc, n * 4));
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
    CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&tb, n * 8));
    struct Case { const char *name; bool hasc, hasb; int lo, hi; } cases[] = {
        {"exact  (u64 hash + u32 pos, bits 16..64)", true, false, 16, 64},
        {"seeds  (u64 code|pos, bits 32..64)", false, false, 32, 64},
        {"seeds with sketches (u64 code|pos + u64 sketch, bits 96..128)", false, true, 96, 128},
    };
    for (const Case &cs : cases) {
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
