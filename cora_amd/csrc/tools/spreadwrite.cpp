// spreadwrite -- does the scatter's WRITE pattern slow down when its 256 output regions are far apart?
// Every workgroup writes 256 runs of `run` bytes, run d of workgroup b at  d * (span / 256) + b * run  (what a radix
// scatter tile does with uniformly distributed digits), for spans from 0.5 GB to 48 GB; reads are a plain stream.
//   spreadwrite [run_bytes=128] [reps=3]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 256 threads; tile = 256 runs x run_words u64 words; thread t of iteration k writes word (k*256+t) of the tile's staged data
__global__ __launch_bounds__(256) void k_spread(const u64 *__restrict__ in, u64 *__restrict__ out, u64 region_words, u32 run_words, u32 xg)
{
    u32 tile = blockIdx.x;
    if (xg > 1) {
        const u32 W = 8 * xg, w = tile / W, r = tile % W;
        if ((w + 1) * W <= gridDim.x) tile = w * W + (r % 8) * xg + r / 8;
    }
    const u64 tile_words = 256ull * run_words;
    const u64 *src = in + (u64)tile * tile_words;
    for (u32 k = 0; k < run_words; k++) {
        const u32 i = k * 256 + threadIdx.x;          // word of the tile
        const u32 d = i / run_words, o = i % run_words;
        const u64 v = __builtin_nontemporal_load(src + i);
        out[(u64)d * region_words + (u64)tile * run_words + o] = v;
    }
}

int main(int argc, char **argv)
{
    const u32 run = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const u32 run_words = run / 8;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gbs[] = {0.5, 24};
    for (double gb : gbs) {
        const u64 bytes = (u64)(gb * (1ull << 30));
        const u64 words = bytes / 8;
        const u64 tile_words = 256ull * run_words;
        const u32 tiles = (u32)(words / tile_words);
        const u64 region_words = (u64)tiles * run_words;
        u64 *in, *out;
        CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
        CK(hipMemsetAsync(in, 1, bytes, s)); CK(hipMemsetAsync(out, 0, bytes, s));
        for (u32 xg : {1u, 16u}) {
            k_spread<<<tiles, 256, 0, s>>>(in, out, region_words, run_words, xg);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++) k_spread<<<tiles, 256, 0, s>>>(in, out, region_words, run_words, xg);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("span %5.1f GB  run %4u B  xcd group %2u: %8.3f ms  %.0f GB/s (read + write)\n", gb, run, xg, ms / reps, 2.0 * bytes * reps / ms / 1e6);
        }
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
