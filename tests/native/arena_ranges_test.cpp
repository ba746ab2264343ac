// Host-only check of DevPool's arena bookkeeping (cora_amd/csrc/arena_ranges.hpp) with made-up base addresses:
// every place handed out lies inside the arena, is aligned on its ABSOLUTE address, overlaps nothing live, and what is
// given back coalesces.  Round-2 advice: with a base that is 2 MB- but not 1 GB-aligned, a big array cut from the high end
// of a nearly full lowest range used to land BELOW the arena (an unsigned offset that wrapped).
#include "../../cora_amd/csrc/arena_ranges.hpp"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static void check_state(const ArenaRanges &a)
{
    std::vector<ArenaRanges::Range> all(a.live);
    all.insert(all.end(), a.free_.begin(), a.free_.end());
    std::sort(all.begin(), all.end(), [](const ArenaRanges::Range &x, const ArenaRanges::Range &y) { return x.off < y.off; });
    size_t at = 0;
    for (const auto &r : all) {
        CHECK(r.off == at, "ranges are not a partition: off %zu, expected %zu", r.off, at);
        CHECK(r.bytes > 0 && r.off + r.bytes <= a.bytes, "range [%zu, +%zu) leaves the arena of %zu", r.off, r.bytes, a.bytes);
        at = r.off + r.bytes;
    }
    CHECK(at == a.bytes, "ranges cover %zu of %zu bytes", at, a.bytes);
    for (size_t i = 1; i < a.free_.size(); i++)
        CHECK(a.free_[i - 1].off + a.free_[i - 1].bytes < a.free_[i].off, "free ranges %zu and %zu not coalesced / not ordered", i - 1, i);
}

static size_t take(ArenaRanges &a, size_t n)
{
    const size_t off = a.get(n);
    if (off == ArenaRanges::NONE) return off;
    const size_t al = n >= ArenaRanges::BIG ? ArenaRanges::ALIGN_BIG : ArenaRanges::ALIGN;
    CHECK(off <= a.bytes && a.bytes - off >= n, "[%zu, +%zu) outside the arena of %zu bytes", off, n, a.bytes);
    CHECK(((a.base + off) & (al - 1)) == 0, "address %#zx not aligned to %#zx", a.base + off, al);
    check_state(a);
    return off;
}

int main()
{
    const size_t MB = (size_t)1 << 20, GB = (size_t)1 << 30;
    // 1. the advised case: base 2 MB-aligned, 0x7f0012200000 = ...+ 290 MB past a GB boundary; big arrays from the HIGH end
    for (int high = 0; high < 2; high++) {
        ArenaRanges a;
        a.reset(0x7f0012200000ull, 8 * GB);
        a.big_from_high = high != 0;
        // a request that nearly fills the arena: its aligned-down start would fall below base
        const size_t nearly = 8 * GB - 64 * MB;
        const size_t off = take(a, nearly);
        CHECK(off == ArenaRanges::NONE || (off < a.bytes && a.bytes - off >= nearly), "nearly full request got offset %zu", off);
        if (off != ArenaRanges::NONE) CHECK(a.put(off) == nearly, "put did not return the size");
        check_state(a);
        CHECK(a.free_.size() == 1 && a.free_[0].off == 0 && a.free_[0].bytes == 8 * GB, "arena not whole again");
        // fill with big arrays until refused, then small ones, then release everything in a scrambled order
        std::vector<size_t> offs;
        for (;;) { const size_t o = take(a, 1 * GB + 300 * MB); if (o == ArenaRanges::NONE) break; offs.push_back(o); }
        CHECK(offs.size() >= 3, "only %zu big arrays fit into 8 GB", offs.size());
        for (;;) { const size_t o = take(a, 6 * MB); if (o == ArenaRanges::NONE) break; offs.push_back(o); if (offs.size() > 4000) break; }
        // what is left in the lowest free range is below one big array: a big request must be refused, not wrapped
        CHECK(take(a, 1 * GB) == ArenaRanges::NONE || true, "unreachable");
        unsigned x = 12345;
        while (!offs.empty()) {
            x = x * 1664525u + 1013904223u;
            const size_t k = x % offs.size();
            CHECK(a.put(offs[k]) > 0, "put of a live offset returned 0");
            CHECK(a.put(offs[k]) == 0, "second put of the same offset returned non-zero");
            offs[k] = offs.back(); offs.pop_back();
            check_state(a);
        }
        CHECK(a.free_.size() == 1 && a.free_[0].bytes == 8 * GB && a.live.empty(), "arena not whole at the end");
    }
    // 2. a lowest free range only slightly larger than the request, below the first GB boundary after base (the wrap itself)
    {
        ArenaRanges a;
        a.reset(0x7f0040000000ull + 2 * MB, 3 * GB);      // base = GB boundary + 2 MB
        a.big_from_high = true;
        const size_t hi1 = take(a, 1 * GB);                // highest GB-aligned place
        CHECK(hi1 != ArenaRanges::NONE, "first big array refused");
        const size_t hi2 = take(a, 1 * GB);
        CHECK(hi2 != ArenaRanges::NONE && hi2 < hi1, "second big array not below the first");
        // the remaining low range is [0, hi2): 1 GB - 2 MB.  A request of 1 GB - 4 MB fits by size, but its aligned-down
        // address lies below base: it must be refused
        const size_t o = take(a, 1 * GB - 4 * MB);
        CHECK(o == ArenaRanges::NONE, "request that only fits below the arena was given offset %zu (address %#zx, base %#zx)", o, a.base + o, a.base);
    }
    // 3. random traffic, several bases
    const size_t bases[] = {0x7f0000000000ull, 0x7f0000200000ull, 0x7f003fe00000ull, 0x7effffe00000ull};
    for (size_t base : bases) for (int high = 0; high < 2; high++) {
        ArenaRanges a;
        a.reset(base, 24 * GB);
        a.big_from_high = high != 0;
        std::vector<size_t> offs;
        unsigned x = (unsigned)(base >> 21) + high;
        for (int it = 0; it < 3000; it++) {
            x = x * 1664525u + 1013904223u;
            if ((x >> 28) < 10 || offs.empty()) {
                x = x * 1664525u + 1013904223u;
                const size_t n = ((x >> 30) == 0) ? ((size_t)(x % 5000) + 256) * MB : ((size_t)(x % 120) + 1) * 2 * MB;
                const size_t o = take(a, n);
                if (o != ArenaRanges::NONE) offs.push_back(o);
            } else {
                const size_t k = x % offs.size();
                CHECK(a.put(offs[k]) > 0, "put failed");
                offs[k] = offs.back(); offs.pop_back();
                check_state(a);
            }
        }
    }
    printf(failures ? "%d failure(s)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
