// arena_ranges.hpp -- the range bookkeeping of DevPool's arena (job_internal.hpp), free of any HIP call so that
// tests/native/arena_ranges_test.cpp can drive it on the host with made-up base addresses.
//
// One allocation of `bytes` bytes at address `base`; arrays are cut from it: big ones (>= BIG) at multiples of ALIGN_BIG
// (1 GB, made on the ABSOLUTE address: hipMalloc only promises 2 MB) from one end, small ones at multiples of ALIGN (2 MB)
// from the other.  All arithmetic is on absolute addresses first and turned into offsets only once a place is known to lie
// inside the free range: rounding an address DOWN can leave the range (or the arena) on the low side, and an unsigned
// offset would wrap there (round-2 advice).
#pragma once
#include <stddef.h>
#include <vector>

struct ArenaRanges {
    struct Range { size_t off, bytes; };
    static constexpr size_t NONE = ~(size_t)0;
    static constexpr size_t BIG = (size_t)256 << 20, ALIGN_BIG = (size_t)1 << 30, ALIGN = (size_t)2 << 20;
    size_t base = 0, bytes = 0;
    std::vector<Range> free_;                            // by offset, coalesced
    std::vector<Range> live;
    bool big_from_high = false;                          // the big arrays come from the high end, the small ones from the low end

    void reset(size_t base_addr, size_t n)
    {
        base = base_addr; bytes = n;
        free_.assign(1, Range{0, n});
        live.clear();
    }
    void clear() { base = bytes = 0; free_.clear(); live.clear(); }

    size_t carve(size_t i, size_t at, size_t n)          // [at, at + n) out of free range i
    {
        const Range r = free_[i];
        free_.erase(free_.begin() + i);
        if (at + n < r.off + r.bytes) free_.insert(free_.begin() + i, Range{at + n, r.off + r.bytes - (at + n)});
        if (at > r.off) free_.insert(free_.begin() + i, Range{r.off, at - r.off});
        live.push_back(Range{at, n});
        return at;
    }
    // offset of a place for n bytes, or NONE
    size_t get(size_t n)
    {
        if (!n || n > bytes) return NONE;
        const bool big = n >= BIG;
        const size_t al = big ? ALIGN_BIG : ALIGN;
        if (big != big_from_high) {                      // lowest place that fits
            for (size_t i = 0; i < free_.size(); i++) {
                const Range r = free_[i];
                const size_t lo = base + r.off, hi = lo + r.bytes;          // absolute
                const size_t abs = (lo + al - 1) & ~(al - 1);
                if (abs < lo || abs > hi || hi - abs < n) continue;          // (abs < lo: the rounding wrapped)
                return carve(i, abs - base, n);
            }
        } else {                                         // highest place that fits
            for (size_t i = free_.size(); i-- > 0;) {
                const Range r = free_[i];
                if (r.bytes < n) continue;
                const size_t lo = base + r.off, hi = lo + r.bytes;
                const size_t abs = (hi - n) & ~(al - 1);                     // rounded DOWN: may fall below the range, even below the arena
                if (abs < lo) continue;
                return carve(i, abs - base, n);
            }
        }
        return NONE;
    }
    // gives [off, ...) back; returns its size, 0 when nothing live starts there
    size_t put(size_t off)
    {
        for (size_t i = 0; i < live.size(); i++) {
            if (live[i].off != off) continue;
            const Range r = live[i];
            live[i] = live.back(); live.pop_back();
            size_t k = 0;
            while (k < free_.size() && free_[k].off < r.off) k++;
            if (k > 0 && free_[k - 1].off + free_[k - 1].bytes == r.off) { k--; free_[k].bytes += r.bytes; }
            else free_.insert(free_.begin() + k, r);
            if (k + 1 < free_.size() && free_[k].off + free_[k].bytes == free_[k + 1].off) {
                free_[k].bytes += free_[k + 1].bytes;
                free_.erase(free_.begin() + k + 1);
            }
            return r.bytes;
        }
        return 0;
    }
};
