// collapse_host.cpp -- the geometry of the hash map the reference collapses with (/root/reference/collapse.cpp:640-643:
// std::tr1::unordered_map<string, hashItem*>, max_load_factor(z / 1.13), rehash(200000000)), asked of libstdc++'s own
// rehash policy.  The GPU path needs the bucket count (the reference prints in bucket order, collapse.cpp:2305) and the
// element count at which the table would re-bucket itself.  Host code only; compiled with g++.
#include "../../include/cora_hip.h"

#include <tr1/unordered_map>

extern "C" int cora_hip_collapse_table_geometry(uint64_t *buckets, uint64_t *first_rehash_at)
{
    typedef std::tr1::__detail::_Prime_rehash_policy Policy;
    const float z = Policy().max_load_factor();          // a fresh table's factor, as the reference reads it
    const float f = z / 1.13;                            // the reference's expression: float / double, stored as float
    Policy pol(f);
    const std::size_t B = pol._M_next_bkt(200000000);
    std::size_t lo = 1, hi = 4 * B;
    while (lo < hi) {                                    // smallest element number whose insertion asks for more buckets
        const std::size_t mid = lo + (hi - lo) / 2;
        Policy q(f);
        q._M_next_resize = 0;
        if (q._M_need_rehash(B, mid - 1, 1).first) hi = mid; else lo = mid + 1;
    }
    if (buckets) *buckets = B;
    if (first_rehash_at) *first_rehash_at = lo;
    return 0;
}
