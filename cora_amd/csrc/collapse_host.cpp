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

// The whole growth history of that table: entry 0 = the bucket count after rehash(request) on the empty table (at[0] = 0);
// entry k >= 1 = the element number (1-based, counted over distinct keys in insertion order) whose insertion makes the
// table re-bucket itself BEFORE that element goes in (tr1/hashtable.h: _M_insert_bucket asks _M_need_rehash first), and the
// bucket count it grows to.  The policy object is libstdc++'s own, driven the way the container drives it (its cached
// _M_next_resize included: the float arithmetic of _M_need_rehash is only reached behind that cache).
// Stops behind the first entry whose element number exceeds max_elements, or at cap entries.  Returns the entry count.
extern "C" int cora_hip_collapse_table_growth(uint64_t request, uint64_t max_elements, uint64_t *at, uint64_t *buckets, int cap)
{
    typedef std::tr1::__detail::_Prime_rehash_policy Policy;
    if (!at || !buckets || cap < 1) return 0;
    const float z = Policy().max_load_factor();
    const float f = z / 1.13;
    Policy pol(f);
    // rehash(n) on the empty table (hashtable.h: max of the two, both touch the cache; the first insertion repairs it)
    const std::size_t b1 = pol._M_next_bkt(request), b2 = pol._M_bkt_for_elements(1);
    std::size_t B = b1 > b2 ? b1 : b2;
    int n_entries = 0;
    at[n_entries] = 0; buckets[n_entries] = B; n_entries++;
    std::size_t n = 0;                                   // elements in the table
    while (n_entries < cap) {
        // smallest n' >= n for which inserting element n' + 1 asks for more buckets
        std::size_t q = n;
        std::pair<bool, std::size_t> r;
        for (;;) {
            r = pol._M_need_rehash(B, q, 1);
            if (r.first) break;
            if (pol._M_next_resize > q) q = pol._M_next_resize;      // nothing is even computed before that element count
            else q++;
        }
        at[n_entries] = q + 1; buckets[n_entries] = r.second; n_entries++;
        B = r.second;
        n = q + 1;                                       // that element is in (and _M_need_rehash is next asked with it counted)
        if (q + 1 > max_elements) break;
    }
    return n_entries;
}
