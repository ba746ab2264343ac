// Test helper: the real libstdc++ std::sort / std::partial_sort on (key, index) records, so the
// oracle's C restatement of that algorithm can be pinned against the library the reference links.
#include <algorithm>
#include <cstdint>
struct Rec { int32_t key; int32_t idx; };
extern "C" void ref_std_sort_desc(Rec *v, long n) { std::sort(v, v + n, [](const Rec &a, const Rec &b) { return a.key > b.key; }); }
extern "C" void ref_std_sort_asc(Rec *v, long n) { std::sort(v, v + n, [](const Rec &a, const Rec &b) { return a.key < b.key; }); }
extern "C" void ref_heap_sort_asc(Rec *v, long n) { std::partial_sort(v, v + n, v + n, [](const Rec &a, const Rec &b) { return a.key < b.key; }); }
