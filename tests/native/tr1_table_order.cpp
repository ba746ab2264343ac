// Test helper (host only): the REAL std::tr1::unordered_map<string, int> of this image's libstdc++, set up as the reference
// sets up its table (/root/reference/collapse.cpp:640-643: max_load_factor(z / 1.13), rehash(n)) but with the rehash
// request given on the command line, fed the keys of stdin (one per line) in order.  Prints
//   G <element number> <bucket count>     for the initial table (element number 0) and whenever the bucket count changed
//                                         with an insertion (element number = that insertion, 1-based over distinct keys)
//   O <key>                               every key in the container's iteration order
// tests/test_collapse_growth.py compares the G lines with cora_hip_collapse_table_growth and the O lines with the order
// the oracle derives from hashes and insertion ranks alone.
#include <tr1/unordered_map>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char **argv)
{
    const unsigned long request = argc > 1 ? strtoul(argv[1], 0, 10) : 200000000ul;
    std::tr1::unordered_map<std::string, int> m;
    float z = m.max_load_factor();
    m.max_load_factor(z / 1.13);
    m.rehash(request);
    std::size_t buckets = m.bucket_count();
    printf("G 0 %zu\n", buckets);
    std::string line;
    while (std::getline(std::cin, line)) {
        const std::size_t before = m.size();
        int &v = m[line];
        if (m.size() != before) v = (int)m.size();
        if (m.bucket_count() != buckets) { buckets = m.bucket_count(); printf("G %zu %zu\n", m.size(), buckets); }
    }
    for (std::tr1::unordered_map<std::string, int>::const_iterator it = m.begin(); it != m.end(); ++it) printf("O %s\n", it->first.c_str());
    return 0;
}
