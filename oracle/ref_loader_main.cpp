// TEST INFRASTRUCTURE ONLY (DESIGN.md sections 2 and 8): drives the REAL reference's homology-table loaders and
// dumps what they built, so that oracle/loader_oracle.c and the HIP loader can be pinned against them.
//
// The reference source is compiled from where it lies (build_ref.sh passes its path as CORA_REF_SRC); nothing
// of it is copied here.  Its own main() is renamed so that this file's main() can call
//   LoadHomologyTables(perf, inexact, readLen, numHomTableMismatches, mapMode)   mappingInference.cpp:1084-1153
// (which calls LoadEquivClasses :654-723 and LoadInexactHomologies :554-651) and then walk the globals
// equivClassList / emHomClassList / equivHub / homHub (:501-538).
//
// usage: ref_loader <Exact> <Inexact> <K> <d> <out.dump> <len of contig 1> [<len of contig 2> ...]
// Dump = the arrays of loader_dump.h in order, each as u64 byte count + data.
#define main cora_reference_main
#include CORA_REF_SRC
#undef main

#include "loader_dump.h"

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: ref_loader Exact Inexact K d out.dump len1 [len2 ...]\n"); return 2; }
    int K = atoi(argv[3]), d = atoi(argv[4]);
    if (d > MAX_NUM_HOM_EDITS) { fprintf(stderr, "this build holds %d offsets per item, d=%d needs more\n", MAX_NUM_HOM_EDITS, d); return 2; }
    int nc = argc - 6;
    std::vector<unsigned long long> start(nc + 2, 0), len(nc + 2, 0);
    for (int c = 1; c <= nc; c++) { len[c] = strtoull(argv[5 + c], 0, 10); start[c + 1] = start[c] + len[c]; }
    unsigned long long G = start[nc + 1];

    // CORA_REF_EXACT_ONLY=1: the loading of the BEST mapping modes (no inexact table, mappingInference.cpp:1090-1093, 1139-1152)
    const char *eo = getenv("CORA_REF_EXACT_ONLY");
    LoadHomologyTables(argv[1], argv[2], K, d, (eo && eo[0] == '1') ? BEST_MAPPING_MODE : ALL_MAPPING_MODE);

    loader_dump D;
    memset(&D, 0, sizeof D);
    D.d = d;
    D.n_eq = equivClassListSize;
    D.n_hm = emHomClassListSize;
    for (unsigned i = 1; i <= equivClassListSize; i++) D.m_eq += equivClassList[i].listSize;
    for (unsigned i = 1; i <= emHomClassListSize; i++) D.m_hm += emHomClassList[i].listSize;
    D.G = G;
    loader_dump_alloc(&D);
    unsigned long long at = 0;
    for (unsigned i = 1; i <= equivClassListSize; i++) {
        equivClassNode &n = equivClassList[i];
        D.eq_chr[i - 1] = n.chrCode; D.eq_pos[i - 1] = n.chrPos; D.eq_block[i - 1] = n.blockLen; D.eq_off[i - 1] = at;
        for (unsigned k = 0; k < n.listSize; k++, at++) {
            D.eqi_dir[at] = (unsigned char)n.list[k].dir; D.eqi_chr[at] = n.list[k].chrCode; D.eqi_pos[at] = n.list[k].chrPos;
        }
    }
    D.eq_off[equivClassListSize] = at;
    at = 0;
    for (unsigned i = 1; i <= emHomClassListSize; i++) {
        emHomClassNode &n = emHomClassList[i];
        D.hm_chr[i - 1] = n.chrCode; D.hm_pos[i - 1] = n.chrPos; D.hm_maxrev[i - 1] = n.maxRevLookup; D.hm_off[i - 1] = at;
        for (unsigned k = 0; k < n.listSize; k++, at++) {
            emHomItem &it = n.list[k];
            D.hmi_dir[at] = it.dir; D.hmi_chr[at] = it.chrCode; D.hmi_pos[at] = it.chrPos; D.hmi_block[at] = it.blockLen;
            for (int e = 0; e < d; e++) D.hmi_offs[at * d + e] = it.offsetList[e];
        }
    }
    D.hm_off[emHomClassListSize] = at;
    unsigned long long outside = 0;
    for (int c = 0; c < MAX_NUM_CHRS + 2; c++) {
        for (int which = 0; which < 2; which++) {
            std::tr1::unordered_map<unsigned int, unsigned int> &m = which ? homHub[c] : equivHub[c];
            uint32_t *dst = which ? D.hom_hub : D.equiv_hub;
            for (std::tr1::unordered_map<unsigned int, unsigned int>::iterator it = m.begin(); it != m.end(); ++it) {
                if (c < 1 || c > nc || it->first < 1 || it->first > len[c]) { if (it->second) outside++; continue; }
                dst[start[c] + it->first - 1] = it->second;   // absent and 0 are the same to every reader (:1841-1848, :6620-6630)
            }
        }
    }
    if (outside) fprintf(stderr, "ref_loader: %llu hub entries outside the contigs\n", outside);
    // CORA_REF_LOOKUP_ALL=<file>: RetrieveOffsetAndDirInfo (mappingInference.cpp:6618-6674) for every position of every
    // contig, 8 bytes each: class index u32 (0 = none), block offset i16, direction u8, 0
    if (const char *lk = getenv("CORA_REF_LOOKUP_ALL")) {
        FILE *f = fopen(lk, "wb");
        if (!f) { perror(lk); return 3; }
        for (int c = 1; c <= nc; c++)
            for (unsigned long long p = 1; p <= len[c]; p++) {
                short off = 0; bool dir = 0;
                unsigned int idx = RetrieveOffsetAndDirInfo((unsigned int)c, (unsigned int)p, off, dir);
                if (!idx) { off = 0; dir = 0; }
                unsigned char r[8] = {(unsigned char)idx, (unsigned char)(idx >> 8), (unsigned char)(idx >> 16), (unsigned char)(idx >> 24),
                                      (unsigned char)off, (unsigned char)((unsigned short)off >> 8), (unsigned char)dir, 0};
                fwrite(r, 1, 8, f);
            }
        fclose(f);
    }
    int rc = loader_dump_write(&D, argv[5]);
    return rc ? 3 : (outside ? 4 : 0);
}
