/* TEST INFRASTRUCTURE ONLY -- never linked or called by the product (DESIGN.md sections 2 and 8: only tests/, smoke() and benchmarks may build or run it).
 *
 * loader_oracle.c: CPU restatement of how the consumer of the two homology tables loads them
 * (MI = /root/reference/mappingInference.cpp), the row SURVEY.md section 8(f).1 names as the next one:
 *   LoadEquivClasses        MI:654-723    <Exact>   -> equivClassList
 *   LoadInexactHomologies   MI:554-651    <Inexact> -> emHomClassList + maxRevLookup
 *   LoadHomologyTables      MI:1084-1153  position hubs equivHub / homHub
 * Pinned against the real thing: oracle/_ref/ref_loader (built from the reference's own source by build_ref.sh)
 * dumps the same arrays; tests/golden/manifest.json holds the sha256 of its dumps for the golden tables.
 *
 *   RetrieveOffsetAndDirInfo MI:6618-6674  which class covers a position, at which block offset, in which direction
 *
 * usage: loader_oracle [-L] [--exact-only] [--lookup-all <out.bin>] <Exact> <Inexact> <d> <out.dump> <len of contig 1> [...]
 * --lookup-all: RetrieveOffsetAndDirInfo for EVERY position of every contig, 8 bytes each: class index u32 (0 = none),
 * block offset i16, direction u8, 0.
 * The hubs are dense arrays over the concatenated contigs instead of the reference's hash maps; a position the maps
 * would not hold reads 0, which is what every reader of the maps treats "absent" as (MI:1841-1848, 6620-6630).
 * Exit 5: the tables reach outside the contigs (the maps would take any key; the dense hubs cannot).
 */
#include "loader_dump.h"

static uint8_t *slurp(const char *path, uint64_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(88); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc(sz ? sz : 1);
    if (sz && fread(b, 1, sz, f) != (size_t)sz) { perror(path); exit(88); }
    fclose(f);
    *n = (uint64_t)sz;
    return b;
}

static uint64_t item_count(const char *table)   /* MI:559-567, 657-665: "<table>.itemCount" holds one decimal number */
{
    char p[4096];
    snprintf(p, sizeof p, "%s.itemCount", table);
    FILE *f = fopen(p, "r");
    if (!f) { perror(p); exit(87); }
    unsigned long long v = 0;
    if (fscanf(f, "%llu", &v) != 1) v = 0;
    fclose(f);
    return v;
}

static uint32_t rd32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rdchr(const uint8_t *p, int cb) { return cb == 2 ? (uint16_t)(p[0] | p[1] << 8) : p[0]; }

static void need(uint64_t at, uint64_t want, uint64_t n, const char *what)
{
    if (at + want > n) { fprintf(stderr, "loader_oracle: %s ends inside a record (offset %llu)\n", what, (unsigned long long)at); exit(6); }
}

static uint64_t *starts; static uint64_t *lens; static int nc;
static uint64_t hub_index(uint32_t chr, uint32_t pos)
{
    if (chr < 1 || (int)chr > nc || pos < 1 || pos > lens[chr]) {
        fprintf(stderr, "loader_oracle: contig %u position %u is outside the reference\n", chr, pos);
        exit(5);
    }
    return starts[chr] + pos - 1;
}

/* MI:6618-6674 on the flat arrays; the reference's unsigned arithmetic (wrap-around included) and its short offset */
static uint32_t retrieve_offset_and_dir(const loader_dump *D, uint32_t chr, uint32_t pos, int16_t *off, uint8_t *dir)
{
    uint32_t idx = D->equiv_hub[hub_index(chr, pos)];
    if (idx == 0) return 0;
    uint32_t i = idx - 1, bl = D->eq_block[i], rp = D->eq_pos[i];
    if (chr == D->eq_chr[i] && pos >= rp && pos < rp + bl) { *dir = 0; *off = (int16_t)(pos - rp); return idx; }
    for (uint64_t k = D->eq_off[i]; k < D->eq_off[i + 1]; k++) {
        uint32_t hp = D->eqi_pos[k];
        if (chr != D->eqi_chr[k]) continue;
        if (D->eqi_dir[k] == 0) {
            if (pos >= hp && pos < hp + bl) { *dir = 0; *off = (int16_t)(pos - hp); return idx; }
        } else {
            if (pos <= hp && pos > hp - bl) { *dir = 1; *off = (int16_t)(hp - pos); return idx; }
        }
    }
    fprintf(stderr, "loader_oracle: position %u:%u is in the hub of class %u but in none of its ranges (the reference asserts, MI:6673)\n", chr, pos, idx);
    exit(7);
}

int main(int argc, char **argv)
{
    int cb = 1, exact_only = 0, a = 1;
    const char *lookup_out = NULL;
    for (; a < argc && argv[a][0] == '-'; a++) {
        if (!strcmp(argv[a], "-L")) cb = 2;
        else if (!strcmp(argv[a], "--exact-only")) exact_only = 1;   /* the BEST mapping modes, MI:1090-1093, 1139-1152 */
        else if (!strcmp(argv[a], "--lookup-all") && a + 1 < argc) lookup_out = argv[++a];
    }
    if (argc - a < 5) { fprintf(stderr, "usage: loader_oracle [-L] [--exact-only] Exact Inexact d out.dump len1 [len2 ...]\n"); return 2; }
    const char *fx = argv[a], *fi = argv[a + 1], *out = argv[a + 3];
    int d = atoi(argv[a + 2]);
    nc = argc - a - 4;
    starts = (uint64_t *)calloc(nc + 2, 8); lens = (uint64_t *)calloc(nc + 2, 8);
    for (int c = 1; c <= nc; c++) { lens[c] = strtoull(argv[a + 3 + c], 0, 10); starts[c + 1] = starts[c] + lens[c]; }

    loader_dump D;
    memset(&D, 0, sizeof D);
    D.d = d; D.G = starts[nc + 1];
    uint64_t nx = 0, ni = 0;
    uint8_t *bx = slurp(fx, &nx), *bi = exact_only ? NULL : slurp(fi, &ni);
    D.n_eq = item_count(fx);
    D.n_hm = exact_only ? 0 : item_count(fi);

    /* first walk: entry counts (the reference grows a buffer instead, MI:668-719) */
    uint64_t at = 0;
    for (uint64_t i = 0; i < D.n_eq; i++) {
        need(at, cb + 4 + 1, nx, "<Exact>"); at += cb + 4;
        while (bx[at] != 0xFF) { need(at, 1 + cb + 4 + 1, nx, "<Exact>"); at += 1 + cb + 4; D.m_eq++; }
        need(at, 5, nx, "<Exact>"); at += 5;
    }
    const int isz = 1 + cb + 4 + d + 2;
    at = 0;
    for (uint64_t i = 0; i < D.n_hm; i++) {
        need(at, cb + 4 + 1, ni, "<Inexact>"); at += cb + 4;
        while (bi[at] != 0xFF) { need(at, isz + 1, ni, "<Inexact>"); at += isz; D.m_hm++; }
        at += 1;
    }
    loader_dump_alloc(&D);

    /* LoadEquivClasses, MI:680-721: chrCode, chrPos, {dir chrCode chrPos}*, 0xFF, blockLen */
    at = 0; uint64_t k = 0;
    for (uint64_t i = 0; i < D.n_eq; i++) {
        D.eq_chr[i] = rdchr(bx + at, cb); D.eq_pos[i] = rd32(bx + at + cb); at += cb + 4;
        D.eq_off[i] = k;
        while (bx[at] != 0xFF) {
            D.eqi_dir[k] = bx[at]; D.eqi_chr[k] = rdchr(bx + at + 1, cb); D.eqi_pos[k] = rd32(bx + at + 1 + cb);
            at += 1 + cb + 4; k++;
        }
        D.eq_block[i] = rd32(bx + at + 1); at += 5;
    }
    D.eq_off[D.n_eq] = k;

    /* LoadInexactHomologies, MI:581-633: blockLen is a 7-bit field, dir a 1-bit field of emHomItem (MI:475-486) */
    at = 0; k = 0;
    for (uint64_t i = 0; i < D.n_hm; i++) {
        D.hm_chr[i] = rdchr(bi + at, cb); D.hm_pos[i] = rd32(bi + at + cb); at += cb + 4;
        D.hm_off[i] = k;
        while (bi[at] != 0xFF) {
            const uint8_t *p = bi + at;
            D.hmi_dir[k] = p[0] & 1; D.hmi_chr[k] = rdchr(p + 1, cb); D.hmi_pos[k] = rd32(p + 1 + cb);
            for (int e = 0; e < d; e++) D.hmi_offs[k * d + e] = p[1 + cb + 4 + e];
            D.hmi_block[k] = (uint8_t)((p[1 + cb + 4 + d] | p[1 + cb + 4 + d + 1] << 8) & 127);
            at += isz; k++;
        }
        at += 1;
        if (D.hm_off[i] == k) { fprintf(stderr, "loader_oracle: inexact node %llu has no item (the reference reads list[0], MI:643)\n", (unsigned long long)i + 1); exit(6); }
    }
    D.hm_off[D.n_hm] = k;

    /* reverse lookups, MI:638-650, on 1-based indices; slot n_hm+1 of the reference's calloc'd array reads as contig 0 */
    {
        uint64_t n = D.n_hm, look = 2;
        for (uint64_t i = 1; i <= n && look <= n; i++) {
            uint32_t chr = D.hm_chr[i - 1], pos = D.hm_pos[i - 1];
            uint32_t limit = pos + D.hmi_block[D.hm_off[i - 1]] - 1;
            while (look <= n && D.hm_chr[look - 1] == chr && D.hm_pos[look - 1] <= limit) {
                D.hm_maxrev[look - 1] = (uint8_t)(look - i);
                look++;
            }
        }
    }

    /* LoadHomologyTables, MI:1093-1152, in the reference's order: later writes win */
    if (!exact_only)
        for (uint64_t i = 1; i <= D.n_hm; i++) D.hom_hub[hub_index(D.hm_chr[i - 1], D.hm_pos[i - 1])] = (uint32_t)i;
    for (uint64_t i = 1; i <= D.n_eq; i++) {
        uint32_t rc = D.eq_chr[i - 1], rp = D.eq_pos[i - 1], bl = D.eq_block[i - 1];
        for (uint32_t j = 0; j < bl; j++) D.equiv_hub[hub_index(rc, rp + j)] = (uint32_t)i;
        if (exact_only) continue;
        for (uint64_t m = D.eq_off[i - 1]; m < D.eq_off[i]; m++) {
            uint32_t mc = D.eqi_chr[m], mp = D.eqi_pos[m];
            for (uint32_t j = 0; j < bl; j++) {
                uint64_t t = hub_index(mc, D.eqi_dir[m] == 0 ? mp + j : mp - j);
                D.equiv_hub[t] = (uint32_t)i;
                D.hom_hub[t] = D.hom_hub[hub_index(rc, rp + j)];
            }
        }
    }
    if (lookup_out) {
        FILE *f = fopen(lookup_out, "wb");
        if (!f) { perror(lookup_out); return 3; }
        for (int c = 1; c <= nc; c++)
            for (uint32_t p = 1; p <= lens[c]; p++) {
                int16_t off = 0; uint8_t dir = 0;
                uint32_t idx = retrieve_offset_and_dir(&D, (uint32_t)c, p, &off, &dir);
                uint8_t r[8] = {(uint8_t)idx, (uint8_t)(idx >> 8), (uint8_t)(idx >> 16), (uint8_t)(idx >> 24),
                                (uint8_t)off, (uint8_t)((uint16_t)off >> 8), idx ? dir : 0, 0};
                if (!idx) r[4] = r[5] = 0;
                fwrite(r, 1, 8, f);
            }
        fclose(f);
    }
    return loader_dump_write(&D, out) ? 3 : 0;
}
