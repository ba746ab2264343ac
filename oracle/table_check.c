/*
 * table_check.c -- semantic checker of CORA's two homology tables.  TEST INFRASTRUCTURE ONLY
 * (like hts_oracle.c: only tests/ and the profiling scripts build or run it; the product never does).
 *
 * It reads a reference FASTA (+ .fai) and a pair of tables in the on-disk formats F1 / F2 of
 * SURVEY.md section 8a, the way the consumer does (/root/reference/mappingInference.cpp:554-723 loads
 * F2 node by node, item 0 first; :1084-1138 loads F1), and checks what every record CLAIMS against the
 * bases, independently of how the table was computed:
 *
 *   F1  chr pos { dir chr pos }* 0xFF blockSize      (written by HTS:572-635)
 *       for b in [0, blockSize): the K-mer at (chr, pos + b) is N-free and every other entry's K-mer at
 *       pos + b (dir 0) or pos - b (dir 1) is the same K-mer or its reverse complement as dir says --
 *       up to ONE flip of all dir bits per record, which the reference applies or skips depending on
 *       the class's creation order (Q3, HTS:421-433);  records ascend by (chr, pos).
 *   F2  chr pos { dir chr pos off1 [off2] [off3] blockLen }* 0xFF      (written by HTS:1353-1432)
 *       for j in [0, blockLen): source K-mer at pos + j versus target K-mer at tpos + j (dir 0) or the
 *       reverse complement of the one at tpos - j (dir 1) differ in exactly the positions off_i - j
 *       (1-based, in the source's coordinates, HTS:1022-1103), 1..d of them, ascending; items descend by
 *       blockLen (HTS:1368); nodes ascend by (chr, pos).
 *   symmetry: the multiset of unit links {(a, b, dir)} equals that of {(b, a, dir)} (every pair is pushed
 *       in both directions, HTS:1022-1103; block merging only regroups unit links) -- compared through
 *       a commutative 64-bit hash sum.  The reference itself breaks this where its quirks act (palindromic
 *       seeds pushed twice, lists with tied targets merged through a single-candidate search: its own tables
 *       for the pal*, lowcx* and multi* fixtures are not symmetric), so it is reported, and an error only
 *       with --symmetric.
 *   the .itemCount files hold the record counts.
 *
 * It says nothing about completeness (a missing homology goes unnoticed); that is what the byte-exact
 * comparisons with the reference's own tables are for.  The checker itself is pinned by running it over
 * those reference-made tables (tests/test_table_check.py).
 *
 * Usage: table_check <ref.fa> <K> <d> <Exact> <Inexact> [L] [--symmetric]      exit 0 = consistent
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

static int K, D, CW, strict_symmetry;
static int n_ctg;
static u8 **base;          /* per contig: 0..3, 4 = not ACGT */
static u32 *clen;

static void die(const char *fmt, u64 a, u64 b, u64 c)
{
    fprintf(stderr, "table_check: ");
    fprintf(stderr, fmt, a, b, c);
    fprintf(stderr, "\n");
    exit(1);
}

static u8 *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "table_check: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    u8 *p = malloc(sz ? sz : 1);
    if (sz && fread(p, 1, sz, f) != (size_t)sz) { fprintf(stderr, "table_check: short read of %s\n", path); exit(2); }
    fclose(f);
    *n = sz;
    return p;
}

static void load_reference(const char *fa)
{
    char fai[4096];
    snprintf(fai, sizeof fai, "%s.fai", fa);
    FILE *f = fopen(fai, "r");
    if (!f) { fprintf(stderr, "table_check: no %s\n", fai); exit(2); }
    char name[1024];
    u64 len;
    int cap = 16;
    clen = malloc(cap * sizeof *clen);
    while (fscanf(f, "%1023s %llu%*[^\n]", name, &len) == 2) {
        if (n_ctg == cap) { cap *= 2; clen = realloc(clen, cap * sizeof *clen); }
        clen[n_ctg++] = (u32)len;
    }
    fclose(f);
    size_t n;
    u8 *img = slurp(fa, &n);
    base = malloc(n_ctg * sizeof *base);
    size_t at = 0;
    for (int c = 0; c < n_ctg; c++) {
        while (at < n && img[at] != '>') at++;
        while (at < n && img[at] != '\n') at++;
        at++;
        base[c] = malloc(clen[c] + 1);
        u32 got = 0;
        while (at < n && img[at] != '>' && got < clen[c]) {
            u8 ch = img[at++];
            if (ch == '\n' || ch == '\r') continue;
            u8 code = 4;
            switch (ch) { case 'A': case 'a': code = 0; break; case 'C': case 'c': code = 1; break;
                          case 'G': case 'g': code = 2; break; case 'T': case 't': code = 3; break; }
            base[c][got++] = code;
        }
        if (got != clen[c]) die("contig %llu holds %llu bases, the .fai says %llu", c + 1, got, clen[c]);
    }
    free(img);
}

/* K-mer of contig c (0-based) at 1-based position pos into out[K] (oriented: rc = reverse complement);
 * returns 0 if it leaves the contig or holds a non-ACGT base */
static int kmer(int c, long long pos, int rc, u8 *out)
{
    if (c < 0 || c >= n_ctg || pos < 1 || pos + K - 1 > (long long)clen[c]) return 0;
    const u8 *b = base[c] + (pos - 1);
    for (int i = 0; i < K; i++) {
        const u8 x = rc ? b[K - 1 - i] : b[i];
        if (x > 3) return 0;
        out[i] = rc ? (u8)(3 - x) : x;
    }
    return 1;
}

static u64 mix(u64 z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
static u64 link_hash(u32 ca, u32 pa, u32 cb, u32 pb, u32 dir)
{
    return mix(mix(((u64)ca << 32) | pa) + 0x9E3779B97F4A7C15ull * mix(((u64)cb << 32) | pb) + dir);
}

static u32 rd_chr(const u8 **p) { u32 c = (*p)[0]; if (CW == 2) c |= (u32)(*p)[1] << 8; *p += CW; return c; }
static u32 rd_u32(const u8 **p) { u32 v; memcpy(&v, *p, 4); *p += 4; return v; }

static u64 read_count(const char *table)
{
    char path[4096];
    snprintf(path, sizeof path, "%s.itemCount", table);
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "table_check: no %s\n", path); exit(2); }
    u64 v = 0;
    if (fscanf(f, "%llu", &v) != 1) { fprintf(stderr, "table_check: %s holds no count\n", path); exit(2); }
    fclose(f);
    return v;
}

static u64 check_exact(const char *path)
{
    size_t n;
    u8 *buf = slurp(path, &n);
    const u8 *p = buf, *end = buf + n;
    u64 records = 0, entries = 0, prev_key = 0;
    u8 a[64], b[64];
    int ecap = 1024;
    u32 *ec = malloc(ecap * sizeof *ec), *ep = malloc(ecap * sizeof *ep);
    u8 *ed = malloc(ecap);
    while (p < end) {
        int ne = 0;
        for (;;) {
            if (p >= end) die("Exact: record %llu runs past the end of the file", records, 0, 0);
            u8 dir = 0;
            if (ne) { dir = *p++; if (dir == 0xFF) break; if (dir > 1) die("Exact: record %llu entry %llu has dir %llu", records, ne, dir); }
            if (p + CW + 4 > end) die("Exact: record %llu is cut short", records, 0, 0);
            if (ne == ecap) { ecap *= 2; ec = realloc(ec, ecap * sizeof *ec); ep = realloc(ep, ecap * sizeof *ep); ed = realloc(ed, ecap); }
            ed[ne] = dir; ec[ne] = rd_chr(&p); ep[ne] = rd_u32(&p);
            ne++;
        }
        if (p + 4 > end) die("Exact: record %llu has no block size", records, 0, 0);
        const u32 bs = rd_u32(&p);
        if (ne < 2) die("Exact: record %llu has %llu entries (a class needs two)", records, ne, 0);
        if (bs < 1) die("Exact: record %llu has block size 0", records, 0, 0);
        const u64 key = ((u64)ec[0] << 32) | ep[0];
        if (records && key <= prev_key) die("Exact: record %llu does not ascend", records, 0, 0);
        prev_key = key;
        int ok_any = 0;                                 /* one flip of all dir bits per record (Q3) */
        u32 bad_s = 0; int bad_e = 0;
        for (int f = 0; f < 2 && !ok_any; f++) {
            int ok = 1;
            for (u32 s = 0; s < bs && ok; s++) {
                if (!kmer((int)ec[0] - 1, (long long)ep[0] + s, 0, a)) die("Exact: record %llu step %llu: no K-mer at the representative", records, s, 0);
                for (int e = 1; e < ne; e++) {
                    const long long pos = ed[e] ? (long long)ep[e] - s : (long long)ep[e] + s;
                    if (!(kmer((int)ec[e] - 1, pos, ed[e] ^ f, b) && memcmp(a, b, K) == 0)) { ok = 0; bad_s = s; bad_e = e; break; }
                }
            }
            ok_any = ok;
        }
        if (!ok_any) die("Exact: record %llu step %llu entry %llu is not the representative's K-mer", records, bad_s, bad_e);
        entries += ne;
        records++;
    }
    free(buf); free(ec); free(ep); free(ed);
    printf("Exact: %llu records, %llu entries: consistent\n", records, entries);
    return records;
}

static u64 check_inexact(const char *path)
{
    size_t n;
    u8 *buf = slurp(path, &n);
    const u8 *p = buf, *end = buf + n;
    u64 nodes = 0, items = 0, units = 0, prev_key = 0, fwd = 0, bwd = 0;
    u8 a[64], b[64];
    const int isz = 1 + CW + 4 + D + 2;
    while (p < end) {
        if (p + CW + 4 > end) die("Inexact: node %llu is cut short", nodes, 0, 0);
        const u32 sc = rd_chr(&p), sp = rd_u32(&p);
        const u64 key = ((u64)sc << 32) | sp;
        if (nodes && key <= prev_key) die("Inexact: node %llu does not ascend", nodes, 0, 0);
        prev_key = key;
        int ni = 0, prev_len = 32768;
        for (;;) {
            if (p >= end) die("Inexact: node %llu runs past the end of the file", nodes, 0, 0);
            if (*p == 0xFF) { p++; break; }
            if (p + isz > end) die("Inexact: node %llu item %llu is cut short", nodes, ni, 0);
            const u32 dir = *p++;
            if (dir > 1) die("Inexact: node %llu item %llu has dir %llu", nodes, ni, dir);
            const u32 tc = rd_chr(&p), tp = rd_u32(&p);
            int off[3] = {0, 0, 0};
            for (int i = 0; i < D; i++) off[i] = *p++;
            int16_t bl;
            memcpy(&bl, p, 2); p += 2;
            if (bl < 1) die("Inexact: node %llu item %llu has block length %llu", nodes, ni, (u64)(long long)bl);
            if (bl > prev_len) die("Inexact: node %llu item %llu: block lengths do not descend", nodes, ni, 0);
            prev_len = bl;
            int nerr = 0;
            for (int i = 0; i < D; i++) {
                if (off[i]) { if (i != nerr) die("Inexact: node %llu item %llu: offsets have a gap", nodes, ni, 0); nerr++; }
                if (i && off[i] && off[i] <= off[i - 1]) die("Inexact: node %llu item %llu: offsets do not ascend", nodes, ni, 0);
            }
            if (nerr < 1) die("Inexact: node %llu item %llu has no mismatch offset", nodes, ni, 0);
            for (int j = 0; j < bl; j++) {
                const long long spos = (long long)sp + j, tpos = dir ? (long long)tp - j : (long long)tp + j;
                if (!kmer((int)sc - 1, spos, 0, a)) die("Inexact: node %llu item %llu step %llu: no source K-mer", nodes, ni, j);
                if (!kmer((int)tc - 1, tpos, (int)dir, b)) die("Inexact: node %llu item %llu step %llu: no target K-mer", nodes, ni, j);
                int e = 0;
                for (int i = 0; i < K; i++) {
                    if (a[i] == b[i]) continue;
                    if (e >= nerr || off[e] - j != i + 1) die("Inexact: node %llu item %llu step %llu: mismatches are not at the offsets", nodes, ni, j);
                    e++;
                }
                if (e != nerr) die("Inexact: node %llu item %llu step %llu: fewer mismatches than offsets", nodes, ni, j);
                fwd += link_hash(sc, (u32)spos, tc, (u32)tpos, dir);
                bwd += link_hash(tc, (u32)tpos, sc, (u32)spos, dir);
                units++;
            }
            ni++;
        }
        if (!ni) die("Inexact: node %llu has no item", nodes, 0, 0);
        items += ni;
        nodes++;
    }
    free(buf);
    if (fwd != bwd && strict_symmetry) die("Inexact: the unit links are not symmetric (hash sums %llx / %llx over %llu links)", fwd, bwd, units);
    printf("Inexact: %llu nodes, %llu items, %llu unit links: consistent, %s\n", nodes, items, units, fwd == bwd ? "symmetric" : "NOT symmetric");
    return nodes;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: table_check <ref.fa> <K> <d> <Exact> <Inexact> [L]\n"); return 2; }
    K = atoi(argv[2]); D = atoi(argv[3]);
    CW = 1;
    for (int i = 6; i < argc; i++) {
        if (!strcmp(argv[i], "L")) CW = 2;
        else if (!strcmp(argv[i], "--symmetric")) strict_symmetry = 1;
    }
    if (K < 1 || K > 64 || D < 1 || D > 3) { fprintf(stderr, "table_check: bad K or d\n"); return 2; }
    load_reference(argv[1]);
    const u64 ne = check_exact(argv[4]);
    if (read_count(argv[4]) != ne) die("Exact.itemCount says %llu, the table holds %llu records", read_count(argv[4]), ne, 0);
    const u64 nn = check_inexact(argv[5]);
    if (read_count(argv[5]) != nn) die("Inexact.itemCount says %llu, the table holds %llu nodes", read_count(argv[5]), nn, 0);
    return 0;
}
