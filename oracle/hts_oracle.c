/*
 * hts_oracle.c -- CPU restatement of CORA's homTable_setup (homology-table
 * construction).  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle of the MI355X build.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, execute or
 * link it; the product (cora_amd/, include/) never does.
 *
 * It restates, in plain C and in the simplest sequential form, what the
 * reference program does, phase by phase, quirks included (Q1-Q4 of SURVEY.md
 * section 8a).  Every function cites the reference lines it follows
 * (HTS = /root/reference/homTable_setup.cpp).  It is *pinned*: tests/ check it
 * byte for byte against tables produced by the real reference binary
 * (oracle/_ref, built by oracle/build_ref.sh) and committed under tests/golden/.
 *
 * Where the reference's output depends on libstdc++'s std::sort tie behaviour
 * (HTS:1141, HTS:1368) the oracle carries its own restatement of that published
 * algorithm (GCC libstdc++ bits/stl_algo.h + bits/stl_heap.h, introsort with a
 * 16-element threshold, depth limit 2*floor(log2 n), heap-sort fallback).
 *
 * Usage (same argv as the reference, HTS:2392-2405):
 *   hts_oracle <ref.fa> <K> <physSplits> <ExactOut> <InexactOut> <parSplits>
 *              <EXACT|BOTH> <d> <FULL> [L]
 * A 10th argument "L" selects the CHR_SHORT (2-byte contig id) layout of
 * homTable_setup_L (HTS:23-27, 210-214, 682-688, 708-712).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <assert.h>
#include <ctype.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define MAX_LINE 250
#define MAX_SEEDS 4

static void die(int code, const char *msg)
{
    fprintf(stderr, "hts_oracle: %s\n", msg);
    exit(code);
}

/* ------------------------------------------------------------------ */
/* Reference genome (HTS:96-202)                                       */
/* ------------------------------------------------------------------ */
static int nchr;
static int *clen;          /* 1-based */
static char **cseq;        /* cseq[c][1..len], '\0' at len+1 */
static char **cmark;       /* representative marks (group id), HTS:86,2577 */
static int K;              /* read-mer length */
static int chr_short;      /* 1 = homTable_setup_L layout */

static int val_of(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; default: return 3; }
}
static char comp_of(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

/* HTS:144-202 (fai parse, line length = length of the file's 2nd token) and
 * HTS:96-142 (line-by-line load at 1-based offsets advancing by the fixed line
 * length; upper-casing; everything outside ACGTN becomes N).  Restated for
 * conforming input; non-conforming input is an error here, not undefined. */
static void load_reference(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) die(50, "cannot open reference");
    char tok1[4096], tok2[4096];
    if (fscanf(f, "%4095s %4095s", tok1, tok2) != 2 || tok1[0] != '>') die(51, "reference not FASTA");
    int linelen = (int)strlen(tok2);
    fclose(f);

    char fai[4096];
    snprintf(fai, sizeof fai, "%s.fai", path);
    FILE *ff = fopen(fai, "r");
    if (!ff) die(52, "cannot open .fai");
    int cap = 16; nchr = 0;
    clen = malloc(sizeof(int) * cap);
    char line[8192];
    while (fgets(line, sizeof line, ff)) {
        char name[4096]; int sz;
        if (sscanf(line, "%4095s %d", name, &sz) != 2) continue;
        nchr++;
        if (nchr + 1 >= cap) { cap *= 2; clen = realloc(clen, sizeof(int) * cap); }
        clen[nchr] = sz;
    }
    fclose(ff);
    cseq = calloc(nchr + 2, sizeof(char *));
    cmark = calloc(nchr + 2, sizeof(char *));
    for (int c = 1; c <= nchr; c++) {
        cseq[c] = calloc((size_t)clen[c] + MAX_LINE + 2, 1);
        cmark[c] = calloc((size_t)clen[c] + MAX_LINE + 2, 1);
    }
    f = fopen(path, "r");
    int cur = 0; long pos = 1;
    char buf[MAX_LINE + 2];
    while (fgets(buf, MAX_LINE, f)) {
        if (buf[0] == '>') {
            if (!strchr(buf, '\n')) die(51, "header line too long for the reference's loader");
            cur++; pos = 1;
            if (cur > nchr) die(51, "more contigs in FASTA than in .fai");
            continue;
        }
        if (cur == 0) die(51, "sequence before first header");
        size_t n = strlen(buf);
        while (n && (buf[n - 1] == '\n')) n--;
        if ((long)(pos + n - 1) > clen[cur]) die(51, "contig longer than .fai says");
        memcpy(cseq[cur] + pos, buf, n);
        pos += linelen;     /* HTS:119: cursor advances by the fixed line length */
    }
    fclose(f);
    for (int c = 1; c <= nchr; c++) {
        char *p = cseq[c];
        for (int i = 1; i <= clen[c]; i++) {
            unsigned char ch = (unsigned char)p[i];
            if (ch == 0) die(51, "contig shorter than .fai says (or ragged lines)");
            if (ch > 'Z') ch -= ('a' - 'A');              /* HTS:131-132 */
            if (!(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N')) ch = 'N'; /* HTS:133-136 */
            p[i] = (char)ch;
        }
        p[clen[c] + 1] = '\0';
    }
}

/* ------------------------------------------------------------------ */
/* Split signals (HTS:2295-2358)                                       */
/* ------------------------------------------------------------------ */
static unsigned char sig2[4][4];        /* group of (first,last) base pair */
static unsigned char sig4[4][4][4][4];  /* thread of the 4-char signal, indexed [c0][c1][c2][c3] of the token */
static int n_groups, n_threads;
static char **group_line;               /* text of each physical group line (for the pre-compact header) */

static void revcomp_str(const char *s, char *out)
{
    int n = (int)strlen(s);
    for (int i = 0; i < n; i++) out[i] = comp_of(s[n - 1 - i]);
    out[n] = 0;
}

/* HTS:2431-2460 + HTS:2499-2509: first line = count, then one group per line;
 * each token and its reverse complement belong to the line's group.  Later
 * lines overwrite earlier ones (HTS:2342, 2508). */
static void load_phys(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) die(66, "cannot open 2-char signal file");
    char line[65536];
    if (!fgets(line, sizeof line, f)) die(66, "empty 2-char signal file");
    n_groups = atoi(line);
    group_line = calloc(n_groups + 64, sizeof(char *));
    int g = 0;
    while (fgets(line, sizeof line, f)) {
        size_t n = strlen(line);
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        g++;
        group_line[g] = strdup(line);
        char *save = NULL; int cnt = 0;
        char tmp[65536]; strcpy(tmp, line);
        for (char *t = strtok_r(tmp, " \t", &save); t; t = strtok_r(NULL, " \t", &save)) {
            char rc[16];
            if (strlen(t) != 2) die(66, "bad 2-char signal");
            sig2[val_of(t[0])][val_of(t[1])] = (unsigned char)g; cnt++;
            revcomp_str(t, rc);
            sig2[val_of(rc[0])][val_of(rc[1])] = (unsigned char)g;
        }
        if (cnt == 0) die(66, "empty 2-char signal group (reference asserts)");
    }
    fclose(f);
    if (g != n_groups) die(66, "2-char signal count mismatch (reference asserts HTS:2656)");
}

/* HTS:2327-2346 */
static void load_par(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) die(67, "cannot open 4-char signal file");
    char line[262144];
    if (!fgets(line, sizeof line, f)) die(67, "empty 4-char signal file");
    n_threads = atoi(line);
    for (int i = 1; i <= n_threads; i++) {
        if (!fgets(line, sizeof line, f)) line[0] = 0;
        char *save = NULL;
        for (char *t = strtok_r(line, " \t\r\n", &save); t; t = strtok_r(NULL, " \t\r\n", &save)) {
            char rc[16];
            if (strlen(t) != 4) die(67, "bad 4-char signal");
            sig4[val_of(t[0])][val_of(t[1])][val_of(t[2])][val_of(t[3])] = (unsigned char)i;
            revcomp_str(t, rc);
            sig4[val_of(rc[0])][val_of(rc[1])][val_of(rc[2])][val_of(rc[3])] = (unsigned char)i;
        }
    }
    fclose(f);
}

/* ------------------------------------------------------------------ */
/* 2-bit k-mer keys (HTS:1757-1789, 1941-1951, 1984-2000)              */
/* ------------------------------------------------------------------ */
typedef struct { u64 w[2]; } key_t2;   /* K<=64: two left-aligned big-endian words */

static key_t2 key_from(const char *s)
{
    key_t2 k = {{0, 0}};
    for (int i = 0; i < K; i++) {
        int w = i >> 5, sh = 62 - 2 * (i & 31);
        k.w[w] |= (u64)val_of(s[i]) << sh;
    }
    return k;
}
static key_t2 key_rc_from(const char *s)
{
    key_t2 k = {{0, 0}};
    for (int i = 0; i < K; i++) {
        int w = i >> 5, sh = 62 - 2 * (i & 31);
        k.w[w] |= (u64)(3 - val_of(s[K - 1 - i])) << sh;
    }
    return k;
}
static int key_cmp(key_t2 a, key_t2 b)
{
    if (a.w[0] != b.w[0]) return a.w[0] < b.w[0] ? -1 : 1;
    if (a.w[1] != b.w[1]) return a.w[1] < b.w[1] ? -1 : 1;
    return 0;
}

/* open-addressing map key -> class id (stands for HTS:2516 unordered_map) */
typedef struct { key_t2 k; long long id; } kslot;
static kslot *ktab; static size_t kcap;
static void ktab_init(size_t n)
{
    size_t c = 1024; while (c < 2 * n + 16) c <<= 1;
    free(ktab); ktab = calloc(c, sizeof(kslot)); kcap = c;
    if (!ktab) die(1, "out of memory");
}
static long long *ktab_get(key_t2 k)
{
    u64 h = (k.w[0] * 0x9E3779B97F4A7C15ULL) ^ (k.w[1] * 0xC2B2AE3D27D4EB4FULL);
    h ^= h >> 29;
    size_t i = (size_t)h & (kcap - 1);
    for (;;) {
        if (ktab[i].id == 0) { ktab[i].k = k; return &ktab[i].id; }
        if (ktab[i].k.w[0] == k.w[0] && ktab[i].k.w[1] == k.w[1]) return &ktab[i].id;
        i = (i + 1) & (kcap - 1);
    }
}

/* HTS:1954-1981.  Returns -1 as soon as the scan index reaches refLen, so a
 * window that ends on the contig's last base is never *found* here (Q4). */
static int first_non_n(const char *ref, int pos, int len)
{
    int start = pos, cur = pos;
    for (;;) {
        if (ref[cur] == 'N') { cur++; start = cur; }
        else { cur++; if (cur - start >= K) return start; }
        if (cur >= len) return -1;
    }
}

/* one line of the pre-compact text (HTS:2580-2616) */
typedef struct { long long cls; int chr; u32 pos; char dir; } eline;
static eline *elines; static size_t n_elines, cap_elines;
static void push_eline(long long cls, int chr, u32 pos, int dir)
{
    if (n_elines == cap_elines) {
        cap_elines = cap_elines ? cap_elines * 2 : (1 << 20);
        elines = realloc(elines, cap_elines * sizeof(eline));
        if (!elines) die(1, "out of memory");
    }
    elines[n_elines].cls = cls; elines[n_elines].chr = chr; elines[n_elines].pos = pos; elines[n_elines].dir = (char)dir;
    n_elines++;
}

static long long n_classes;

/* HTS:2499-2640: exact construct.  pre != NULL also writes the text intermediate. */
static void exact_construct(FILE *pre)
{
    long long classIdx = 0;
    for (int g = 1; g <= n_groups; g++) {
        if (pre) fprintf(pre, "signals %s\n", group_line[g]);
        /* size the map: windows of this group */
        size_t cnt = 0;
        for (int c = 1; c <= nchr; c++) cnt += (size_t)clen[c];
        ktab_init(cnt / (n_groups > 1 ? 2 : 1) + 1024);
        size_t used = 0;
        for (int c = 1; c <= nchr; c++) {
            if (pre) fprintf(pre, "chrCode %d\n", c);
            long long prevCls = 0, prevPos = 0;
            const char *ref = cseq[c]; char *marks = cmark[c];
            int start = first_non_n(ref, 1, clen[c]);
            if (start < 0) die(54, "contig without a findable first K-mer: reference behaviour undefined (HTS:2530)");
            for (int cur = start;; cur++) {
                if (sig2[val_of(ref[cur])][val_of(ref[cur + K - 1])] == g) {
                    key_t2 fw = key_from(ref + cur), rc = key_rc_from(ref + cur);
                    int cmp = key_cmp(fw, rc);      /* HTS:2555: <0 fw smaller, 0 palindrome */
                    int sw = 0; long long *slot;
                    if (cmp <= 0) slot = ktab_get(fw);
                    else { sw = 1; slot = ktab_get(rc); }
                    if (marks[cur] != 0) die(57, "marks corruption");
                    if (*slot == 0) {
                        if (++used * 2 > kcap) die(1, "oracle hash table too small");
                        classIdx++; *slot = classIdx; marks[cur] = (char)g;      /* HTS:2573-2578 */
                    }
                    long long cls = *slot;
                    for (int rep = 0; rep < (cmp == 0 ? 2 : 1); rep++) {         /* Q2: palindrome printed twice, HTS:2597-2616 */
                        int dir = rep ? 1 : sw;
                        if (pre) {
                            if (cls - prevCls != 1) fprintf(pre, "E%lld\t", cls - prevCls);
                            fprintf(pre, "%d", dir);
                            if (cur - prevPos != 1) fprintf(pre, "\t%lld", cur - prevPos);
                            fprintf(pre, "\n");
                        }
                        prevCls = cls; prevPos = cur;
                        push_eline(cls, c, (u32)cur, dir);
                    }
                }
                char nc = ref[cur + K];
                if (nc == '\0') break;                                           /* HTS:2619-2623 */
                if (nc == 'N') {
                    cur = first_non_n(ref, cur + K + 1, clen[c]) - 1;            /* HTS:2625-2634 */
                    if (cur < 0) break;
                }
            }
        }
    }
    n_classes = classIdx;
}

/* ------------------------------------------------------------------ */
/* libstdc++ std::sort restated (bits/stl_algo.h, bits/stl_heap.h)     */
/* generic over an element size through a comparator on pointers       */
/* ------------------------------------------------------------------ */
typedef int (*less_fn)(const void *, const void *);
typedef struct { char *base; size_t es; less_fn less; char *tmp; char *tmp2; } sorter;

#define EL(s, i) ((s)->base + (size_t)(i) * (s)->es)
static void el_swap(sorter *s, long a, long b)
{
    if (a == b) return;
    memcpy(s->tmp2, EL(s, a), s->es); memcpy(EL(s, a), EL(s, b), s->es); memcpy(EL(s, b), s->tmp2, s->es);
}
static void ss_push_heap(sorter *s, long first, long hole, long top, const char *value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && s->less(EL(s, first + parent), value)) {
        memcpy(EL(s, first + hole), EL(s, first + parent), s->es);
        hole = parent; parent = (hole - 1) / 2;
    }
    memcpy(EL(s, first + hole), value, s->es);
}
static void ss_adjust_heap(sorter *s, long first, long hole, long len, const char *value_in)
{
    char *value = malloc(s->es); memcpy(value, value_in, s->es);
    long top = hole, second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (s->less(EL(s, first + second), EL(s, first + second - 1))) second--;
        memcpy(EL(s, first + hole), EL(s, first + second), s->es);
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        memcpy(EL(s, first + hole), EL(s, first + second - 1), s->es);
        hole = second - 1;
    }
    ss_push_heap(s, first, hole, top, value);
    free(value);
}
static void ss_heap_sort(sorter *s, long first, long last)
{
    long len = last - first;
    if (len >= 2) {                                   /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) {
            memcpy(s->tmp, EL(s, first + parent), s->es);
            ss_adjust_heap(s, first, parent, len, s->tmp);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {                        /* __sort_heap / __pop_heap */
        --last;
        memcpy(s->tmp, EL(s, last), s->es);
        memcpy(EL(s, last), EL(s, first), s->es);
        ss_adjust_heap(s, first, 0, last - first, s->tmp);
    }
}
static void ss_median_to_first(sorter *s, long result, long a, long b, long c)
{
    if (s->less(EL(s, a), EL(s, b))) {
        if (s->less(EL(s, b), EL(s, c))) el_swap(s, result, b);
        else if (s->less(EL(s, a), EL(s, c))) el_swap(s, result, c);
        else el_swap(s, result, a);
    } else if (s->less(EL(s, a), EL(s, c))) el_swap(s, result, a);
    else if (s->less(EL(s, b), EL(s, c))) el_swap(s, result, c);
    else el_swap(s, result, b);
}
static long ss_unguarded_partition(sorter *s, long first, long last, long pivot)
{
    for (;;) {
        while (s->less(EL(s, first), EL(s, pivot))) ++first;
        --last;
        while (s->less(EL(s, pivot), EL(s, last))) --last;
        if (!(first < last)) return first;
        el_swap(s, first, last);
        ++first;
    }
}
static void ss_introsort_loop(sorter *s, long first, long last, long depth)
{
    while (last - first > 16) {
        if (depth == 0) { ss_heap_sort(s, first, last); return; }
        --depth;
        long mid = first + (last - first) / 2;
        ss_median_to_first(s, first, first + 1, mid, last - 1);
        long cut = ss_unguarded_partition(s, first + 1, last, first);
        ss_introsort_loop(s, cut, last, depth);
        last = cut;
    }
}
static void ss_unguarded_linear_insert(sorter *s, long last)
{
    memcpy(s->tmp, EL(s, last), s->es);
    long next = last - 1;
    while (s->less(s->tmp, EL(s, next))) {
        memcpy(EL(s, last), EL(s, next), s->es);
        last = next; --next;
    }
    memcpy(EL(s, last), s->tmp, s->es);
}
static void ss_insertion_sort(sorter *s, long first, long last)
{
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (s->less(EL(s, i), EL(s, first))) {
            memcpy(s->tmp, EL(s, i), s->es);
            memmove(EL(s, first + 1), EL(s, first), (size_t)(i - first) * s->es);
            memcpy(EL(s, first), s->tmp, s->es);
        } else ss_unguarded_linear_insert(s, i);
    }
}
/* public so that tests can pin the restatement against the real std::sort */
void oracle_std_sort(void *base, long n, size_t es, less_fn less, long forced_depth)
{
    if (n <= 0) return;
    sorter s; s.base = base; s.es = es; s.less = less; s.tmp = malloc(es); s.tmp2 = malloc(es);
    long lg = 0; { unsigned long v = (unsigned long)n; while (v >>= 1) lg++; }
    ss_introsort_loop(&s, 0, n, forced_depth >= 0 ? forced_depth : lg * 2);
    if (n > 16) {
        ss_insertion_sort(&s, 0, 16);
        for (long i = 16; i != n; ++i) ss_unguarded_linear_insert(&s, i);
    } else ss_insertion_sort(&s, 0, n);
    free(s.tmp); free(s.tmp2);
}

/* ------------------------------------------------------------------ */
/* Exact compaction (HTS:249-452, 455-497, 572-635, 650-670)           */
/* ------------------------------------------------------------------ */
typedef struct { char dir; int chr; u32 pos; } eitem;
typedef struct { eitem *list; int count; } eclass;

static int less_class_by_rep(const void *a, const void *b)
{
    const eclass *x = a, *y = b;
    if (x->list[0].chr != y->list[0].chr) return x->list[0].chr < y->list[0].chr;
    return x->list[0].pos < y->list[0].pos;
}
static void put_chr(FILE *f, int c)
{
    if (chr_short) { short v = (short)c; fwrite(&v, 2, 1, f); }
    else { char v = (char)c; fwrite(&v, 1, 1, f); }
}

static void exact_compact(const char *out)
{
    eclass *set = calloc((size_t)n_classes + 1, sizeof(eclass));
    for (size_t i = 0; i < n_elines; i++) set[elines[i].cls - 1].count++;
    for (long long i = 0; i < n_classes; i++) {
        if (set[i].count > 1) { set[i].list = malloc(sizeof(eitem) * set[i].count); set[i].count = 0; }
        else set[i].count = -1;                                   /* HTS:336-347 */
    }
    for (size_t i = 0; i < n_elines; i++) {
        eclass *e = &set[elines[i].cls - 1];
        if (e->count != -1) {
            e->list[e->count].dir = elines[i].dir; e->list[e->count].chr = elines[i].chr; e->list[e->count].pos = elines[i].pos;
            e->count++;
        }
    }
    /* HTS:415-437: compaction; the direction flip sits inside if(i!=finalI) (Q3) */
    long long fin = 0;
    for (long long i = 0; i < n_classes; i++) {
        if (set[i].count > 1) {
            if (i != fin) {
                set[fin] = set[i];
                if (set[fin].list[0].dir == 1)
                    for (int j = 0; j < set[fin].count; j++) set[fin].list[j].dir = !set[fin].list[j].dir;
            }
            fin++;
        }
    }
    oracle_std_sort(set, (long)fin, sizeof(eclass), less_class_by_rep, -1);   /* HTS:450 */

    FILE *f = fopen(out, "wb");
    if (!f) die(68, "cannot write exact table");
    long long printed = 0;
    for (long long i = 0; i < fin; i++) {
        printed++;
        /* HTS:455-497 */
        unsigned int dist = 1;
        eitem *base = set[i].list; int bn = set[i].count;
        while (i + dist < fin && base[0].chr == set[i + dist].list[0].chr && base[0].pos + dist == set[i + dist].list[0].pos) {
            eitem *cmp = set[i + dist].list;
            int ok = (set[i + dist].count == bn);
            for (int k = 1; ok && k < bn; k++) {
                if (base[k].dir != cmp[k].dir || base[k].chr != cmp[k].chr) ok = 0;
                else if (base[k].dir == 0) { if (base[k].pos + dist != cmp[k].pos) ok = 0; }
                else { if (base[k].pos - dist != cmp[k].pos) ok = 0; }
            }
            if (!ok) break;
            dist++;
        }
        /* HTS:599-618 */
        put_chr(f, base[0].chr); fwrite(&base[0].pos, 4, 1, f);
        for (int k = 1; k < bn; k++) { fwrite(&base[k].dir, 1, 1, f); put_chr(f, base[k].chr); fwrite(&base[k].pos, 4, 1, f); }
        char term = -1; fwrite(&term, 1, 1, f); fwrite(&dist, 4, 1, f);
        i += dist - 1;
    }
    fclose(f);
    char cnt[4200]; snprintf(cnt, sizeof cnt, "%s.itemCount", out);
    FILE *fc = fopen(cnt, "w"); fprintf(fc, "%lld\n", printed); fclose(fc);
    for (long long i = 0; i < fin; i++) free(set[i].list);
    free(set);
}

/* ------------------------------------------------------------------ */
/* Seeds (HTS:2015-2116, 1793-1838, 1904-1911)                          */
/* ------------------------------------------------------------------ */
static char seg_tmpl[MAX_LINE + 2];
static int sub_seg_len;

static void init_templates(int d)
{
    int n = d + 1;
    memset(seg_tmpl, 0, sizeof seg_tmpl);
    if (K >= n * 16) {
        for (int i = 0; i < n * 8; i++) { seg_tmpl[i] = (char)(1 + i % n); seg_tmpl[K - i - 1] = (char)(1 + i % n); }
        sub_seg_len = 8;
    } else {
        int half = K / 2, empty = (K % (2 * n)) / 2;
        for (int i = 0; i < half - empty; i++) { seg_tmpl[i] = (char)(1 + i % n); seg_tmpl[K - i - 1] = (char)(1 + i % n); }
        sub_seg_len = K / (n * 2);
    }
}
/* HTS:2069-2116 */
static void seed_offsets(int s, int *so, int *eo, int *ts1, int *te1, int *ts2, int *te2)
{
    int cnt = 0;
    *so = *eo = *ts1 = *te1 = *ts2 = *te2 = -1;
    for (int i = 0; i < K; i++) if (seg_tmpl[i] == s) {
        if (cnt == 0) { *so = i; cnt++; } else if (cnt == 1) { *ts1 = i; cnt++; } else { *ts2 = i; break; }
    }
    cnt = 0;
    for (int i = K - 1; i > *ts2; i--) if (seg_tmpl[i] == s) {
        if (cnt == 0) { *eo = i; cnt++; } else if (cnt == 1) { *te1 = i; cnt++; } else { *te2 = i; break; }
    }
}
/* HTS:1793-1838.  NB: the reference tests readLen >= 16*NUM_MAX_SEEDS (=64), not
 * >= 16*(d+1): for K<64 the second branch runs even when seeds are full length;
 * with sub_seg_len==8 it adds no padding and visits the same positions. */
static u32 seed_code(const char *s, int seed)
{
    u32 v = 0;
    if (K >= 16 * MAX_SEEDS) {
        for (int i = 0; i < K; i++) if (seg_tmpl[i] == seed) v = (v << 2) | (u32)val_of(s[i]);
    } else {
        for (int i = 0; i < K / 2; i++) if (seg_tmpl[i] == seed) v = (v << 2) | (u32)val_of(s[i]);
        for (int k = 0; k < 8 - sub_seg_len; k++) { v = (v << 2) | 0; v = (v << 2) | 3; }
        for (int i = K / 2; i < K; i++) if (seg_tmpl[i] == seed) v = (v << 2) | (u32)val_of(s[i]);
    }
    return v;
}
static u32 rc_code(u32 in)
{
    u32 out = 0;
    for (int i = 0; i < 16; i++) { out = (out << 2) | (3 - (in & 3)); in >>= 2; }
    return out;
}

/* ------------------------------------------------------------------ */
/* Inexact construct (HTS:2671-2800, 2138-2265)                        */
/* ------------------------------------------------------------------ */
typedef struct { int c1, c2; u32 p1, p2; char dir, e1, e2, e3; } hpair;
typedef struct { hpair *v; size_t n, cap; } pairvec;
static pairvec *tfile;      /* one per thread file, 1..n_threads */

static void push_pair(pairvec *pv, hpair h)
{
    if (pv->n == pv->cap) { pv->cap = pv->cap ? pv->cap * 2 : 4096; pv->v = realloc(pv->v, pv->cap * sizeof(hpair)); if (!pv->v) die(1, "out of memory"); }
    pv->v[pv->n++] = h;
}

typedef struct { int chr; int pos; char dir; int next; } mseg;   /* HTS:2002-2012 */
static mseg *segs; static size_t n_segs, cap_segs;
static int new_seg(int chr, int pos, int dir, int next)
{
    if (n_segs == cap_segs) { cap_segs = cap_segs ? cap_segs * 2 : (1 << 16); segs = realloc(segs, cap_segs * sizeof(mseg)); if (!segs) die(1, "out of memory"); }
    segs[n_segs].chr = chr; segs[n_segs].pos = pos; segs[n_segs].dir = (char)dir; segs[n_segs].next = next;
    return (int)n_segs++;
}
/* code -> chain head */
typedef struct { u32 code; int head; char used; } cslot;
static cslot *ctab; static size_t ccap;
static int *ctab_get(u32 code)
{
    size_t i = ((size_t)code * 2654435761u) & (ccap - 1);
    for (;;) {
        if (!ctab[i].used) { ctab[i].used = 1; ctab[i].code = code; ctab[i].head = -1; return &ctab[i].head; }
        if (ctab[i].code == code) return &ctab[i].head;
        i = (i + 1) & (ccap - 1);
    }
}

/* HTS:2138-2265 */
static void compare_and_report(const mseg *s1, const mseg *s2, pairvec *out, int seed, int d, int palFlag)
{
    const char *r1 = cseq[s1->chr] + s1->pos, *r2 = cseq[s2->chr] + s2->pos;
    int nerr = 0, epos[3] = {0, 0, 0};
    unsigned char segerr[MAX_SEEDS + 1] = {0, 0, 0, 0, 0};
    if (s1->dir == s2->dir) {
        for (int i = 0; i < K; i++) if (seg_tmpl[i] != seed && r1[i] != r2[i]) {
            segerr[(int)seg_tmpl[i]]++; if (nerr < 3) epos[nerr] = i + 1; nerr++; if (nerr > d) return;
        }
    } else {
        for (int i = 0; i < K; i++) if (seg_tmpl[i] != seed && r1[i] != comp_of(r2[K - i - 1])) {
            segerr[(int)seg_tmpl[i]]++; if (nerr < 3) epos[nerr] = i + 1; nerr++; if (nerr > d) return;
        }
    }
    if (nerr == 0) {
        if (palFlag && s1->chr == s2->chr && s1->pos == s2->pos) return;        /* HTS:2192-2195 */
        die(70, "zero-mismatch pair among representatives (reference asserts HTS:2200)");
    }
    switch (seed) {                                                              /* HTS:2222-2232 */
        case 4: if (segerr[3] == 0) return; /* fallthrough */
        case 3: if (segerr[2] == 0) return; /* fallthrough */
        case 2: if (segerr[1] == 0) return; /* fallthrough */
        default: break;
    }
    hpair h; h.c1 = s1->chr; h.c2 = s2->chr; h.p1 = (u32)s1->pos; h.p2 = (u32)s2->pos;
    h.dir = (char)!(s1->dir == s2->dir);
    h.e1 = (char)epos[0]; h.e2 = (char)(nerr >= 2 ? epos[1] : 0); h.e3 = (char)(nerr >= 3 ? epos[2] : 0);
    push_pair(out, h);
}

static void inexact_construct(int d)
{
    int nseeds = d + 1;
    tfile = calloc(n_threads + 2, sizeof(pairvec));
    size_t nreps = 0;
    for (int c = 1; c <= nchr; c++) for (int p = 1; p <= clen[c] - K + 1; p++) if (cmark[c][p]) nreps++;
    ccap = 1024; while (ccap < 2 * nreps + 16) ccap <<= 1;
    ctab = malloc(ccap * sizeof(cslot));
    for (int g = 1; g <= n_groups; g++)
        for (int t = 1; t <= n_threads; t++)          /* HTS:2676: one OpenMP thread each; files are per thread */
            for (int s = 1; s <= nseeds; s++) {
                memset(ctab, 0, ccap * sizeof(cslot)); n_segs = 0;
                int so, eo, ts1, te1, ts2, te2;
                seed_offsets(s, &so, &eo, &ts1, &te1, &ts2, &te2);
                for (int c = 1; c <= nchr; c++) {
                    const char *ref = cseq[c]; const char *marks = cmark[c];
                    int lim = clen[c] - K + 1;
                    for (int p = 1; p <= lim; p++) {
                        if (!marks[p]) continue;
                        if (sig2[val_of(ref[p + so])][val_of(ref[p + eo])] != g) continue;                 /* HTS:2721 */
                        /* HTS:2722 + 2355-2358: token = (S1, S2, E2, E1) */
                        if (sig4[val_of(ref[p + ts1])][val_of(ref[p + ts2])][val_of(ref[p + te2])][val_of(ref[p + te1])] != t) continue;
                        u32 fw = seed_code(ref + p, s), rc = rc_code(fw);
                        int sw = 0, pal = 0; int *head;
                        if (fw <= rc) { head = ctab_get(fw); if (fw == rc) pal = 1; }
                        else { head = ctab_get(rc); sw = 1; }
                        int me = new_seg(c, p, sw, -1);
                        for (int o = *head; o != -1; o = segs[o].next) { mseg a = segs[me], b = segs[o]; compare_and_report(&a, &b, &tfile[t], s, d, 0); }
                        segs[me].next = *head; *head = me;
                        if (pal) {                                                                         /* Q1, HTS:2769-2792 */
                            int pm = new_seg(c, p, 1, -1);
                            head = ctab_get(fw);    /* segs may have been reallocated, head pointer is into ctab: still valid */
                            for (int o = *head; o != -1; o = segs[o].next) { mseg a = segs[pm], b = segs[o]; compare_and_report(&a, &b, &tfile[t], s, d, 1); }
                            segs[pm].dir = (char)sw;
                            segs[pm].next = *head; *head = pm;
                        }
                    }
                }
            }
    free(ctab);
}

/* ------------------------------------------------------------------ */
/* Inexact compaction (HTS:727-765, 1022-1103, 1132-1270, 1353-1638)   */
/* ------------------------------------------------------------------ */
typedef struct { char dir; int chr; int pos; char o1, o2, o3; short blen; } hitem;  /* HTS:705-718 */
typedef struct { int sc; u32 sp; u64 seq; hitem it; } ditem;

static int cmp_ditem(const void *a, const void *b)
{
    const ditem *x = a, *y = b;
    if (x->sc != y->sc) return x->sc < y->sc ? -1 : 1;
    if (x->sp != y->sp) return x->sp < y->sp ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}
static int less_by_target(const void *a, const void *b)     /* HTS:1105-1108 */
{
    const hitem *x = a, *y = b;
    return x->chr == y->chr ? (x->pos < y->pos) : (x->chr < y->chr);
}
static int less_by_blen_desc(const void *a, const void *b)  /* HTS:1110-1113 */
{
    const hitem *x = a, *y = b;
    return x->blen > y->blen;
}

/* HTS:1159-1191 */
static int search_in_vec(const hitem *v, int n, int last, int chr, int dir, int pos, int o1, int o2, int o3)
{
    int cur = last;
    while (cur < n && (v[cur].chr < chr || (v[cur].chr == chr && v[cur].pos < pos))) cur++;
    if (cur >= n) cur = n - 1;
    while (cur >= 0 && (v[cur].chr > chr || (v[cur].chr == chr && v[cur].pos > pos))) cur--;
    if (cur < 0) cur = 0;
    if (v[cur].chr != chr || v[cur].pos != pos || v[cur].dir != dir || v[cur].o1 != o1 || v[cur].o2 != o2 || v[cur].o3 != o3) return -1;
    return cur;
}

/* which contigs are compacted in 50 Mbp windows (d == 3 only).
 * HTS:727-765 (split runs), HTS:1467-1474 (only the run's LAST contig length is
 * tested, '=' not '+='), HTS:1479,1528-1581, HTS:2845-2860. */
static int *windowed;
static void plan_windows(int d)
{
    windowed = calloc(nchr + 2, sizeof(int));
    int maxContig = (d == 3) ? 50000000 : 0;
    if (!maxContig) return;
    int nruns = (nchr > 250 ? 250 : nchr) + 2;
    int *split = calloc(nruns + 2, sizeof(int));
    long long total = 0; for (int i = 1; i <= nchr; i++) total += clen[i];
    long long aimed = total / nruns, sofar = 0; int sn = 1;
    for (int i = 1; i <= nchr; i++) {
        if (i >= nchr) { nruns = sn; break; }
        sofar += clen[i];
        if (sofar >= aimed) { split[sn] = i; sofar = 0; sn++; }
    }
    split[nruns] = nchr;
    for (int r = 0; r < nruns; r++) {
        int a = split[r] + 1, b = split[r + 1];
        if (b < a) continue;
        if (maxContig < clen[b]) for (int c = a; c <= b; c++) if (clen[c] > maxContig) windowed[c] = maxContig;
    }
    free(split);
}

static void inexact_compact(const char *out, int d)
{
    size_t npairs = 0;
    for (int t = 1; t <= n_threads; t++) npairs += tfile[t].n;
    ditem *items = malloc((2 * npairs + 1) * sizeof(ditem));
    size_t ni = 0; u64 seq = 0;
    for (int t = 1; t <= n_threads; t++)                                        /* HTS:1591-1619 */
        for (size_t i = 0; i < tfile[t].n; i++) {
            hpair h = tfile[t].v[i];
            ditem a; a.sc = h.c1; a.sp = h.p1; a.seq = seq++;                   /* HTS:1029-1046 */
            a.it.chr = h.c2; a.it.pos = (int)h.p2; a.it.o1 = h.e1; a.it.o2 = h.e2; a.it.o3 = h.e3; a.it.dir = h.dir; a.it.blen = 1;
            items[ni++] = a;
            ditem b; b.sc = h.c2; b.sp = h.p2; b.seq = seq++;                   /* HTS:1054-1096 */
            b.it.chr = h.c1; b.it.pos = (int)h.p1; b.it.dir = h.dir; b.it.blen = 1;
            if (h.dir) {
                if (h.e2 == 0) { b.it.o1 = (char)(K - h.e1 + 1); b.it.o2 = 0; b.it.o3 = 0; }
                else if (h.e3 == 0) { b.it.o1 = (char)(K - h.e2 + 1); b.it.o2 = (char)(K - h.e1 + 1); b.it.o3 = 0; }
                else { b.it.o1 = (char)(K - h.e3 + 1); b.it.o2 = (char)(K - h.e2 + 1); b.it.o3 = (char)(K - h.e1 + 1); }
            } else { b.it.o1 = h.e1; b.it.o2 = h.e2; b.it.o3 = h.e3; }
            items[ni++] = b;
        }
    qsort(items, ni, sizeof(ditem), cmp_ditem);   /* lists in push order: keys are unique, so qsort is deterministic */

    /* lists */
    size_t nl = 0;
    for (size_t i = 0; i < ni; i++) if (i == 0 || items[i].sc != items[i - 1].sc || items[i].sp != items[i - 1].sp) nl++;
    size_t *lstart = malloc((nl + 1) * sizeof(size_t));
    hitem *hv = malloc((ni + 1) * sizeof(hitem));
    nl = 0;
    for (size_t i = 0; i < ni; i++) {
        if (i == 0 || items[i].sc != items[i - 1].sc || items[i].sp != items[i - 1].sp) lstart[nl++] = i;
        hv[i] = items[i].it;
    }
    lstart[nl] = ni;
    for (size_t l = 0; l < nl; l++)                                              /* HTS:1132-1145 */
        oracle_std_sort(hv + lstart[l], (long)(lstart[l + 1] - lstart[l]), sizeof(hitem), less_by_target, -1);

    plan_windows(d);
    FILE *f = fopen(out, "wb");
    if (!f) die(68, "cannot write inexact table");
    unsigned int printed = 0;
    for (size_t l = 0; l < nl; l++) {
        int sc = items[lstart[l]].sc; u32 sp = items[lstart[l]].sp;
        hitem *base = hv + lstart[l]; int bn = (int)(lstart[l + 1] - lstart[l]);
        /* HTS:1203-1270 */
        for (int dist = 1;; dist++) {
            size_t m = l + dist;
            if (m >= nl) break;
            if (items[lstart[m]].sc != sc || items[lstart[m]].sp != sp + (u32)dist) break;   /* next position has no list */
            if (windowed[sc] && (sp - 1) / (u32)windowed[sc] != (sp + dist - 1) / (u32)windowed[sc]) break; /* other 50 Mbp pass */
            hitem *cmp = hv + lstart[m]; int cn = (int)(lstart[m + 1] - lstart[m]);
            int last = 0;
            for (int k = 0; k < bn; k++) {
                if (base[k].blen != dist) continue;
                int want = base[k].dir == 0 ? base[k].pos + dist : base[k].pos - dist;
                if (want <= 0 || want > clen[base[k].chr]) continue;
                int o1 = base[k].o1 - dist, o2 = 0, o3 = 0;
                if (base[k].o2 != 0) o2 = base[k].o2 - dist;
                if (base[k].o3 != 0) o3 = base[k].o3 - dist;
                if (o1 <= 0 || o2 < 0 || o3 < 0) continue;
                int at = search_in_vec(cmp, cn, last, base[k].chr, base[k].dir, want, o1, o2, o3);
                if (at != -1) { last = at; base[k].blen++; cmp[at].blen = -1; }
            }
        }
        oracle_std_sort(base, bn, sizeof(hitem), less_by_blen_desc, -1);        /* HTS:1368 */
        int header = 0;
        for (int p = 0; p < bn; p++) {
            if (base[p].blen == -1) continue;
            if (!header) { put_chr(f, sc); int kk = (int)sp; fwrite(&kk, 4, 1, f); header = 1; }
            fwrite(&base[p].dir, 1, 1, f); put_chr(f, base[p].chr); fwrite(&base[p].pos, 4, 1, f);
            fwrite(&base[p].o1, 1, 1, f);
            if (d > 1) fwrite(&base[p].o2, 1, 1, f);
            if (d > 2) fwrite(&base[p].o3, 1, 1, f);
            fwrite(&base[p].blen, 2, 1, f);
        }
        if (header) { printed++; char e = -1; fwrite(&e, 1, 1, f); }
    }
    fclose(f);
    char cnt[4200]; snprintf(cnt, sizeof cnt, "%s.itemCount", out);
    FILE *fc = fopen(cnt, "w"); fprintf(fc, "%u\n", printed); fclose(fc);
    free(items); free(lstart); free(hv);
}

#ifndef HTS_ORACLE_NO_MAIN
int main(int argc, char **argv)
{
    if (argc != 10 && argc != 11) {
        fprintf(stderr, "usage: hts_oracle ref K physSplits ExactOut InexactOut parSplits EXACT|BOTH d FULL [L]\n");
        return 19;
    }
    K = atoi(argv[2]);
    if (K < 33 || K > 64) return 65;                                          /* HTS:2418-2424 */
    chr_short = (argc == 11 && argv[10][0] == 'L');
    int d = atoi(argv[8]);
    if (strcmp(argv[9], "FULL") != 0) die(19, "the oracle restates run mode FULL only");
    load_phys(argv[3]);
    load_par(argv[6]);
    load_reference(argv[1]);
    if (!chr_short && nchr > 127) die(19, "more than 127 contigs need the L layout (cora.cpp:25,488-493)");

    char pre[4200]; snprintf(pre, sizeof pre, "%s_preCompact_toDelete", argv[4]);
    FILE *fp = fopen(pre, "w");
    if (!fp) die(68, "cannot write pre-compact file");
    exact_construct(fp);
    fclose(fp);
    if (strcmp(argv[7], "EXACT") == 0) return 0;                               /* HTS:2644-2648 */
    if (d < 1 || d > 3) die(6, "mismatches must be 1..3");                     /* HTS:2017-2021 */
    init_templates(d);
    inexact_construct(d);
    for (int t = 1; t <= n_threads; t++) {                                     /* raw pair files, HTS:2660-2668, 2803-2810 */
        char nm[4300]; snprintf(nm, sizeof nm, "%s_preCompact_toDelete_t%d", argv[5], t);
        FILE *ft = fopen(nm, "wb");
        if (!ft) continue;
        for (size_t i = 0; i < tfile[t].n; i++) {
            hpair h = tfile[t].v[i];
            if (chr_short) { short a = (short)h.c1, b = (short)h.c2; fwrite(&a, 2, 1, ft); fwrite(&b, 2, 1, ft); }
            else { char a = (char)h.c1, b = (char)h.c2, z[2] = {0, 0}; fwrite(&a, 1, 1, ft); fwrite(&b, 1, 1, ft); fwrite(z, 2, 1, ft); }
            fwrite(&h.p1, 4, 1, ft); fwrite(&h.p2, 4, 1, ft);
            fwrite(&h.dir, 1, 1, ft); fwrite(&h.e1, 1, 1, ft); fwrite(&h.e2, 1, 1, ft); fwrite(&h.e3, 1, 1, ft);
        }
        fclose(ft);
    }
    exact_compact(argv[4]);
    inexact_compact(argv[5], d);
    return 0;
}
#endif
