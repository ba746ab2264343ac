/*
 * collapse_oracle.c -- TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 *
 * Plain-C sequential restatement of CORA's `collapse` (/root/reference/collapse.cpp, CL below): its whole-input run mode
 * (argv[13] = NONE, the form `cora search` issues when reads are not fragmented, cora.cpp:1674-1692) and its fragmented
 * mode (argv[13] = the signals of one group, argv[1] = that group's fragment file written by fastqSplit, cora.cpp:1733-1760):
 * every read-mer of the reference and of the reads is reduced to the lexicographically smaller of itself and its
 * reverse complement (CL:97-122, 125-246), identical read-mers are collapsed (CL:640-643 hash map), every read-mer of
 * the input gets an encoded id (CL:276-289: base-94 digits of 2 * running index + direction), and three text files
 * come out (CL:2295-2405): the unique read-mers that are not in the reference (FASTA / faux FASTQ for the coarse
 * mapper), the perfect matches, and the list of names too long for the mapper.
 *
 * Output ORDER is the iteration order of the reference's std::tr1::unordered_map<string, hashItem*> (CL:2305).  With
 * libstdc++'s tr1 hashtable that is: bucket index ascending, and inside a bucket the latest insertion first (nodes are
 * pushed at the bucket's head).  Hash = FNV-1a 64 over the key's bytes (tr1::hash<string>), bucket = hash mod B with
 * B = 206062531 = the table's bucket count after max_load_factor(1/1.13) and rehash(200000000) (CL:641-643;
 * measured on the real container, see tests/test_collapse_oracle.py::test_hashtable_constants).
 *
 * The table GROWS: with its 182356234th element it re-buckets itself (then at 370310930, 751990753, ... elements; the
 * history is libstdc++'s rehash policy's and comes from cora_hip_collapse_table_growth, pinned against the real container
 * by tests/test_collapse_growth.py; CORA_COLLAPSE_GROWTH="at:buckets,..." replaces the built-in history of
 * rehash(200000000) so that tests reach the growth steps with small inputs).  tr1's _M_rehash walks the old buckets in
 * ascending order, each from head to tail, and pushes every node at the HEAD of its new bucket; a new element is pushed
 * at the head of its bucket.  So, with S(k) the iteration order while the table has its k-th bucket count B(k):
 *     S(k) = stable sort by (hash mod B(k)) of [ the elements inserted under B(k), latest first ] ++ reverse(S(k-1))
 * (order_nodes below).  The model -- not the reference at that size: a run with 1.8e8 distinct read-mers does not fit this
 * container -- is pinned against the real std::tr1::unordered_map (tests/native/tr1_table_order.cpp) for thousands of
 * random keys and small initial tables; beyond the first growth step the REFERENCE's own output order is "parity
 * unpinned".
 *
 * Pinned: tests/golden/collapse_cases/ holds the three output files the REAL reference (oracle/_ref/collapse[_L])
 * wrote for every mode combination (tests/golden/make_collapse_golden.py); tests/test_collapse_oracle.py compares.
 *
 * Bytes outside A C G T N: the reference's complement table is zero for them (CL:86-94), so a reference window that holds
 * one (IUPAC letters; the loader only upper-cases, CL:388-400) enters the table under a key that is either cut at the
 * NUL or still holds the letter (canonical() below follows CL:97-122 byte for byte).  No read-mer over A C G T N can equal
 * such a key, so these windows never reach an output file, but they are elements of the table: they count towards its
 * growth and take their place in the insertion order.
 * A READ with such a byte is refused with status 78 (the reference would cut its key).
 *
 *   collapse_oracle <the reference's 16 arguments> [L]         L = collapse_L (-DCHR_SHORT: contig ids up to 65530)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* growth history of the table for rehash(200000000): element number that triggers the step, bucket count from then on */
static uint64_t g_at[32] = {0, 182356234ULL, 370310930ULL, 751990753ULL, 1527068289ULL, 3101019521ULL, 7601712386ULL};
static uint64_t g_buckets[32] = {206062531ULL, 418451333ULL, 849749479ULL, 1725587117ULL, 3504151727ULL, 8589934583ULL, 25769803693ULL};
static int n_growth = 7;            /* (the last entry lies beyond the 2^32 elements this restatement can hold) */
#define MAXLINE 300                       /* CL:27 */

typedef struct {
    uint64_t hash;          /* FNV-1a 64 of the canonical read-mer */
    uint64_t key;           /* offset of the key in the pool */
    uint16_t klen;          /* its length: R, or less for the cut key of a window with a byte outside A C G T N */
    uint32_t ref_code;      /* 0 = not in the reference */
    uint32_t ref_pos;
    int8_t ref_dir;         /* -1 = created by a read (CL:953) */
    uint32_t count;         /* identCount */
    uint32_t id_head, id_tail;
} node_t;

static node_t *nodes; static size_t n_nodes, cap_nodes;
static char *pool; static size_t pool_len, pool_cap;
static uint32_t *slots; static size_t n_slots;              /* open addressing, value = node index + 1 */
static uint64_t *id_val; static uint32_t *id_next; static size_t n_ids, cap_ids;   /* identity lists, 1-based links */
static int R;                                                /* read-mer length */

static uint64_t fnv1a(const char *s, int n)
{
    uint64_t h = 14695981039346656037ULL;
    for (int i = 0; i < n; i++) { h ^= (uint64_t)(unsigned char)s[i]; h *= 1099511628211ULL; }
    return h;
}

static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N'; } return 0; }

/* CL:97-122: 1 iff the reverse complement is lexicographically smaller; out = the key the table is asked for, *len its
 * length (R unless the reverse complement was taken and holds a NUL: std::string(char*) stops there) */
static int canonical(const char *s, char *out, int *len)
{
    *len = R;
    for (int i = 0; i < R; i++) {
        char c = comp(s[R - 1 - i]);
        if (s[i] < c) { memcpy(out, s, R); return 0; }
        if (s[i] > c) {
            for (int k = 0; k < R; k++) out[k] = comp(s[R - 1 - k]);
            for (int k = 0; k < R; k++) if (!out[k]) { *len = k; break; }
            return 1;
        }
    }
    memcpy(out, s, R);
    return 0;
}

static void grow_slots(void)
{
    size_t m = n_slots ? n_slots * 2 : (1u << 16);
    uint32_t *t = (uint32_t *)calloc(m, 4);
    for (size_t i = 0; i < n_nodes; i++) {
        size_t p = (nodes[i].hash * 0x9E3779B97F4A7C15ULL) >> 20 & (m - 1);
        while (t[p]) p = (p + 1) & (m - 1);
        t[p] = (uint32_t)i + 1;
    }
    free(slots); slots = t; n_slots = m;
}

/* find or insert; *created says which */
static node_t *lookup(const char *key, int klen, int *created)
{
    if ((n_nodes + 1) * 2 > n_slots) grow_slots();
    const uint64_t h = fnv1a(key, klen);
    size_t p = (h * 0x9E3779B97F4A7C15ULL) >> 20 & (n_slots - 1);
    while (slots[p]) {
        node_t *nd = &nodes[slots[p] - 1];
        if (nd->hash == h && nd->klen == klen && memcmp(pool + nd->key, key, (size_t)klen) == 0) { *created = 0; return nd; }
        p = (p + 1) & (n_slots - 1);
    }
    if (n_nodes + 1 >= 0xFFFFFFFFull) { fprintf(stderr, "more than 2^32 distinct read-mers\n"); exit(77); }
    if (n_nodes == cap_nodes) { cap_nodes = cap_nodes ? cap_nodes * 2 : 1024; nodes = (node_t *)realloc(nodes, cap_nodes * sizeof(node_t)); }
    if (pool_len + R > pool_cap) { pool_cap = pool_cap ? pool_cap * 2 : (1u << 20); pool = (char *)realloc(pool, pool_cap); }
    node_t *nd = &nodes[n_nodes];
    memset(nd, 0, sizeof *nd);
    nd->hash = h; nd->key = (uint64_t)pool_len; nd->klen = (uint16_t)klen;
    memcpy(pool + pool_len, key, (size_t)klen); pool_len += (size_t)klen;
    slots[p] = (uint32_t)n_nodes + 1;
    n_nodes++;
    *created = 1;
    return nd;
}

static void bad_byte(const char *what, int c) { fprintf(stderr, "%s holds byte 0x%02x: outside A C G T N\n", what, c & 0xFF); exit(78); }

static uint64_t item_index;        /* running index of read-mers over the whole input (CL:276-289, 884-889) */

/* fragmented run mode (argv[13] != NONE): the signals of this run's group (CL:405-488) */
static char frag_sig[1000][4]; static int n_frag_sig, frag_len;

static void setup_fragmentation(const char *text, int len)                  /* CL:414-462 */
{
    if (text[0] == 'N') text += strlen(text) >= 2 ? 2 : strlen(text);       /* nFLAG is set and never read */
    const size_t n = strlen(text);
    if (n % (size_t)(len + 1)) { fprintf(stderr, "fragmentation signals: not a whole number of signals\n"); exit(19); }   /* the reference asserts */
    n_frag_sig = (int)(n / (size_t)(len + 1));
    if (n_frag_sig > 1000) { fprintf(stderr, "more than 1000 fragmentation signals (CL:410)\n"); exit(19); }
    for (int i = 0; i < n_frag_sig; i++) memcpy(frag_sig[i], text + (size_t)(len + 1) * i, (size_t)len);
    frag_len = len;
}

static int frag_included(const char *w)                                       /* CL:464-488 on a window's first / last bases */
{
    if (!frag_len) return 1;
    char sig[4];
    if (frag_len == 2) { sig[0] = w[0]; sig[1] = w[R - 1]; }
    else { sig[0] = w[0]; sig[1] = w[1]; sig[2] = w[R - 2]; sig[3] = w[R - 1]; }
    for (int i = 0; i < n_frag_sig; i++) if (!memcmp(frag_sig[i], sig, (size_t)frag_len)) return 1;
    return 0;
}

static void add_readmer_coded(const char *s, uint64_t code);

static void add_readmer(const char *s)
{
    add_readmer_coded(s, 2 * item_index);
}

static void add_readmer_coded(const char *s, uint64_t code)
{
    char key[MAXLINE + 1];
    for (int i = 0; i < R; i++) if (!comp(s[i])) bad_byte("a read", s[i]);
    int klen;
    const int sw = canonical(s, key, &klen);
    int created;
    node_t *nd = lookup(key, klen, &created);
    int flag = sw;
    if (created) { nd->ref_code = 0; nd->ref_dir = -1; }
    else if (nd->ref_dir != -1) flag = sw != nd->ref_dir;                    /* CL:963-968 */
    if (n_ids == cap_ids) { cap_ids = cap_ids ? cap_ids * 2 : 1024; id_val = (uint64_t *)realloc(id_val, cap_ids * 8); id_next = (uint32_t *)realloc(id_next, cap_ids * 4); }
    id_val[n_ids] = code + (uint64_t)flag; id_next[n_ids] = 0;
    if (nd->count == 0) nd->id_head = (uint32_t)n_ids + 1; else id_next[nd->id_tail - 1] = (uint32_t)n_ids + 1;
    nd->id_tail = (uint32_t)n_ids + 1;
    n_ids++; nd->count++;
    item_index++;
}

/* The table's iteration order over the nodes that are printed (see the header): generation of a node = number of growth
 * steps that came before its insertion (node i is element i + 1). */
typedef struct { uint64_t b; uint32_t pos, node; } ord_t;
static int cmp_ord(const void *a, const void *b)
{
    const ord_t *x = (const ord_t *)a, *y = (const ord_t *)b;
    if (x->b != y->b) return x->b < y->b ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);                  /* stable */
}
static int generation_of(uint64_t element)
{
    int g = 0;
    while (g + 1 < n_growth && g_at[g + 1] <= element) g++;
    return g;
}
static size_t order_nodes(uint32_t *order)
{
    size_t m = 0, total = 0;
    for (size_t i = 0; i < n_nodes; i++) total += nodes[i].count != 0;
    ord_t *w = (ord_t *)malloc((total + 1) * sizeof(ord_t));
    uint32_t *cur = (uint32_t *)malloc((total + 1) * 4);
    const int g_last = n_nodes ? generation_of((uint64_t)n_nodes) : 0;
    if (n_nodes && g_last + 1 >= n_growth && g_at[n_growth - 1] <= (uint64_t)n_nodes) { fprintf(stderr, "the growth history ends before element %zu\n", n_nodes); exit(77); }
    for (int k = 0; k <= g_last; k++) {
        size_t n = 0;
        const uint64_t first = k ? g_at[k] : 1, last = (k + 1 < n_growth && g_at[k + 1] - 1 < n_nodes) ? g_at[k + 1] - 1 : n_nodes;   /* elements first .. last */
        for (uint64_t e = last; e >= first && e >= 1; e--) if (nodes[e - 1].count) { w[n].node = (uint32_t)(e - 1); n++; }   /* latest first */
        for (size_t j = m; j-- > 0;) { w[n].node = cur[j]; n++; }                                                          /* reverse(S(k-1)) */
        for (size_t j = 0; j < n; j++) { w[j].pos = (uint32_t)j; w[j].b = nodes[w[j].node].hash % g_buckets[k]; }
        qsort(w, n, sizeof(ord_t), cmp_ord);
        for (size_t j = 0; j < n; j++) cur[j] = w[j].node;
        m = n;
    }
    memcpy(order, cur, m * 4);
    free(w); free(cur);
    return m;
}

static char *read_file(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc(sz + 2);
    if (sz && fread(b, 1, sz, f) != (size_t)sz) { fclose(f); free(b); return NULL; }
    fclose(f);
    b[sz] = 0; *n = sz;
    return b;
}

/* next line of a buffer as fgets(.., 300, ..) would cut it: returns its start, sets *len (without '\n'); NULL at the end */
static char *next_line(char *buf, size_t n, size_t *at, size_t *len)
{
    if (*at >= n) return NULL;
    char *s = buf + *at;
    size_t l = 0;
    while (*at + l < n && s[l] != '\n' && l < MAXLINE - 1) l++;
    *len = l;
    *at += l + ((*at + l < n && s[l] == '\n') ? 1 : 0);
    return s;
}

int main(int argc, char **argv)
{
    if (argc != 17 && !(argc == 18 && strcmp(argv[17], "L") == 0)) { fprintf(stderr, "usage: collapse_oracle <16 arguments of collapse> [L]\n"); return 10; }
    const int chr_short = argc == 18;
    if (getenv("CORA_COLLAPSE_GROWTH")) {                                    /* "at:buckets,at:buckets,..." (tests: a small table) */
        const char *e = getenv("CORA_COLLAPSE_GROWTH");
        n_growth = 0;
        while (*e && n_growth < 32) {
            char *end;
            g_at[n_growth] = strtoull(e, &end, 10);
            if (*end != ':') break;
            g_buckets[n_growth] = strtoull(end + 1, &end, 10);
            n_growth++;
            e = *end == ',' ? end + 1 : end;
        }
        if (n_growth < 1 || g_at[0] != 0) { fprintf(stderr, "bad CORA_COLLAPSE_GROWTH\n"); return 19; }
    }
    const unsigned invalid_code = chr_short ? 65531u : 251u;                 /* CL:62-66 */
    uint64_t lrn_count = (uint64_t)atoi(argv[16]);
    const int fragmented = strcmp(argv[13], "NONE") != 0;
    const int input_len = atoi(argv[6]);
    int splits;
    if (!strcmp(argv[11], "FULL")) splits = 1; else if (!strcmp(argv[11], "HALF")) splits = 2; else if (!strcmp(argv[11], "THREEWAY")) splits = 3;
    else { printf("ERROR: Split mode unrecognized\n"); return 9; }
    R = input_len / splits;
    if (R < 1 || input_len >= MAXLINE) { fprintf(stderr, "read length out of range\n"); return 19; }
    enum { GENERIC, BWA, BWA_MEM, BOWTIE_2 } mapper = GENERIC;
    if (!strcmp(argv[12], "BWA")) mapper = BWA; else if (!strcmp(argv[12], "BWA_MEM")) mapper = BWA_MEM; else if (!strcmp(argv[12], "BOWTIE_2")) mapper = BOWTIE_2;
    int paired;
    if (!strcmp(argv[10], "PAIRED")) paired = 1; else if (!strcmp(argv[10], "SINGLE")) paired = 0;
    else { printf("ERROR: Input mode unrecognized\n"); return 8; }
    const int D = atoi(argv[2]);
    if (D < 1 || D > 10) { fprintf(stderr, "digit length out of range\n"); return 19; }
    const int ref_line = atoi(argv[5]);
    if (fragmented) {                                                        /* CL:542-563 */
        const int fl = atoi(argv[14]);
        if (fl != 2 && fl != 4) { printf("fragSignalLen can only be 2 or 4\n"); return 6; }
        if (R < fl) { fprintf(stderr, "read-mers shorter than the signal\n"); return 19; }
        setup_fragmentation(argv[13], fl);
    }

    /* ---- reference read-mers (CL:647-711; loader CL:337-402) ---- */
    if (strcmp(argv[4], "NOREF") != 0) {
        char fai_path[4096];
        snprintf(fai_path, sizeof fai_path, "%s.fai", argv[4]);
        size_t fn, rn;
        char *fai = read_file(fai_path, &fn), *ref = read_file(argv[4], &rn);
        if (!fai || !ref) { fprintf(stderr, "cannot read the reference or its .fai\n"); return 50; }
        size_t n_ctg = 0;
        for (size_t i = 0; i < fn; i++) n_ctg += fai[i] == '\n';
        if (fn && fai[fn - 1] != '\n') n_ctg++;
        if (n_ctg >= invalid_code) { fprintf(stderr, "too many contigs for this contig-id width\n"); return 19; }
        uint32_t *len = (uint32_t *)calloc(n_ctg + 2, 4);
        char **seq = (char **)calloc(n_ctg + 2, sizeof(char *));
        { size_t at = 0, c = 0; while (at < fn && c < n_ctg) { char name[1024]; long long sz = 0; sscanf(fai + at, "%1023s %lld", name, &sz); len[++c] = (uint32_t)sz; while (at < fn && fai[at] != '\n') at++; at++; } }
        for (size_t c = 1; c <= n_ctg; c++) seq[c] = (char *)calloc((size_t)len[c] + 2 * MAXLINE + (size_t)ref_line, 1);
        size_t at = 0, l, cur = 0, pos = 1;
        char *s;
        char *sink = (char *)calloc(2 * MAXLINE + rn + 16, 1);           /* text ahead of the first header goes nowhere */
        while ((s = next_line(ref, rn, &at, &l)) != NULL) {
            if (l && s[0] == '>') { cur++; pos = 1; if (cur > n_ctg) break; continue; }
            char *dst = cur ? seq[cur] : sink;
            if (cur && pos + l > (size_t)len[cur] + MAXLINE + (size_t)ref_line) break;
            memcpy(dst + pos, s, l);
            pos += ref_line;
        }
        free(sink);
        char key[MAXLINE + 1];
        for (size_t c = 1; c <= n_ctg; c++) {
            char *q = seq[c];
            for (uint32_t i = 1; i <= len[c]; i++) if (q[i] > 'Z') q[i] -= 32;
            if (len[c] < (uint32_t)R) continue;
            const uint32_t lim = len[c] - R + 1;
            int n_count = 0;
            for (int i = 1; i <= R; i++) n_count += q[i] == 'N';
            for (uint32_t i = 1; i <= lim; i++) {
                if (n_count == 0 && frag_included(q + i)) {                  /* CL:712-835: only this group's windows */
                    int klen;
                    const int sw = canonical(q + i, key, &klen);            /* (a window with other letters: see the header) */
                    int created;
                    node_t *nd = lookup(key, klen, &created);
                    if (created) { nd->ref_code = (uint32_t)c; nd->ref_pos = i; nd->ref_dir = (int8_t)sw; }
                }
                n_count -= q[i] == 'N';
                if (i + R <= len[c]) n_count += q[i + R] == 'N';
            }
            free(q);
        }
        free(seq); free(len); free(fai); free(ref);
    }

    /* ---- fragmented mode (CL:2173-2284): argv[1] is a fragment file of fastqSplit: records of 8 bytes (number of the
     *      read-mer x 2) + R characters; whole records only (fread counts items) ---- */
    if (fragmented) {
        size_t n;
        char *b = read_file(argv[1], &n);
        if (!b) { printf("Binary input file: %s doesn't exist\n", argv[1]); return 5; }
        const size_t item = 8 + (size_t)R;
        for (size_t at = 0; at + item <= n; at += item) {
            uint64_t code;
            memcpy(&code, b + at, 8);
            add_readmer_coded(b + at + 8, code);
        }
        free(b);
    } else
    /* ---- reads (CL:861-2171): list file = sample count, then per sample: file [mate file] read count ---- */
    {
        size_t ln;
        char *list = read_file(argv[1], &ln);
        if (!list) { fprintf(stderr, "cannot read the file list\n"); return 5; }
        char *tok = strtok(list, " \t\r\n");
        const int n_samples = tok ? atoi(tok) : 0;
        if (n_samples > 65535) return 28;
        const int full = R * splits;
        while ((tok = strtok(NULL, " \t\r\n")) != NULL) {
            char *f1 = tok, *f2 = NULL;
            if (paired) { f2 = strtok(NULL, " \t\r\n"); }
            strtok(NULL, " \t\r\n");                                         /* read count, unused (CL:894) */
            size_t n1 = 0, n2 = 0, a1 = 0, a2 = 0, l1, l2;
            char *b1 = read_file(f1, &n1), *b2 = NULL;
            if (!b1) { printf("Input fastq file: %s doesn't exist\n", f1); return 5; }
            if (paired) { b2 = f2 ? read_file(f2, &n2) : NULL; if (!b2) { printf("Read mate input file: %s doesn't exist\n", f2 ? f2 : ""); return 6; } }
            char *s1, *s2 = NULL;
            while (next_line(b1, n1, &a1, &l1) != NULL) {                    /* header */
                if (paired && !next_line(b2, n2, &a2, &l2)) { fprintf(stderr, "mate file ends early\n"); return 7; }
                s1 = next_line(b1, n1, &a1, &l1);
                if (!s1 || l1 < (size_t)full) { fprintf(stderr, "a read is shorter than %d\n", full); return 7; }
                if (paired) { s2 = next_line(b2, n2, &a2, &l2); if (!s2 || l2 < (size_t)full) { fprintf(stderr, "a mate is shorter than %d\n", full); return 7; } }
                size_t junk;
                if (!next_line(b1, n1, &a1, &junk) || !next_line(b1, n1, &a1, &junk)) { fprintf(stderr, "truncated FASTQ record\n"); return 7; }
                if (paired && (!next_line(b2, n2, &a2, &junk) || !next_line(b2, n2, &a2, &junk))) { fprintf(stderr, "truncated FASTQ record\n"); return 7; }
                for (int k = 0; k < splits; k++) add_readmer(s1 + k * R);    /* CL:937-1380 */
                if (paired) for (int k = 0; k < splits; k++) add_readmer(s2 + k * R);   /* CL:1382-1742 */
            }
            free(b1); free(b2);
        }
        free(list);
    }
    if (!fragmented) { uint64_t lim = 1; for (int i = 0; i < D; i++) lim *= 94; if (item_index && 2 * item_index - 1 >= lim) { fprintf(stderr, "digit length %d cannot number %llu read-mers\n", D, (unsigned long long)item_index); return 19; } }

    /* ---- output in the hash map's iteration order (CL:2295-2405) ---- */
    uint32_t *order = (uint32_t *)malloc((n_nodes + 1) * 4);
    const size_t m = order_nodes(order);
    FILE *fr = fopen(argv[3], "w"), *fp = fopen(argv[7], "w"), *fl = fopen(argv[9], "w");
    if (!fr || !fp || !fl) { fprintf(stderr, "cannot open the outputs\n"); return 68; }
    char *name = NULL; size_t name_cap = 0;
    char qual[MAXLINE + 1];
    memset(qual, 'S', R); qual[R] = 0;                                       /* CL:602-606 */
    for (size_t j = 0; j < m; j++) {
        node_t *nd = &nodes[order[j]];
        const size_t nl = (size_t)D * nd->count;
        if (nl + 1 > name_cap) { name_cap = 2 * nl + 64; name = (char *)realloc(name, name_cap); }
        size_t w = 0;
        for (uint32_t e = nd->id_head; e; e = id_next[e - 1]) {               /* CL:491-506 digits */
            uint64_t v = id_val[e - 1];
            for (int k = D - 1; k >= 0; k--) { name[w + k] = (char)(v % 94 + 33); v /= 94; }
            w += D;
        }
        name[nl] = 0;
        if (nd->ref_code == 0) {
            int lrn;
            const char c1 = nl >= 2 ? name[nl - 2] : 0, c2 = name[nl - 1];
            if (mapper == BWA) lrn = nl > 5000000 || (c1 == '/' && (c2 == '1' || c2 == '2'));          /* CL:2339-2342 */
            else if (mapper == BWA_MEM) lrn = nl > 5000000 || (c1 == '/' && c2 >= '0' && c2 <= '9');   /* CL:2360-2363 */
            else lrn = nl > 194;                                                                       /* CL:2321, 2381 */
            const char lead = mapper == BOWTIE_2 ? '@' : '>';
            if (lrn) {
                char code[64];
                lrn_count++;
                int n = sprintf(code + 1, "%d", (int)lrn_count);                                       /* CL:312-326 */
                code[0] = '#'; code[n + 1] = '#';
                if ((n + 2) % D != 0) code[n + 2] = 0; else { code[n + 2] = '#'; code[n + 3] = 0; }
                fprintf(fl, "%s\t%s\n", code, name);
                fprintf(fr, "%c%s\n", lead, code);
            } else fprintf(fr, "%c%s\n", lead, name);
            fwrite(pool + nd->key, 1, R, fr);
            if (mapper == BOWTIE_2) fprintf(fr, "\n+\n%s\n", qual); else fputc('\n', fr);
        } else {
            fprintf(fp, "%s\t%d\t%d\t%d\n", name, (int)nd->ref_code, (int)nd->ref_pos, R);             /* CL:2400-2401 */
        }
    }
    fclose(fr); fclose(fp); fclose(fl);
    FILE *fk = fopen(argv[15], "w");
    if (fk) { fputs("Y\n", fk); fclose(fk); }                                 /* CL:2425-2426 */
    size_t n_other = 0;                       /* elements under keys of windows with a byte outside A C G T N (see the header) */
    for (size_t i = 0; i < n_nodes; i++) {
        int o = nodes[i].klen != R;
        for (int k = 0; k < nodes[i].klen && !o; k++) o = !comp(pool[nodes[i].key + k]);
        n_other += (size_t)o;
    }
    fprintf(stderr, "read-mers %llu  distinct %zu  printed %zu  other-letter-keys %zu\n", (unsigned long long)item_index, n_nodes - n_other, m, n_other);
    return 0;
}
