/*
 * fastqsplit_oracle.c -- TEST INFRASTRUCTURE ONLY (checker for tests/ and bench.py's CPU legs).
 *
 * Plain-C restatement of CORA's `fastqSplit` (/root/reference/fastqSplit.cpp, FS below): every read-mer of the input
 * (reads, mates, splits in file order) goes to the fragment file of the group its signal belongs to -- the signal is the
 * read-mer's first and last base (2-character signals, FS:131-134) or its first two and last two bases (4-character
 * signals, FS:136-139) -- as a record of 8 bytes (2 x running read-mer index, little endian, FS:157-173) + the read-mer's
 * characters.  Signals the file does not list belong to group 0 (the tables are zero-initialised globals, FS:16-17).
 *
 * Kept as the reference has it: with 2-character signals, THREEWAY_PAIRED writes the SECOND third of mate 1 as the body
 * of its third record (FS:579-580; the group is taken from the third third); records are found by skipping 2 L + 5 bytes
 * from a sequence line's start and then to the next line end (FS:257-283), here over the whole file in memory instead of
 * 2 MB buffers.  A signal byte above 'T' would index outside the reference's tables: refused here with status 78.
 *
 * Pinned: tests/golden/fqsplit_manifest.json holds sha256 + size of every fragment file the REAL reference
 * (oracle/_ref/fastqSplit) wrote for all six mapping modes, both signal sizes and several group counts
 * (tests/golden/make_fqsplit_golden.py); tests/test_fqsplit_oracle.py compares.
 *
 *   fastqsplit_oracle <signals file> <read length> <FASTQ list> <output prefix> <mapping mode>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned char hash2[85][85], hash4[85][85][85][85];      /* FS:16-17 */
static unsigned n_groups, sig_size;
static const char NUC[5] = {'N', 'A', 'C', 'G', 'T'};            /* FS:21-40 */

static void die(int code, const char *msg) { fprintf(stderr, "%s\n", msg); exit(code); }

static void setup(const char *path)                              /* FS:54-129 */
{
    FILE *f = fopen(path, "r");
    if (!f) die(66, "cannot read the signals file");
    int lines;
    if (fscanf(f, "%d %u", &lines, &sig_size) != 2) die(7, "signals file: no header");
    if (sig_size != 2 && sig_size != 4) die(7, "fragsignalsize should be either 2 or 4");
    static char tok[1 << 20];
    while (fscanf(f, "%1048575s", tok) == 1) {
        char *s = tok;
        if (s[0] == 'N') {
            for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
                if (sig_size == 2) { if (a == 0 || b == 0) hash2[(int)NUC[a]][(int)NUC[b]] = (unsigned char)n_groups; }
                else for (int c = 0; c < 5; c++) for (int d = 0; d < 5; d++)
                    if (a == 0 || b == 0 || c == 0 || d == 0) hash4[(int)NUC[a]][(int)NUC[b]][(int)NUC[c]][(int)NUC[d]] = (unsigned char)n_groups;
            }
            s += 2;
        }
        const size_t len = strlen(s), step = sig_size + 1;
        if (len % step) die(7, "signals file: a group's text is not a whole number of signals");
        for (size_t i = 0; i < len / step; i++) {
            const unsigned char *q = (const unsigned char *)s + step * i;
            for (unsigned k = 0; k < sig_size; k++) if (q[k] >= 85) die(7, "signals file: byte above 'T'");
            if (sig_size == 2) hash2[q[0]][q[1]] = (unsigned char)n_groups;
            else hash4[q[0]][q[1]][q[2]][q[3]] = (unsigned char)n_groups;
        }
        n_groups++;
        if (n_groups > 255) die(7, "more than 255 groups");       /* FS:12 */
    }
    fclose(f);
}

typedef struct { char *p; size_t n, at, next; } reader_t;

static size_t skip_lines(const reader_t *r, size_t i, unsigned k)   /* FS:191-202 */
{
    while (k > 0 && i < r->n) { if (r->p[i] == '\n') k--; i++; }
    return i;
}

static void open_reader(reader_t *r, const char *path)              /* FS:235-243 */
{
    FILE *f = fopen(path, "rb");
    if (!f) die(67, "cannot read a FASTQ file");
    fseek(f, 0, SEEK_END);
    r->n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    r->p = malloc(r->n + 1);
    if (r->n && fread(r->p, 1, r->n, f) != r->n) die(67, "short read");
    fclose(f);
    r->at = 0;
    r->next = skip_lines(r, 0, 1);
}

static char *next_read(reader_t *r, unsigned min_skip)              /* FS:257-283 */
{
    r->at = r->next;
    r->next = skip_lines(r, r->at + min_skip, 1);
    if (r->next >= r->n && r->at >= r->n) return NULL;
    return r->p + r->at;
}

static FILE *fout[256];

static void emit(const char *sig_from, const char *body, unsigned len, uint64_t code)   /* FS:157-173 */
{
    const unsigned char *s = (const unsigned char *)sig_from;
    unsigned g;
    if (sig_size == 2) {
        if (s[0] >= 85 || s[len - 1] >= 85) die(78, "a signal byte above 'T'");
        g = hash2[s[0]][s[len - 1]];
    } else {
        if (s[0] >= 85 || s[1] >= 85 || s[len - 2] >= 85 || s[len - 1] >= 85) die(78, "a signal byte above 'T'");
        g = hash4[s[0]][s[1]][s[len - 2]][s[len - 1]];
    }
    fwrite(&code, 8, 1, fout[g]);
    fwrite(body, 1, len, fout[g]);
}

int main(int argc, char **argv)
{
    if (argc != 6) die(1, "usage: fastqsplit_oracle <signals> <read length> <FASTQ list> <output prefix> <mode>");
    const char *mode = argv[5];
    const int L = atoi(argv[2]);
    unsigned splits, paired;
    if (!strcmp(mode, "SINGLE")) { splits = 1; paired = 0; }
    else if (!strcmp(mode, "SPLIT_SINGLE")) { splits = 2; paired = 0; }
    else if (!strcmp(mode, "THREEWAY_SINGLE")) { splits = 3; paired = 0; }
    else if (!strcmp(mode, "PAIRED")) { splits = 1; paired = 1; }
    else if (!strcmp(mode, "SPLIT_PAIRED")) { splits = 2; paired = 1; }
    else if (!strcmp(mode, "THREEWAY_PAIRED")) { splits = 3; paired = 1; }
    else die(5, "ERROR: Mapping mode undefined... ");
    const unsigned R = (unsigned)L / splits, min_skip = 2u * (unsigned)L + 5u;       /* FS:331, 388, 424 */
    FILE *lst = fopen(argv[3], "r");
    if (!lst) die(66, "cannot read the FASTQ list");
    int samples;
    if (fscanf(lst, "%d", &samples) != 1) die(66, "FASTQ list: no sample count");
    setup(argv[1]);
    for (unsigned g = 0; g < n_groups; g++) {                                          /* FS:341-347 */
        static char name[4096];
        snprintf(name, sizeof name, "%s.frag%u", argv[4], g);
        fout[g] = fopen(name, "wb");
        if (!fout[g]) die(68, "cannot write a fragment file");
    }
    uint64_t code = 0;
    static char f1[4096], f2[4096];
    for (int k = 0; k < samples; k++) {
        long long count;
        reader_t a, b;
        if (paired) { if (fscanf(lst, "%4095s %4095s %lld", f1, f2, &count) != 3) die(66, "FASTQ list: short line"); }
        else if (fscanf(lst, "%4095s %lld", f1, &count) != 2) die(66, "FASTQ list: short line");
        open_reader(&a, f1);
        if (paired) open_reader(&b, f2);
        for (;;) {
            char *r1 = next_read(&a, min_skip), *r2 = paired ? next_read(&b, min_skip) : NULL;
            if (!r1) break;
            if (paired && !r2) die(6, "mate file ends early");
            for (unsigned m = 0; m <= paired; m++) {
                char *r = m ? r2 : r1;
                for (unsigned s = 0; s < splits; s++, code += 2) {
                    const char *body = r + s * R;
                    if (sig_size == 2 && paired && splits == 3 && m == 0 && s == 2) body = r + R;      /* FS:579-580 */
                    emit(r + s * R, body, R, code);
                }
            }
        }
        free(a.p);
        if (paired) free(b.p);
    }
    for (unsigned g = 0; g < n_groups; g++) fclose(fout[g]);
    return 0;
}
