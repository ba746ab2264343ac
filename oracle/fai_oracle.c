/*
 * fai_oracle.c -- TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 *
 * Plain-C restatement of `cora faiGenerate` (/root/reference/cora.cpp:1006-1029), which is this awk program
 * run under LC_ALL=C:
 *
 *   substr($1,1,1) == ">" { if (curChr != 0) print name, chrLen, chrOffset, lineBases, lineBases+1
 *                           name = substr($1,2); curChr++; prevLineIsChrName = 1;
 *                           totalOffset += length($0)+1; chrOffset = totalOffset; chrLen = 0 }
 *   substr($1,1,1) != ">" { if (prevLineIsChrName) { prevLineIsChrName = 0; lineBases = length($1) }
 *                           totalOffset += length($0)+1; chrLen += length($1) }
 *   END                   { print name, chrLen, chrOffset, lineBases, lineBases+1 }
 *
 * $1 = the first run of bytes that are not space / tab (awk's default field splitting), lengths in bytes.
 * Pinned: tests/test_fai.py compares it with the output of the real awk command for every FASTA under
 * tests/golden (expected .fai files written by tests/golden/make_fai_golden.py, which runs that command).
 *
 *   fai_oracle <ref.fa>            the .fai text on stdout
 *   fai_oracle --bases <ref.fa>    one line per contig: its bases = the $1 bytes of its sequence lines, joined
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    int bases_mode = argc == 3 && strcmp(argv[1], "--bases") == 0;
    if (argc != 2 && !bases_mode) { fprintf(stderr, "usage: fai_oracle [--bases] <ref.fa>\n"); return 2; }
    FILE *f = fopen(argv[argc - 1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[argc - 1]); return 1; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc(n + 1);
    if (n && fread(b, 1, n, f) != (size_t)n) return 1;
    fclose(f);

    long curChr = 0, chrLen = 0, lineBases = 0, totalOffset = 0, chrOffset = 0;
    int prevLineIsChrName = 0;
    const char *name = ""; long nameLen = 0;
    long s = 0;
    while (s < n) {
        long e = s;
        while (e < n && b[e] != '\n') e++;                 /* record = [s, e) */
        long t = s;
        while (t < e && (b[t] == ' ' || b[t] == '\t')) t++;
        long u = t;
        while (u < e && b[u] != ' ' && b[u] != '\t') u++;   /* $1 = [t, u) */
        if (u > t && b[t] == '>') {
            if (curChr != 0) {
                if (bases_mode) printf("\n");
                else printf("%.*s\t%ld\t%ld\t%ld\t%ld\n", (int)nameLen, name, chrLen, chrOffset, lineBases, lineBases + 1);
            }
            name = b + t + 1; nameLen = u - t - 1;
            curChr++; prevLineIsChrName = 1;
            totalOffset += (e - s) + 1; chrOffset = totalOffset; chrLen = 0;
        } else {
            if (prevLineIsChrName) { prevLineIsChrName = 0; lineBases = u - t; }
            totalOffset += (e - s) + 1; chrLen += u - t;
            if (bases_mode && curChr != 0) fwrite(b + t, 1, u - t, stdout);
        }
        s = e + 1;
    }
    if (bases_mode) { if (curChr != 0) printf("\n"); }
    else printf("%.*s\t%ld\t%ld\t%ld\t%ld\n", (int)nameLen, name, chrLen, chrOffset, lineBases, lineBases + 1);
    free(b);
    return 0;
}
