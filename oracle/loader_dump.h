/* TEST INFRASTRUCTURE ONLY.  The "loaded image" of the two homology tables as flat arrays, and its dump file:
 * the 18 arrays below in this order, each as a u64 byte count followed by the data (little endian).
 * Written by oracle/loader_oracle.c and oracle/ref_loader_main.cpp, read by tests/helpers.py; the HIP loader exports
 * the same arrays (include/cora_hip.h, enum cora_hip_tables_array).  Indices are 0-based here; the hub VALUES are the
 * reference's 1-based list indices, 0 = nothing at this position. */
#ifndef CORA_LOADER_DUMP_H
#define CORA_LOADER_DUMP_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct loader_dump {
    int d;                       /* offsets per inexact item */
    uint64_t n_eq, m_eq;         /* exact records, their entries */
    uint64_t n_hm, m_hm;         /* inexact nodes, their items */
    uint64_t G;                  /* sum of contig lengths = hub length */
    uint16_t *eq_chr;  uint32_t *eq_pos;  uint32_t *eq_block; uint64_t *eq_off;     /* n_eq (+1 for off) */
    uint8_t *eqi_dir;  uint16_t *eqi_chr; uint32_t *eqi_pos;                        /* m_eq */
    uint16_t *hm_chr;  uint32_t *hm_pos;  uint64_t *hm_off;   uint8_t *hm_maxrev;   /* n_hm */
    uint8_t *hmi_dir;  uint16_t *hmi_chr; uint32_t *hmi_pos;  uint8_t *hmi_block; uint8_t *hmi_offs;  /* m_hm (offs: m_hm*d) */
    uint32_t *equiv_hub, *hom_hub;                                                  /* G */
} loader_dump;

static void *ld_zalloc(uint64_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if (!p) { fprintf(stderr, "out of memory\n"); exit(9); } return p; }

static void loader_dump_alloc(loader_dump *D)
{
    D->eq_chr = (uint16_t *)ld_zalloc(D->n_eq, 2);  D->eq_pos = (uint32_t *)ld_zalloc(D->n_eq, 4);
    D->eq_block = (uint32_t *)ld_zalloc(D->n_eq, 4); D->eq_off = (uint64_t *)ld_zalloc(D->n_eq + 1, 8);
    D->eqi_dir = (uint8_t *)ld_zalloc(D->m_eq, 1);  D->eqi_chr = (uint16_t *)ld_zalloc(D->m_eq, 2); D->eqi_pos = (uint32_t *)ld_zalloc(D->m_eq, 4);
    D->hm_chr = (uint16_t *)ld_zalloc(D->n_hm, 2);  D->hm_pos = (uint32_t *)ld_zalloc(D->n_hm, 4);
    D->hm_off = (uint64_t *)ld_zalloc(D->n_hm + 1, 8); D->hm_maxrev = (uint8_t *)ld_zalloc(D->n_hm, 1);
    D->hmi_dir = (uint8_t *)ld_zalloc(D->m_hm, 1);  D->hmi_chr = (uint16_t *)ld_zalloc(D->m_hm, 2); D->hmi_pos = (uint32_t *)ld_zalloc(D->m_hm, 4);
    D->hmi_block = (uint8_t *)ld_zalloc(D->m_hm, 1); D->hmi_offs = (uint8_t *)ld_zalloc(D->m_hm * (uint64_t)D->d, 1);
    D->equiv_hub = (uint32_t *)ld_zalloc(D->G, 4);  D->hom_hub = (uint32_t *)ld_zalloc(D->G, 4);
}

static int ld_put(FILE *f, const void *p, uint64_t bytes) { return fwrite(&bytes, 8, 1, f) != 1 || (bytes && fwrite(p, 1, bytes, f) != bytes); }

static int loader_dump_write(const loader_dump *D, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    int bad = 0;
    bad |= ld_put(f, D->eq_chr, D->n_eq * 2);   bad |= ld_put(f, D->eq_pos, D->n_eq * 4);
    bad |= ld_put(f, D->eq_block, D->n_eq * 4); bad |= ld_put(f, D->eq_off, (D->n_eq + 1) * 8);
    bad |= ld_put(f, D->eqi_dir, D->m_eq);      bad |= ld_put(f, D->eqi_chr, D->m_eq * 2);  bad |= ld_put(f, D->eqi_pos, D->m_eq * 4);
    bad |= ld_put(f, D->hm_chr, D->n_hm * 2);   bad |= ld_put(f, D->hm_pos, D->n_hm * 4);
    bad |= ld_put(f, D->hm_off, (D->n_hm + 1) * 8); bad |= ld_put(f, D->hm_maxrev, D->n_hm);
    bad |= ld_put(f, D->hmi_dir, D->m_hm);      bad |= ld_put(f, D->hmi_chr, D->m_hm * 2);  bad |= ld_put(f, D->hmi_pos, D->m_hm * 4);
    bad |= ld_put(f, D->hmi_block, D->m_hm);    bad |= ld_put(f, D->hmi_offs, D->m_hm * (uint64_t)D->d);
    bad |= ld_put(f, D->equiv_hub, D->G * 4);   bad |= ld_put(f, D->hom_hub, D->G * 4);
    bad |= fclose(f) != 0;
    return bad;
}
#endif
