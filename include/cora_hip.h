/*
 * cora_hip.h -- C-ABI of libcora_hip.so, the MI355X (gfx950) homology-table
 * builder that stands in for CORA's homTable_setup hot path.
 *
 * The reference has no FFI for this path: its boundary is the process
 *   homTable_setup[_L] <ref.fa> <K> <physSplits> <ExactOut> <InexactOut>
 *                      <parSplits> <EXACT|BOTH> <d> <FULL|...>
 * (/root/reference/homTable_setup.cpp:2389-2405, called from cora.cpp:1586).
 * The drop-in executable of the same name (cora_amd/csrc/hts_main.cpp) keeps
 * that command line and the on-disk formats and calls the functions below.
 * Plain C types only: pointers, sizes, integer status (0 = ok, otherwise the
 * reference's exit code where one exists); no exceptions cross this seam.
 * HTS = /root/reference/homTable_setup.cpp.
 */
#ifndef CORA_HIP_H
#define CORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORA_HIP_MAX_SEEDS 4

/* One contig of the reference, as the loader needs it.
 * Replaces the per-contig state of ConfigureRelatedParameters / ReadReference
 * (HTS:144-202, 96-142): name order and length come from the .fai (columns 1-2),
 * file_offset is where the contig's first base sits in the FASTA file image. */
typedef struct cora_hip_contig {
    uint64_t file_offset;   /* byte offset of the first base in the file image */
    uint32_t length;        /* bases (from .fai column 2) */
    uint32_t reserved;
} cora_hip_contig;

/* Run parameters (argv[2], argv[8], -DCHR_SHORT, and the two split files after
 * parsing: HTS:2418-2460, 2295-2346). */
typedef struct cora_hip_params {
    int32_t K;                    /* read-mer length, 33..64 (HTS:2418-2424) */
    int32_t d;                    /* Hamming radius 1..3 (HTS:2017-2021) */
    int32_t chr_short;            /* 0: homTable_setup (1-byte contig ids), 1: homTable_setup_L (2-byte) */
    int32_t line_width;           /* bases per FASTA line (HTS:155-166) */
    int32_t n_groups;             /* physical splits p (first line of argv[3]) */
    int32_t n_threads;            /* parallel splits t (first line of argv[6]) */
    uint8_t sig2[16];             /* group (1..p) of 2-mer [4*first+last], codes A0 C1 G2 T3 (HTS:2292, 2503-2509) */
    uint8_t sig4[256];            /* thread (1..t) of 4-mer token [64*c0+16*c1+4*c2+c3] (HTS:2291, 2327-2346) */
} cora_hip_params;

/* Per-run figures a caller may want to print or benchmark with. */
typedef struct cora_hip_stats {
    uint64_t genome_bases;        /* sum of contig lengths */
    uint64_t kmers;               /* valid K-mer windows (Q4 rule applied, HTS:1954-1981) */
    uint64_t representatives;     /* distinct canonical K-mers = marked positions (HTS:2573-2578) */
    uint64_t exact_classes;       /* classes with >= 2 entries (HTS:336-347) */
    uint64_t exact_nodes;         /* records in <Exact> after block merging (HTS:572-635) */
    uint64_t candidate_pairs;     /* K-mer comparisons done, all seeds (HTS:2753-2762) */
    uint64_t hom_pairs;           /* pair records reported (HTS:2234-2264) */
    uint64_t inexact_items;       /* directed items before block merging (HTS:1022-1103) */
    uint64_t inexact_nodes;       /* nodes written to <Inexact> (HTS:1353-1432) */
    uint64_t dirty_nodes;         /* nodes with tied targets, finished by the exact-order host path */
    uint64_t big_nodes;           /* nodes with > 16 items (std::sort's unstable regime, HTS:1368) */
    /* timings of the last run, milliseconds (HIP events on the job's stream; host = wall) */
    float ms_pack, ms_exact, ms_seeds, ms_items, ms_host_finish, ms_total;
    float ms_sort_scatter;        /* time inside radix scatter launches (the dominant kernel) */
    uint32_t n_sort_scatter;      /* number of those launches */
    uint64_t sort_scatter_bytes;  /* algorithmic bytes those launches moved (read + write of every record) */
    uint64_t hash_mixed_runs;     /* equal-hash runs that held different K-mers and were re-ordered by key */
    uint64_t full_key_fallback;   /* 1 if the exact stage had to redo its sort on the full 128-bit keys */
    uint64_t seed_sketch;         /* 1 if the seed records carried the other seeds' codes through their sorts (large references) */
    uint64_t verify_slots;        /* partners the sorted records could not dismiss: K-mer pairs read from the genome, all seeds */
    uint64_t pool_peak_bytes;     /* most device memory the job's stage arrays held at once since the job was created (this rank's, in a sharded build) */
} cora_hip_stats;

typedef struct cora_hip_job cora_hip_job;   /* opaque */

/* Library / device set-up.  device = HIP ordinal.  Fails (non-zero) when no
 * gfx950 device is usable: there is no CPU fallback. */
int cora_hip_init(int device);
int cora_hip_shutdown(void);
/* Text of the last error on the calling thread ("" if none). */
const char *cora_hip_last_error(void);

/* Creates a job for one reference.  Replaces main()'s set-up (HTS:2407-2471). */
int cora_hip_job_create(const cora_hip_params *params, cora_hip_job **out);
void cora_hip_job_destroy(cora_hip_job *job);

/* Hands over the FASTA file image (host memory) and its contig table and
 * copies the image to HBM.  Replaces ReadReference (HTS:96-142): the kernels
 * read bases straight from the file image using the fixed line width, so the
 * host does no per-base work. */
int cora_hip_job_set_reference(cora_hip_job *job, const uint8_t *fasta_image, size_t image_bytes,
                               const cora_hip_contig *contigs, int32_t n_contigs);

/* Builds both tables from the HBM-resident image: 2-bit pack, canonical K-mers,
 * radix sort, classes and representatives (HTS:2499-2640, 249-497), per-seed
 * code sort + equal-run pair verification (HTS:2671-2800, 2138-2265), directed
 * item sort and diagonal block merge (HTS:1022-1270), final ordering and
 * serialisation (HTS:572-635, 1353-1432).  May be called repeatedly. */
int cora_hip_job_run(cora_hip_job *job);

/* The tables of the last run, in the reference's on-disk byte layout
 * (F1/F2 of SURVEY.md section 8a); pointers stay valid until the next run or destroy. */
int cora_hip_job_tables(const cora_hip_job *job,
                        const uint8_t **exact, size_t *exact_bytes, uint64_t *exact_nodes,
                        const uint8_t **inexact, size_t *inexact_bytes, uint64_t *inexact_nodes);
int cora_hip_job_stats(const cora_hip_job *job, cora_hip_stats *out);

/* The tables straight into their files, for callers that want files and nothing else (the drop-in executable).  Replaces the fopen /
 * fwrite / fclose of PrintBinaryEquivData and PrintBinaryHomMapData_SplitAware (HTS:572-635, 1353-1432, 1451-1456, 2485-2489).
 * Call before cora_hip_job_run[_multi]: <Inexact> is then assembled in a shared mapping of inexact_path (no copy of the 0.8 GB
 * into the page cache afterwards) and <Exact> is written to exact_path by a helper thread as soon as the exact stage is through,
 * beside the inexact stages.  cora_hip_job_finish_table_files cuts <Inexact> to its size, closes both files and says whether
 * every write succeeded (status 68, the reference's "Can't write to file").  Between the two calls cora_hip_job_tables
 * still shows both tables; after the second the <Inexact> pointer is NULL (the bytes are the file's). */
int cora_hip_job_set_table_files(cora_hip_job *job, const char *exact_path, const char *inexact_path);
int cora_hip_job_finish_table_files(cora_hip_job *job);

/* ---- multi-GPU: one process per GPU, no counterpart in the reference (it has no distributed code) ----
 * Every rank holds the packed reference, scans 1/R of the positions and owns 1/R of the key space.  The
 * library computes per rank; the caller moves the "slots" between ranks with its collective library
 * (cora_amd/dist.py: torch.distributed all_to_all_single / all_reduce over RCCL).  Stage order and the
 * exchange after each stage are listed at the top of cora_amd/csrc/shard.hip.  All pointers handed to
 * slot_export / slot_import are DEVICE pointers.  Restates, sharded, the same reference phases as
 * cora_hip_job_run (HTS:2499-2800, 249-1638). */
enum cora_hip_shard_stage_id { CORA_ST_KMERS = 1, CORA_ST_EXACT, CORA_ST_PARTS, CORA_ST_MERGE, CORA_ST_REPS,
                               CORA_ST_SEED, CORA_ST_VERIFY, CORA_ST_ITEMS, CORA_ST_FINISH, CORA_ST_ITEM_CUT, CORA_ST_COLLECT };
enum cora_hip_shard_slot_id { CORA_SLOT_SEND_A = 0, CORA_SLOT_SEND_C, CORA_SLOT_RECV_A, CORA_SLOT_RECV_C, CORA_SLOT_REPMASK,
                              CORA_SLOT_X_CLS, CORA_SLOT_X_CNT, CORA_SLOT_X_ENT, CORA_SLOT_X_DIR,
                              CORA_SLOT_RX_CLS, CORA_SLOT_RX_CNT, CORA_SLOT_RX_ENT, CORA_SLOT_RX_DIR,
                              CORA_SLOT_WCOUNT, CORA_SLOT_F2, CORA_SLOT_RX_F2, CORA_SLOT_STATS,
                              CORA_SLOT_SEND_B, CORA_SLOT_RECV_B /* the seed records' sketches (large references): moved like SEND_A / RECV_A when present */ };
/* The same flow from ONE process: `world` ranks, one host thread each, rank r on HIP device devices[r] (NULL = rank r on
 * device r modulo the device count; devices may repeat -- several ranks on one GPU, the rehearsal the tests run).
 * devices[0] must be the device the job was created on.  The other ranks' replicas of the reference are copied from
 * the job's own (device to device) and kept for later calls; records move between the ranks by peer-to-peer copies over
 * xGMI (hipMemcpyPeerAsync, every receiver pulling from every sender).  On return the job holds both tables and the
 * whole-job counts, exactly as after cora_hip_job_run -- byte for byte the same tables.  This is what the drop-in
 * executable runs when it finds more than one GPU (cora.cpp:1586 starts one process, whatever the machine holds). */
int cora_hip_job_run_multi(cora_hip_job *job, int world, const int *devices);
int cora_hip_device_count(int *n);
int cora_hip_shard_begin(cora_hip_job *job, int rank, int world);
/* Runs one stage; counts[world] (may be NULL) receives how many records of the SEND slots go to each rank. */
int cora_hip_shard_stage(cora_hip_job *job, int stage, int arg, uint64_t *counts);
int cora_hip_shard_slot_info(cora_hip_job *job, int slot, uint64_t *n_elems, uint32_t *elem_bytes);
int cora_hip_shard_slot_export(cora_hip_job *job, int slot, void *dst_device);
/* Device address of a slot's data, valid until the job's next stage or import call (send without a copy). */
int cora_hip_shard_slot_ptr(cora_hip_job *job, int slot, void **ptr);
int cora_hip_shard_slot_import(cora_hip_job *job, int slot, const void *src_device, uint64_t n_elems);
/* Sizes a receive slot (RECV_*, RX_*) for n_elems and returns its device address: the caller's collective then delivers
 * straight into the library's memory instead of into a tensor that slot_import would copy. */
int cora_hip_shard_slot_reserve(cora_hip_job *job, int slot, uint64_t n_elems, void **ptr);
/* After a SEED stage: 1 if the seed records carry sketches, i.e. SEND_B / RECV_B must be moved like SEND_A / RECV_A (same answer on every rank). */
int cora_hip_shard_seed_sketch(cora_hip_job *job, int *on);
/* How the K-mer and seed records reach their owner in this build (chosen by cora_hip_shard_begin: world <= 4, or
 * CORA_SHARD_FILTER=0/1).  *on = 0: the caller moves SEND_* to RECV_* by all-to-all after the KMERS and SEED stages.
 * *on = 1: every rank scans ALL positions itself and nothing is moved after KMERS and SEED: the records of every position
 * stay in SEND_* and the first pass of the following EXACT / VERIFY stage's sort reads them there and keeps the ones this rank
 * owns (a GPU pair shares one xGMI link: on 2 or 4 GPUs the all-to-all costs more than the extra scan).
 * set >= 0 overrides the choice (before the KMERS stage; every rank the same).  No counterpart in the reference. */
int cora_hip_shard_filter(cora_hip_job *job, int set, int *on);
int cora_hip_shard_get_first_singleton(cora_hip_job *job, uint64_t *key);
int cora_hip_shard_set_first_singleton(cora_hip_job *job, uint64_t key);

/* ---- the consumer side of the tables: load <Exact>/<Inexact> into HBM ----
 * MI = /root/reference/mappingInference.cpp.  Replaces LoadHomologyTables (MI:1084-1153), i.e. LoadEquivClasses
 * (MI:654-723), LoadInexactHomologies incl. the reverse-lookup walk (MI:554-651) and the position hubs equivHub /
 * homHub (MI:537-538, 1093-1152).  The lists become CSR arrays, the two hash maps become dense u32 arrays over the
 * concatenated contigs (index = sum of earlier contig lengths + pos - 1); hub VALUES are the reference's 1-based list
 * indices and 0 = no entry, which is how every reader of the maps treats an absent key (MI:1841-1848, 6620-6630).
 * Everything is computed on the GPU from the table bytes; results equal the reference's loaders field by field. */
typedef struct cora_hip_tables cora_hip_tables;   /* opaque */

enum cora_hip_tables_array {            /* element type, length */
    CORA_TA_EQ_CHR = 0,                 /* u16  records        equivClassNode::chrCode (MI:520-531) */
    CORA_TA_EQ_POS,                     /* u32  records        ::chrPos */
    CORA_TA_EQ_BLOCK,                   /* u32  records        ::blockLen */
    CORA_TA_EQ_OFF,                     /* u64  records + 1    entries of record i are [off[i], off[i+1])  (::list, ::listSize) */
    CORA_TA_EQI_DIR,                    /* u8   entries        eqItem::dir (MI:508-517) */
    CORA_TA_EQI_CHR,                    /* u16  entries */
    CORA_TA_EQI_POS,                    /* u32  entries */
    CORA_TA_HM_CHR,                     /* u16  nodes          emHomClassNode::chrCode (MI:488-499) */
    CORA_TA_HM_POS,                     /* u32  nodes */
    CORA_TA_HM_OFF,                     /* u64  nodes + 1 */
    CORA_TA_HM_MAXREV,                  /* u8   nodes          ::maxRevLookup (MI:638-650) */
    CORA_TA_HMI_DIR,                    /* u8   items          emHomItem::dir, 1 bit (MI:475-486) */
    CORA_TA_HMI_CHR,                    /* u16  items */
    CORA_TA_HMI_POS,                    /* u32  items */
    CORA_TA_HMI_BLOCK,                  /* u8   items          ::blockLen, 7 bits */
    CORA_TA_HMI_OFFS,                   /* u8   items * d      ::offsetList[0..d) */
    CORA_TA_EQUIV_HUB,                  /* u32  hub positions  equivHub (MI:537) */
    CORA_TA_HOM_HUB,                    /* u32  hub positions  homHub (MI:538) */
    CORA_TA_COUNT
};

typedef struct cora_hip_tables_info {
    uint64_t exact_records, exact_entries;   /* equivClassListSize, sum of listSize */
    uint64_t inexact_nodes, inexact_items;   /* emHomClassListSize, sum of listSize */
    uint64_t hub_positions;                  /* sum of contig lengths */
    uint64_t device_bytes;                   /* HBM held by the loaded image */
    uint64_t serial_hub_replay;              /* 1 if some hub position was both copied from and copied to (no table builder
                                                writes that): homHub was then filled in the reference's sequential order */
    int32_t d, chr_short;
    float ms_parse, ms_lookups, ms_hubs, ms_total;   /* HIP events; table bytes already in HBM when the clock starts */
} cora_hip_tables_info;

/* exact / inexact: the two files' bytes (host memory); *_count: the number in "<table>.itemCount" (MI:559-567, 657-665).
 * contig_len[n_contigs]: contig lengths in .fai order.  exact_only != 0 is the loading of the BEST mapping modes
 * (MI:1090-1093, 1139-1152): no inexact table, equivHub over the representatives' ranges only.
 * Errors (non-zero, text in cora_hip_last_error): 87/88/89/90 are the reference's exit codes for unreadable inputs and
 * are returned by the file-level callers; this function returns 91 for bytes that do not parse to exactly *_count
 * records, 92 for nodes out of (contig, position) order or without items, 93 for positions outside the contigs
 * (the reference's maps would take any key; the dense hubs cannot). */
int cora_hip_tables_load(const uint8_t *exact, size_t exact_bytes, uint64_t exact_count,
                         const uint8_t *inexact, size_t inexact_bytes, uint64_t inexact_count,
                         const uint32_t *contig_len, int32_t n_contigs, int32_t d, int32_t chr_short, int32_t exact_only,
                         cora_hip_tables **out);
/* Same, straight from the tables of a job's last run (no trip through files); contig lengths come from the job. */
int cora_hip_tables_from_job(const cora_hip_job *job, int32_t exact_only, cora_hip_tables **out);
int cora_hip_tables_get_info(const cora_hip_tables *t, cora_hip_tables_info *out);
/* Device address, element count and element size of one array of the loaded image (for the caller's own kernels). */
int cora_hip_tables_array_ptr(const cora_hip_tables *t, int which, const void **device_ptr, uint64_t *n_elems, uint32_t *elem_bytes);
/* Copies one array to host memory; dst_bytes must be at least n_elems * elem_bytes. */
int cora_hip_tables_export(const cora_hip_tables *t, int which, void *dst_host, uint64_t dst_bytes);
/* RetrieveOffsetAndDirInfo (MI:6618-6674) for n positions at once: class_index[i] = 1-based index of the exact class
 * that covers (chr[i], pos[i]), 0 if none; block_offset[i] / dir[i] = where in the class's block the position sits and
 * whether it maps to the representative reversed.  on_device != 0: all five arrays are DEVICE pointers (results stay in
 * HBM for the caller's kernels); otherwise host arrays are copied in and out.  94: a table whose hub names a class that
 * does not hold the position (the reference asserts). */
int cora_hip_tables_lookup_equiv(const cora_hip_tables *t, const uint32_t *chr, const uint32_t *pos, uint64_t n, int on_device,
                                 uint32_t *class_index, int16_t *block_offset, uint8_t *dir);
void cora_hip_tables_destroy(cora_hip_tables *t);

/* ---- the reference side's file handling: contig index and a loader for any FASTA ----
 * Replaces `cora faiGenerate` (/root/reference/cora.cpp:1006-1029, an awk program run through system()) and lifts
 * the fixed-line-width restriction of ReadReference (HTS:96-142).  The file image is scanned on the GPU: separator
 * positions, one thread per line, scans for header rank and sequence bytes, one record per contig. */
typedef struct cora_hip_fasta cora_hip_fasta;   /* opaque: the image in HBM and its line / contig index */

typedef struct cora_hip_fasta_info {
    uint64_t image_bytes, lines, contigs, bases;   /* bases = sum of the contig lengths the .fai reports */
    int32_t conforming;      /* 1: headers start their lines and every contig's lines hold line_width bytes but its last
                                (1..line_width): the layout ReadReference assumes, read in place without a copy */
    int32_t line_width;      /* lineBases of the first contig (cora.cpp:1023), 0 if it has no sequence line */
    float ms_scan;           /* HIP events; image already in HBM when the clock starts */
} cora_hip_fasta_info;

/* Scans a FASTA file image (host memory; copied to HBM). */
int cora_hip_fasta_scan(const uint8_t *image, size_t image_bytes, cora_hip_fasta **out);
int cora_hip_fasta_get_info(const cora_hip_fasta *f, cora_hip_fasta_info *out);
/* The bytes `cora faiGenerate` writes to <ref>.fai for this file: name, length, offset, lineBases, lineBases + 1 per
 * contig, byte for byte what the awk program prints (stale lineBases after an empty contig and the single line for
 * a file without any header included).  Valid until destroy. */
int cora_hip_fasta_fai_text(const cora_hip_fasta *f, const char **text, size_t *bytes);
/* One contig of the index (any output pointer may be NULL). */
int cora_hip_fasta_contig(const cora_hip_fasta *f, int32_t index, const char **name, uint64_t *length, uint64_t *offset,
                          uint64_t *line_bases);
void cora_hip_fasta_destroy(cora_hip_fasta *f);
/* cora_hip_job_set_reference without a .fai and without the fixed line width: contigs and lengths come from the scan
 * (a contig's bases = the first blank-delimited token of each of its lines, which is what the .fai length counts).
 * A conforming file is read in place exactly as cora_hip_job_set_reference reads it, so the tables are the same
 * bytes; any other file is first gathered into a dense image on the GPU.  The job's params.line_width is ignored. */
int cora_hip_job_set_reference_fasta(cora_hip_job *job, const cora_hip_fasta *f);

/* ---- the read side: `collapse`, the read-mer collapsing stage of `cora search` ----
 * CL = /root/reference/collapse.cpp.  Replaces its main() between "arguments parsed" and "files written" for the
 * whole-input run mode (argv[13] = NONE, cora.cpp:1674-1692): reference read-mers (CL:647-711), read-mers of the reads,
 * mates and splits (CL:861-2171), collapsing of identical canonical read-mers and the encoded ids (CL:276-289), and
 * the text of the three output files in the iteration order of the reference's hash map (CL:2295-2405). */
typedef struct cora_hip_collapse cora_hip_collapse;   /* opaque */

typedef struct cora_hip_collapse_params {
    int32_t digit_len;      /* argv[2]: characters per read-mer id (2..10) */
    int32_t read_len;       /* argv[6]: length of an input read */
    int32_t splits;         /* argv[11]: 1 FULL, 2 HALF, 3 THREEWAY (CL:568-588); read-mer length = read_len / splits */
    int32_t paired;         /* argv[10]: 0 SINGLE, 1 PAIRED (files are then added as mate 1, mate 2 per sample) */
    int32_t mapper;         /* argv[12]: 0 generic, 1 BWA, 2 BWA_MEM, 3 BOWTIE_2 (CL:590-611) */
    int32_t chr_short;      /* 0: collapse (up to 250 contigs), 1: collapse_L (-DCHR_SHORT, CL:62-77) */
    uint64_t lrn_offset;    /* argv[16]: long read names are numbered from this value + 1 (CL:541) */
} cora_hip_collapse_params;

typedef struct cora_hip_collapse_stats {
    uint64_t reads;            /* sequence lines read (mates count separately) */
    uint64_t read_mers;        /* reads x splits */
    uint64_t ref_windows;      /* N-free reference windows of read-mer length */
    uint64_t distinct;         /* distinct canonical read-mers (elements of the reference's hash map) */
    uint64_t printed;          /* those that hold at least one read-mer of the input (CL:2311-2314) */
    uint64_t lrn_names;        /* names moved to the long-read-name list (CL:2321-2328) */
    uint64_t hash_collisions;  /* neighbours with one hash and different read-mers, re-ordered by the repair kernel */
    uint64_t reads_bytes, perfect_bytes, lrn_bytes;
    float ms_parse, ms_keys, ms_group, ms_order, ms_text, ms_device, ms_total;   /* HIP events; ms_total = wall of the run call */
    float ms_sort_scatter;     /* time inside the radix scatter launches of this run */
    uint32_t n_sort_scatter;
    uint64_t sort_scatter_bytes;
    uint32_t table_growth_steps;   /* how often the reference's hash map would have re-bucketed itself during this run (0: below 182 356 234 read-mers) */
    uint32_t reserved0;
} cora_hip_collapse_stats;

/* Bucket count of the reference's std::tr1::unordered_map after max_load_factor(1 / 1.13) and rehash(200000000)
 * (CL:640-643) and the number of the element whose insertion makes it re-bucket itself for the first time, both asked of
 * libstdc++'s own rehash policy at run time.  The output order follows from the bucket counts; runs that pass the second
 * number are ordered through the table's whole growth history (cora_hip_collapse_table_growth). */
int cora_hip_collapse_table_geometry(uint64_t *buckets, uint64_t *first_rehash_at);
/* The growth history of that table for rehash(request) (the reference: 200000000): buckets[0] = the initial bucket count
 * (at[0] = 0); for k >= 1, at[k] = the number (1-based, over distinct keys in insertion order) of the element whose
 * insertion makes the table re-bucket itself first, to buckets[k] buckets.  Asked of libstdc++'s own rehash policy.
 * Returns the number of entries (at most cap), ending with the first one beyond max_elements. */
int cora_hip_collapse_table_growth(uint64_t request, uint64_t max_elements, uint64_t *at, uint64_t *buckets, int cap);
int cora_hip_collapse_create(const cora_hip_collapse_params *params, cora_hip_collapse **out);
/* The reference (argv[4]; leave out for NOREF): a scanned FASTA.  The .fai and the line length (argv[5]) are not needed. */
int cora_hip_collapse_set_reference(cora_hip_collapse *c, const cora_hip_fasta *reference);
/* One FASTQ file image (host memory, copied), in the order of the list file (argv[1]): per sample the reads file, then,
 * if paired, the mates file.  Records are four lines; the sequence line must hold at least read_len bytes of A C G T N
 * (status 78 otherwise: the reference would cut its keys at the first other byte). */
int cora_hip_collapse_add_reads(cora_hip_collapse *c, const uint8_t *fastq, size_t bytes);
/* The fragmented run mode (argv[13] != NONE, CL:542-563, 712-835, 2173-2284; one call per group, cora.cpp:1733-1760):
 * signals = argv[13] (the group's line of the signals file), signal_len = argv[14].  The ONE input added with
 * cora_hip_collapse_add_reads is then that group's fragment file as fastqSplit writes it (records of 8 bytes + a read-mer),
 * only reference windows whose signal the group lists enter the table, and ids are built on the numbers in the records. */
int cora_hip_collapse_set_fragment(cora_hip_collapse *c, const char *signals, int32_t signal_len);
/* One object can take group after group (cora.cpp:1733-1760 starts a process per group): cora_hip_collapse_set_fragment
 * drops the previous input and keeps the reference; this sets argv[16] for the next run (long read names are numbered on). */
int cora_hip_collapse_set_lrn_offset(cora_hip_collapse *c, uint64_t lrn_offset);
/* Fragmented mode only, in place of cora_hip_collapse_add_reads: the fragment is already in this GPU's memory (as
 * cora_hip_fqsplit_group_device hands it out) and is read where it lies: the fastqSplit -> collapse chain without the
 * fragments crossing PCIe.  No counterpart in the reference, whose two programs meet in files.  The memory is borrowed: it
 * must stay valid and unchanged until cora_hip_collapse_run has returned. */
int cora_hip_collapse_add_reads_device(cora_hip_collapse *c, const void *fragment_dev, size_t bytes);
int cora_hip_collapse_run(cora_hip_collapse *c);
/* The three files' bytes (argv[3], argv[7], argv[9]); valid until the next run or destroy. */
int cora_hip_collapse_outputs(const cora_hip_collapse *c, const uint8_t **reads, size_t *reads_bytes, const uint8_t **perfect,
                              size_t *perfect_bytes, const uint8_t **lrn, size_t *lrn_bytes);
int cora_hip_collapse_get_stats(const cora_hip_collapse *c, cora_hip_collapse_stats *out);
void cora_hip_collapse_destroy(cora_hip_collapse *c);

/* ---- the read side, before collapse: `fastqSplit`, signal-keyed partitioning of the reads ----
 * FS = /root/reference/fastqSplit.cpp.  Replaces its main() between "arguments and signals read" and "fragment files
 * written" (FS:341-867): every read-mer of the input goes, as 8 bytes of 2 x its running number + its characters
 * (FS:157-173), to the fragment file of the group its signal belongs to, in input order. */
typedef struct cora_hip_fqsplit cora_hip_fqsplit;   /* opaque */

typedef struct cora_hip_fqsplit_params {
    int32_t read_len;       /* argv[2] */
    int32_t splits;         /* argv[5]: 1 SINGLE / PAIRED, 2 SPLIT_*, 3 THREEWAY_* (read-mer length = read_len / splits) */
    int32_t paired;         /* argv[5]: 0 *SINGLE, 1 *PAIRED (files are then added as mate 1, mate 2 per sample) */
    int32_t signal_size;    /* 2: first and last base of a read-mer; 4: first two and last two (FS:131-139) */
    int32_t n_groups;       /* fragment files <prefix>.frag0 .. (FS:341-347), 1..255 */
    uint8_t group_of[628];  /* group of a signal: index ((s0 * 5 + s1) * 5 + s2) * 5 + s3, or s0 * 5 + s1 for 2-character signals,
                               with N 0, A 1, C 2, G 3, T 4 (FS:21-40); signals the file does not list are group 0 (FS:16-17) */
} cora_hip_fqsplit_params;

typedef struct cora_hip_fqsplit_stats {
    uint64_t reads, read_mers, record_bytes, output_bytes, image_bytes;
    float ms_parse, ms_classify, ms_order, ms_write, ms_device, ms_total;   /* HIP events; ms_total = wall of the run call */
} cora_hip_fqsplit_stats;

/* Host only: signal_size, n_groups and group_of from the text of the signals file (argv[1], FS:54-129; written by
 * cora.cpp:335-390).  Status 7 for what the reference exits 7 on or asserts against. */
int cora_hip_fqsplit_parse_signals(const char *text, size_t bytes, cora_hip_fqsplit_params *out);
int cora_hip_fqsplit_create(const cora_hip_fqsplit_params *params, cora_hip_fqsplit **out);
/* One FASTQ file image (host memory, copied), in the order of the list file (argv[3]): per sample the reads file, then, if
 * paired, the mates file.  Records are four lines (the reference finds them by skipping 2 x read_len + 5 bytes and then
 * to the next line end, FS:257-283: the same records when the third line is a bare '+').  Status 78 for a sequence line
 * shorter than read_len or a signal byte outside A C G T N (the reference copies across the line end / indexes outside
 * its tables there). */
int cora_hip_fqsplit_add_reads(cora_hip_fqsplit *s, const uint8_t *fastq, size_t bytes);
int cora_hip_fqsplit_run(cora_hip_fqsplit *s);
/* Bytes of <prefix>.frag<group>; valid until the next run or destroy. */
int cora_hip_fqsplit_group(const cora_hip_fqsplit *s, int32_t group, const uint8_t **bytes, size_t *n);
/* on != 0: runs leave the fragment files in HBM only (no copy to the host; cora_hip_fqsplit_group then fails with 19). */
int cora_hip_fqsplit_set_device_only(cora_hip_fqsplit *s, int32_t on);
/* The same bytes where the run left them in HBM (device pointer; valid until the next run or destroy). */
int cora_hip_fqsplit_group_device(const cora_hip_fqsplit *s, int32_t group, const void **dev, size_t *n);
int cora_hip_fqsplit_get_stats(const cora_hip_fqsplit *s, cora_hip_fqsplit_stats *out);
void cora_hip_fqsplit_destroy(cora_hip_fqsplit *s);

/* ---- stage-level entry points (used by the parity tests and the benchmark) ---- */

/* Stable LSD radix sort of n records on device memory, keyed on bits
 * [bit_lo, bit_hi) of the 128-bit key (key_hi:key_lo); key_lo and val may be
 * NULL.  All pointers are DEVICE pointers; tmp_* are same-sized scratch arrays.
 * On return the sorted records are in the original arrays.  This is the
 * kernel family behind every sort of the path (stands for the reference's
 * unordered_map / std::sort uses: HTS:2516, 2681, 450, 1141). */
int cora_hip_radix_sort(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val,
                        uint64_t *tmp_hi, uint64_t *tmp_lo, uint32_t *tmp_val,
                        uint64_t n, int bit_lo, int bit_hi);

/* cora_hip_radix_sort for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc, code that runs between the launches of a captured graph.  (Not for use INSIDE a stream capture: like
 * cora_hip_radix_sort it launches on the null stream and returns after hipStreamSynchronize, neither of which a capture
 * allows.)  It has NO counterpart in the reference.  Same arguments, same result (stable,
 * sorted records back in the original arrays); the per-tile digit histograms that cora_hip_radix_sort keeps in
 * hipMalloc'ed memory live in a 4.1 MiB static device array instead, which every GPU the library runs on holds until
 * the process ends.  Limit: n <= cora_hip_radix_sort_noalloc_max() = 2^24 records; more is refused with status 19.
 * The call makes no hipMalloc / hipFree / event creation.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU race (serialise them); nothing in the array has to be cleared between calls. */
int cora_hip_radix_sort_noalloc(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val,
                                uint64_t *tmp_hi, uint64_t *tmp_lo, uint32_t *tmp_val,
                                uint64_t n, int bit_lo, int bit_hi);
uint64_t cora_hip_radix_sort_noalloc_max(void);

/* Host-buffer convenience wrappers for tests: copy in, run the kernels, copy out. */
int cora_hip_test_sort_host(uint64_t *key_hi, uint64_t *key_lo, uint32_t *val, uint64_t n, int bit_lo, int bit_hi);
/* The two-word sort as the seed sorts of a large build run it: the second word does not ping-pong between two arrays but
 * climbs through `chain` arrays of one allocation, one per pass (1: the ping-pong).  Same result as cora_hip_test_sort_host.
 * No counterpart in the reference (a property of this sort: radix_sort.hip, "the second word climbs"). */
int cora_hip_test_sort_chain_host(uint64_t *key_hi, uint64_t *key_lo, uint64_t n, int bit_lo, int bit_hi, int chain);
/* The same sort with the owner filter of the multi-GPU flow on few GPUs (cora_hip_shard_filter): of the n records only those
 * whose owner is `me` of `world` ranks are kept -- owner = top 32 bits of key_hi (owner_mode 1: as they are, hash ranges;
 * 2: multiplied by 0x9E3779B1 mod 2^32, seed codes) * world >> 32 -- in the first pass of the sort, which reads all
 * records and writes the kept ones.  out_* (room for cap records) receive them sorted; *n_kept their number.  With key_lo the
 * bit range must lie in key_hi (bit_lo >= 64).  No counterpart in the reference. */
int cora_hip_test_sort_kept_host(const uint64_t *key_hi, const uint64_t *key_lo, const uint32_t *val, uint64_t n, int bit_lo, int bit_hi,
                                 int owner_mode, uint32_t world, uint32_t me, uint64_t *out_hi, uint64_t *out_lo, uint32_t *out_val,
                                 uint64_t cap, uint64_t *n_kept);
/* The device replica of libstdc++'s std::sort with the reference's comparator sort_by_decreasing_blockLen (HTS:1110-1113,
 * used at HTS:1368): sorts n u32 elements (block length + 1) << 25 | index, comparing the top 7 bits only, with ONE GPU
 * thread, as the table writer does for every list above 16 items.  forced_depth >= 0 replaces introsort's depth limit
 * 2 lg n (0 = heapsort at once), so that tests reach the heapsort branch. */
int cora_hip_test_list_sort(uint32_t *elems, uint64_t n, int forced_depth);
/* 2-bit pack + validity of the reference image of a job (after set_reference):
 * fills packed[ceil(G/32)] (base i of the padded global coordinate at bits
 * 62-2*(i%32) of word i/32), bad[ceil(G/32)] (bit 31-(i%32)), and contig_start[n_contigs]. */
int cora_hip_test_pack(cora_hip_job *job, uint64_t *packed, uint32_t *bad, uint64_t *contig_start, uint64_t *padded_len);
/* Canonical keys of all valid windows, in position order (after set_reference). */
int cora_hip_test_kmers(cora_hip_job *job, uint64_t *key_hi, uint64_t *key_lo, uint32_t *gpos, uint64_t cap, uint64_t *n_out);
/* "Is there a K-mer window at position g" (HTS:2619-2636 minus the case of HTS:1954-1981) for all positions, one bit each
 * (bit 31-(g%32) of word g/32), computed twice: fast[] by the 32-positions-at-once path the build uses, slow[] by the
 * per-position definition.  cap_words >= ceil(G/32) (n_words receives it). */
int cora_hip_test_window_bits(cora_hip_job *job, uint32_t *fast, uint32_t *slow, uint64_t cap_words, uint64_t *n_words);

#ifdef __cplusplus
}
#endif
#endif /* CORA_HIP_H */
