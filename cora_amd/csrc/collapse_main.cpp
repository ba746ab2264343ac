// collapse_main.cpp -- drop-in `collapse` / `collapse_L` (-DCHR_SHORT) executable.
//
// Same sixteen arguments, output files and argument errors as the reference program
// (/root/reference/collapse.cpp:508-533, CL below; called by cora.cpp:1677-1691 and 1733-1760), for its whole-input run
// mode (argv[13] = NONE) and its fragmented mode (argv[13] = one group's signals, argv[1] = that group's fragment file).
// The collapsing itself runs on the MI355X through libcora_hip.so (cora_hip_collapse_*); this file parses the arguments
// and the file list, maps the inputs and writes the three results.  No CPU path.
// Differences by design: the process ends with status 0 instead of sending itself SIGINT (CL:2425-2427: a shortcut around slow
// deallocation; cora ignores the status, cora.cpp:1691) -- the kill-signal file (argv[15]) is still written;
// the reference's .fai and line length (argv[5]) are not needed: the FASTA is scanned on the GPU.
#include "../../include/cora_hip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#ifdef CHR_SHORT
static const int kChrShort = 1;
#else
static const int kChrShort = 0;
#endif

struct Mapped { const uint8_t *p = nullptr; size_t n = 0; };

static bool map_file(const std::string &path, Mapped &m)
{
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return false; }
    m.n = (size_t)sb.st_size;
    m.p = m.n ? (const uint8_t *)mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, fd, 0) : (const uint8_t *)"";
    close(fd);
    return m.p != MAP_FAILED;
}

static bool write_file(const char *path, const uint8_t *data, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = n == 0 || fwrite(data, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

int main(int argc, char **argv)
{
    if (argc != 17) {
        std::cout << "collapse (MI355X build): collapses the read-mers of a read dataset for coarse mapping\n"
                     "  1 FASTQ file list            2 digits per read-mer id      3 collapsed reads (out)\n"
                     "  4 reference FASTA | NOREF    5 reference line length       6 read length\n"
                     "  7 perfect matches (out)      8 unused                      9 long read names (out)\n"
                     " 10 SINGLE|PAIRED             11 FULL|HALF|THREEWAY         12 BWA|BWA_MEM|BOWTIE_2|other\n"
                     " 13 NONE | the signals of one fragment group (1 is then its fragment file)  14 fragment signal length\n"
                     " 15 kill signal file (out)    16 first long-read-name number - 1\n";
        return 10;                                                                                  // CL:512-533
    }
    cora_hip_collapse_params prm;
    memset(&prm, 0, sizeof prm);
    prm.lrn_offset = (uint64_t)atoi(argv[16]);
    const std::string frag = argv[13];
    const bool fragmented = frag != "NONE";
    const int frag_len = fragmented ? atoi(argv[14]) : 0;
    if (fragmented && frag_len != 2 && frag_len != 4) { std::cout << "fragSignalLen can only be 2 or 4" << std::endl; return 19; }   // CL:557-561 (exit 19, CL:560)
    prm.read_len = atoi(argv[6]);
    const std::string split = argv[11];
    if (split == "FULL") prm.splits = 1; else if (split == "HALF") prm.splits = 2; else if (split == "THREEWAY") prm.splits = 3;
    else { std::cout << "ERROR: Split mode unrecognized" << std::endl; return 9; }                  // CL:584-588
    const std::string mapper = argv[12];
    prm.mapper = mapper == "BWA" ? 1 : mapper == "BWA_MEM" ? 2 : mapper == "BOWTIE_2" ? 3 : 0;       // CL:590-611
    const std::string mode = argv[10];
    if (mode == "PAIRED") prm.paired = 1; else if (mode == "SINGLE") prm.paired = 0;
    else { std::cout << "ERROR: Input mode unrecognized" << std::endl; return 8; }                  // CL:624-628
    prm.digit_len = atoi(argv[2]);
    prm.chr_short = kChrShort;
    std::cout << "Total digit length of all read identities is: " << prm.digit_len << std::endl;

    int rc = cora_hip_init(0);
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    cora_hip_collapse *col = nullptr;
    rc = cora_hip_collapse_create(&prm, &col);
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }

    if (fragmented) {
        rc = cora_hip_collapse_set_fragment(col, argv[13], frag_len);
        if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    }
    const std::string ref = argv[4];
    if (ref != "NOREF") {
        Mapped m;
        if (!map_file(ref, m)) { std::cout << "ERROR: Could not find reference file: " << ref << std::endl; return 50; }
        cora_hip_fasta *fa = nullptr;
        rc = cora_hip_fasta_scan(m.p, m.n, &fa);
        if (!rc) rc = cora_hip_collapse_set_reference(col, fa);
        if (fa) cora_hip_fasta_destroy(fa);
        if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    }

    if (fragmented) {                                                                              // CL:2173-2184: argv[1] is the fragment file itself
        Mapped a;
        if (!map_file(argv[1], a)) { std::cout << "Binary input file: " << argv[1] << " doesn't exist" << std::endl; return 5; }
        rc = cora_hip_collapse_add_reads(col, a.p, a.n);
        if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
        if (a.n) munmap((void *)a.p, a.n);
    } else
    // the list: number of samples, then per sample the reads file [the mates file] and a read count (CL:861-912)
    {
        std::ifstream list(argv[1]);
        int n_samples = 0;
        list >> n_samples;
        if (n_samples > 65535) { std::cout << "This version will crash if the number of samples is greater than 65535" << std::endl; return 28; }
        std::cout << "number of Samples: " << n_samples << std::endl;
        std::string f1, f2;
        long long count_unused;
        while (list >> f1) {
            Mapped a, b;
            if (!map_file(f1, a)) { std::cout << "Input fastq file: " << f1 << " doesn't exist" << std::endl; return 5; }
            if (prm.paired) {
                list >> f2 >> count_unused;
                if (!map_file(f2, b)) { std::cout << "Read mate input file: " << f2 << " doesn't exist" << std::endl; return 6; }
            } else list >> count_unused;
            rc = cora_hip_collapse_add_reads(col, a.p, a.n);
            if (!rc && prm.paired) rc = cora_hip_collapse_add_reads(col, b.p, b.n);
            if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
            if (a.n) munmap((void *)a.p, a.n);
            if (prm.paired && b.n) munmap((void *)b.p, b.n);
        }
    }

    rc = cora_hip_collapse_run(col);
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    const uint8_t *r, *p, *l;
    size_t rn, pn, ln;
    cora_hip_collapse_outputs(col, &r, &rn, &p, &pn, &l, &ln);
    if (!write_file(argv[3], r, rn) || !write_file(argv[7], p, pn) || !write_file(argv[9], l, ln)) {
        std::cout << "ERROR: Can't write the output files" << std::endl;
        return 68;
    }
    cora_hip_collapse_stats st;
    cora_hip_collapse_get_stats(col, &st);
    std::cout << "numTotal: " << st.printed << std::endl;                                          // CL:2409
    std::cout << "read-mers " << st.read_mers << "  reference windows " << st.ref_windows << "  distinct " << st.distinct
              << "  long names " << st.lrn_names << "\nms: parse " << st.ms_parse << "  keys " << st.ms_keys << "  group " << st.ms_group
              << "  order " << st.ms_order << "  text " << st.ms_text << "  total " << st.ms_total << std::endl;
    { std::ofstream kill(argv[15]); kill << "Y" << std::endl; }                                    // CL:2425-2426
    cora_hip_collapse_destroy(col);
    cora_hip_shutdown();
    return 0;
}
