// fastqsplit_main.cpp -- drop-in `fastqSplit` executable.
//
// Same five arguments, fragment files (<prefix>.frag<i>) and argument errors as the reference program
// (/root/reference/fastqSplit.cpp:316-347, FS below; called by cora.cpp:1696-1730).  The partitioning runs on the MI355X
// through libcora_hip.so (cora_hip_fqsplit_*); this file parses the arguments, the signals file and the file list, maps
// the inputs and writes the fragment files.  No CPU path.
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
#include <sstream>
#include <string>
#include <vector>

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

static int die(int rc)
{
    std::cout << "ERROR: " << cora_hip_last_error() << std::endl;
    return rc;
}

int main(int argc, char **argv)
{
    if (argc != 6) {                                                               // FS:318-327
        std::cout << "fastqSplit (MI355X build): splits a set of FASTQ files into fragmented binary read-mer files\n"
                     "ARGV[1] = fragmentation signals\nARGV[2] = readLen\nARGV[3] = Input FASTQ file list\n"
                     "ARGV[4] = Output prefix name (*.frag0 ....)\n"
                     "ARGV[5] = mapping mode: SINGLE, SPLIT_SINGLE, PAIRED, SPLIT_PAIRED, THREEWAY_SINGLE, THREEWAY_PAIRED" << std::endl;
        return 1;
    }
    cora_hip_fqsplit_params prm;
    memset(&prm, 0, sizeof prm);
    prm.read_len = atoi(argv[2]);
    const std::string mode(argv[5]);
    if (mode == "SINGLE") { prm.splits = 1; prm.paired = 0; }
    else if (mode == "SPLIT_SINGLE") { prm.splits = 2; prm.paired = 0; }
    else if (mode == "THREEWAY_SINGLE") { prm.splits = 3; prm.paired = 0; }
    else if (mode == "PAIRED") { prm.splits = 1; prm.paired = 1; }
    else if (mode == "SPLIT_PAIRED") { prm.splits = 2; prm.paired = 1; }
    else if (mode == "THREEWAY_PAIRED") { prm.splits = 3; prm.paired = 1; }
    else { std::cout << "ERROR: Mapping mode undefined... " << std::endl; return 5; }   // FS:600-604, 851-855

    Mapped sig;
    if (!map_file(argv[1], sig)) { std::cout << "ERROR: cannot read the fragmentation signals: " << argv[1] << std::endl; return 66; }
    if (int rc = cora_hip_fqsplit_parse_signals((const char *)sig.p, sig.n, &prm)) return die(rc);

    std::ifstream lst(argv[3]);
    int samples = 0;
    if (!(lst >> samples)) { std::cout << "ERROR: cannot read the FASTQ list: " << argv[3] << std::endl; return 66; }
    if (cora_hip_init(0)) return die(102);
    cora_hip_fqsplit *h = nullptr;
    if (int rc = cora_hip_fqsplit_create(&prm, &h)) return die(rc);
    for (int k = 0; k < samples; k++) {
        std::string f1, f2;
        long long count;
        if (prm.paired ? !(lst >> f1 >> f2 >> count) : !(lst >> f1 >> count)) { std::cout << "ERROR: FASTQ list ends early" << std::endl; return 66; }
        Mapped m;
        if (!map_file(f1, m)) { std::cout << "ERROR: cannot read " << f1 << std::endl; return 67; }
        if (int rc = cora_hip_fqsplit_add_reads(h, m.p, m.n)) return die(rc);
        if (m.n) munmap((void *)m.p, m.n);
        if (prm.paired) {
            if (!map_file(f2, m)) { std::cout << "ERROR: cannot read " << f2 << std::endl; return 67; }
            if (int rc = cora_hip_fqsplit_add_reads(h, m.p, m.n)) return die(rc);
            if (m.n) munmap((void *)m.p, m.n);
        }
    }
    if (int rc = cora_hip_fqsplit_run(h)) return die(rc);
    for (int g = 0; g < prm.n_groups; g++) {                                       // FS:341-347, 862-865
        const uint8_t *bytes; size_t n;
        if (int rc = cora_hip_fqsplit_group(h, g, &bytes, &n)) return die(rc);
        std::ostringstream name;
        name << argv[4] << ".frag" << g;
        FILE *f = fopen(name.str().c_str(), "wb");
        if (!f || (n && fwrite(bytes, 1, n, f) != n) || fclose(f) != 0) { std::cout << "ERROR: cannot write " << name.str() << std::endl; return 68; }
    }
    cora_hip_fqsplit_destroy(h);
    return 0;
}
