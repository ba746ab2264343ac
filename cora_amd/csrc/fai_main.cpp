// fai_main.cpp -- `cora_faiGenerate <ref.fa>`: writes <ref.fa>.fai, the file `cora faiGenerate <ref.fa>` produces
// (/root/reference/cora.cpp:1006-1029 runs an awk program through system() for it).  The index is computed on the
// MI355X by libcora_hip.so (cora_hip_fasta_scan); this file only maps the FASTA and writes the text.
// Exit codes: 0 ok, 19 usage, 50 unreadable reference (cora's VerifyRefFile), 68 cannot write, others from the library.
#include "../../include/cora_hip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <iostream>
#include <string>

int main(int argc, char **argv)
{
    if (argc != 2 || std::string(argv[1]) == "--help" || std::string(argv[1]) == "help") {
        std::cout << "cora_faiGenerate <Reference_File>   (writes <Reference_File>.fai; MI355X build of `cora faiGenerate`)\n";
        return 19;
    }
    const std::string ref = argv[1];
    int fd = open(ref.c_str(), O_RDONLY);
    if (fd < 0) { std::cout << "ERROR: Could not find reference file: " << ref << std::endl; return 50; }
    struct stat sb;
    fstat(fd, &sb);
    const size_t n = (size_t)sb.st_size;
    const uint8_t *image = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : (const uint8_t *)"";
    if (image == MAP_FAILED) { std::cout << "ERROR: cannot map " << ref << std::endl; return 50; }
    int rc = cora_hip_init(0);
    cora_hip_fasta *fa = nullptr;
    if (!rc) rc = cora_hip_fasta_scan(image, n, &fa);
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    const char *text; size_t bytes;
    cora_hip_fasta_fai_text(fa, &text, &bytes);
    FILE *f = fopen((ref + ".fai").c_str(), "wb");
    const bool ok = f && (bytes == 0 || fwrite(text, 1, bytes, f) == bytes);
    if (!f || fclose(f) != 0 || !ok) { std::cout << "ERROR: Can't write to file: " << ref << ".fai" << std::endl; return 68; }
    cora_hip_fasta_info in;
    cora_hip_fasta_get_info(fa, &in);
    std::cout << in.contigs << " contigs, " << in.bases << " bases, " << in.lines << " lines; scan " << in.ms_scan << " ms"
              << (in.conforming ? "" : "  (not in fixed-line-width layout)") << std::endl;
    cora_hip_fasta_destroy(fa);
    cora_hip_shutdown();
    return 0;
}
