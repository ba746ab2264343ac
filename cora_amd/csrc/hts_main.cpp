// hts_main.cpp -- drop-in `homTable_setup` / `homTable_setup_L` (-DCHR_SHORT) executable.
//
// Same command line, exit codes and output files as the reference program
// (/root/reference/homTable_setup.cpp:2389-2405, called by cora.cpp:1586), so `cora coraIndex`
// can run it unmodified.  All table construction happens on the MI355X through libcora_hip.so
// (include/cora_hip.h); this file only parses the arguments and the split files, maps the FASTA
// and writes the results.  There is no CPU path: without a gfx950 device it exits with an error.
// HTS = /root/reference/homTable_setup.cpp.
#include "../../include/cora_hip.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#ifdef CHR_SHORT
static const int kChrShort = 1;
#else
static const int kChrShort = 0;
#endif

static int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

static std::string revcomp(const std::string &s)
{
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) c = (c == 'A') ? 'T' : (c == 'C') ? 'G' : (c == 'G') ? 'C' : (c == 'T') ? 'A' : c;
    return r;
}

// A signal token and its reverse complement belong to the line's group (HTS:2295-2324).
static bool add_signal(const std::string &tok, size_t want_len, int group, uint8_t *table)
{
    if (tok.size() != want_len) return false;
    for (const std::string &t : {tok, revcomp(tok)}) {
        int idx = 0;
        for (char c : t) { int b = base_code(c); if (b < 0) return false; idx = idx * 4 + b; }
        table[idx] = (uint8_t)group;
    }
    return true;
}

static void make_parent_dir(const std::string &path)          // HTS:2365-2387
{
    size_t slash = path.rfind('/');
    if (slash == std::string::npos) return;
    std::string dir = path.substr(0, slash + 1);
    struct stat st;
    if (stat(dir.c_str(), &st) == -1) mkdir(dir.c_str(), 0700);
}

static bool write_file(const std::string &path, const void *data, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = n == 0 || fwrite(data, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

static double ms_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv)
{
    const auto t_start = std::chrono::steady_clock::now();
    if (argc != 10) {
        std::cout << "homTable_setup (MI355X build): exact and inexact homology tables of a reference\n"
                     "  1 reference FASTA (with .fai)   2 read-mer length [33-64]   3 2-char split file\n"
                     "  4 exact table (out)             5 inexact table (out)       6 4-char split file\n"
                     "  7 BOTH                          8 mismatches [1-3]          9 FULL\n"
                     "  (EXACT alone and the ONLY_CONSTRUCT / ONLY_*COMPACT run modes of the reference, which stop at or start from its\n"
                     "   internal pre-compact files, are not built: cora never issues them)\n";
        return 19;
    }
    const std::string ref_path = argv[1], exact_out = argv[4], inexact_out = argv[5];
    const std::string what = argv[7], run_mode = argv[9];

    cora_hip_params prm;
    memset(&prm, 0, sizeof prm);
    prm.K = atoi(argv[2]);
    if (prm.K < 33 || prm.K > 64) {
        std::cout << "This homology table constructor works for 33-64bp read-mers" << std::endl;
        return 65;
    }
    prm.d = atoi(argv[8]);
    prm.chr_short = kChrShort;
    make_parent_dir(exact_out);

    // 2-char (physical) splits: first line the count, then one group per line (HTS:2431-2442, 2499-2509)
    {
        std::ifstream f(argv[3]);
        if (!f.is_open()) { std::cout << "ERROR: Can't open file: " << argv[3] << std::endl; return 66; }
        std::string line;
        std::getline(f, line);
        prm.n_groups = atoi(line.c_str());
        int g = 0;
        while (std::getline(f, line)) {
            g++;
            std::stringstream ss(line);
            std::string tok;
            int cnt = 0;
            while (ss >> tok) { if (!add_signal(tok, 2, g, prm.sig2)) { std::cout << "ERROR: bad 2-char signal '" << tok << "'" << std::endl; return 66; } cnt++; }
            if (cnt == 0) { std::cout << "ERROR: empty 2-char signal group" << std::endl; return 66; }
        }
        if (g != prm.n_groups || g > 255) { std::cout << "ERROR: 2-char split file announces " << prm.n_groups << " groups, holds " << g << std::endl; return 66; }
    }
    // 4-char (parallel) splits (HTS:2327-2346)
    {
        std::ifstream f(argv[6]);
        if (!f.is_open()) { std::cout << "ERROR: can't open file: " << argv[6] << std::endl; return 67; }
        std::string line;
        std::getline(f, line);
        prm.n_threads = atoi(line.c_str());
        if (prm.n_threads < 1 || prm.n_threads > 255) { std::cout << "ERROR: bad 4-char split count" << std::endl; return 67; }
        for (int t = 1; t <= prm.n_threads; t++) {
            if (!std::getline(f, line)) break;
            std::stringstream ss(line);
            std::string tok;
            while (ss >> tok) if (!add_signal(tok, 4, t, prm.sig4)) { std::cout << "ERROR: bad 4-char signal '" << tok << "'" << std::endl; return 67; }
        }
    }
    make_parent_dir(inexact_out);

    if (what != "BOTH" || run_mode != "FULL") {
        std::cout << "ERROR: this build runs 'BOTH ... FULL' (the only form cora issues, cora.cpp:1581-1587), not EXACT or the ONLY_* run modes; got '" << what << "' '"
                  << run_mode << "'" << std::endl;
        return 19;
    }
    if (prm.d < 1 || prm.d > 3) {
        std::cout << "ERROR: Number of mismatches for inexact homology needs to be between 1 and 3" << std::endl;
        return 6;
    }

    // reference: line width = length of the file's 2nd token (HTS:155-166); contigs from .fai columns 1-2 (HTS:171-197)
    int fd = open(ref_path.c_str(), O_RDONLY);
    if (fd < 0) { std::cout << "ERROR: Could not find reference file: " << ref_path << std::endl; return 50; }
    struct stat sb;
    fstat(fd, &sb);
    const size_t image_bytes = (size_t)sb.st_size;
    const uint8_t *image = (const uint8_t *)mmap(nullptr, image_bytes ? image_bytes : 1, PROT_READ, MAP_PRIVATE, fd, 0);
    if (image == MAP_FAILED || image_bytes == 0 || image[0] != '>') {
        std::cout << "ERROR: Reference file is not properly formatted..." << std::endl;
        return 51;
    }
    // contig names and lengths as the .fai gives them (HTS:171-197; exit 52 without it)
    std::vector<long long> fai_len;
    {
        std::ifstream fai((ref_path + ".fai").c_str());
        if (!fai.is_open()) { std::cout << "Could not find fai file for reference file: " << ref_path << std::endl; return 52; }
        std::string line;
        while (std::getline(fai, line)) {
            std::stringstream ss(line);
            std::string name;
            long long size = 0;
            ss >> name >> size;
            if (name.empty()) continue;
            fai_len.push_back(size);
        }
    }
    if (fai_len.empty()) { std::cout << "ERROR: empty .fai" << std::endl; return 52; }

    const double ms_parse = ms_since(t_start);
    int rc = cora_hip_init(0);
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    const double ms_init = ms_since(t_start);
    // The image is scanned on the GPU (lines, headers, contig lengths: cora_hip_fasta_scan).  A file in the layout
    // ReadReference assumes (HTS:96-142: one line width) is read in place; any other is gathered first, where the
    // reference would silently mis-load it.  The .fai must describe the same contigs.
    cora_hip_fasta *fa = nullptr;
    if (getenv("CORA_TRACE")) std::cerr << "[cora] scan starts at " << ms_since(t_start) << " ms" << std::endl;
    rc = cora_hip_fasta_scan(image, image_bytes, &fa);
    if (getenv("CORA_TRACE")) std::cerr << "[cora] scan returned at " << ms_since(t_start) << " ms" << std::endl;
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    cora_hip_fasta_info fin;
    cora_hip_fasta_get_info(fa, &fin);
    std::cout << "Reference line length: " << fin.line_width << "bp .." << (fin.conforming ? "" : " (lines of other lengths present)") << std::endl;
    if (fin.contigs != fai_len.size()) {
        std::cout << "ERROR: the .fai lists " << fai_len.size() << " contigs, the reference holds " << fin.contigs << std::endl;
        return 51;
    }
    for (size_t c = 0; c < fai_len.size(); c++) {
        uint64_t len = 0;
        cora_hip_fasta_contig(fa, (int32_t)c, nullptr, &len, nullptr, nullptr);
        if ((long long)len != fai_len[c]) {
            std::cout << "ERROR: contig " << c + 1 << " has " << len << " bases, the .fai says " << fai_len[c] << std::endl;
            return 51;
        }
    }
    prm.line_width = 1;                                      // unused on this path
    cora_hip_job *job = nullptr;
    const double ms_scan = ms_since(t_start);
    rc = cora_hip_job_create(&prm, &job);
    if (!rc) rc = cora_hip_job_set_reference_fasta(job, fa);
    if (!rc) rc = cora_hip_job_set_table_files(job, exact_out.c_str(), inexact_out.c_str());
    const double ms_setref = ms_since(t_start);
    // (the scan object -- 7 GB of device arrays for a 3.1 Gbp reference -- is given back after the tables are written, not here:
    //  a release synchronises the device and can be slow on a fresh box; 288 GB hold the build beside it)
    if (getenv("CORA_TRACE")) std::cerr << "[cora] scan done at " << ms_scan << " ms, reference set at " << ms_setref << std::endl;
    const double ms_load = ms_since(t_start);
    // Every GPU of the node takes part (cora starts ONE process, cora.cpp:1586): one host thread per GPU, records exchanged
    // peer to peer.  A reference below 256 Mbp is built on one GPU (the exchanges would cost more than they save).
    // CORA_GPUS=n limits the GPUs used; CORA_RANKS=n forces n ranks, dealt over the GPUs in use (n > GPUs: ranks share a
    // GPU -- the rehearsal the tests run on a one-GPU box).
    int n_gpus = 1, ranks = 1;
    if (!rc) rc = cora_hip_device_count(&n_gpus);
    if (const char *e = getenv("CORA_GPUS")) { const int v = atoi(e); if (v >= 1 && v < n_gpus) n_gpus = v; }
    unsigned long long total_bases = 0;
    for (long long l : fai_len) total_bases += (unsigned long long)l;
    ranks = total_bases >= 256000000ull ? n_gpus : 1;
    if (const char *e = getenv("CORA_RANKS")) { const int v = atoi(e); if (v >= 1 && v <= 64) ranks = v; }
    if (!rc) {
        if (ranks > 1) {
            std::vector<int> devs(ranks);
            for (int r = 0; r < ranks; r++) devs[r] = r % n_gpus;
            std::cout << "Building on " << ranks << " ranks over " << (ranks < n_gpus ? ranks : n_gpus) << " GPU(s)" << std::endl;
            rc = cora_hip_job_run_multi(job, ranks, devs.data());
            if (rc) {
                // The peer-to-peer flow has only ever run with all ranks on ONE device: whatever stops it
                // on a real multi-GPU node (peer access, memory, a link) must not fail `cora coraIndex` -- one GPU builds the
                // same tables.  CORA_GPUS=1 skips the attempt.
                std::cout << "WARNING: the build over " << ranks << " ranks failed (" << cora_hip_last_error() << "); falling back to one GPU" << std::endl;
                rc = cora_hip_job_run(job);
            }
        } else rc = cora_hip_job_run(job);
    }
    if (rc) { std::cout << "ERROR: " << cora_hip_last_error() << std::endl; return rc; }
    const double ms_build = ms_since(t_start);

    const uint8_t *ex, *inx;
    size_t exb, inb;
    uint64_t exn, inn;
    cora_hip_job_tables(job, &ex, &exb, &exn, &inx, &inb, &inn);
    std::string cnt;
    // both tables are in their files already (cora_hip_job_set_table_files above): <Exact> written beside the inexact stages,
    // <Inexact> assembled in a mapping of its file; this cuts <Inexact> to size, closes both and reports failed writes
    bool ok = cora_hip_job_finish_table_files(job) == 0;
    cnt = std::to_string(exn) + "\n";
    ok = ok && write_file(exact_out + ".itemCount", cnt.data(), cnt.size());                    // HTS:663-665
    cnt = std::to_string(inn) + "\n";
    ok = ok && write_file(inexact_out + ".itemCount", cnt.data(), cnt.size());                  // HTS:1634
    // cora removes the reference's intermediates afterwards (cora.cpp:1590-1595): leave empty ones
    ok = ok && write_file(exact_out + "_preCompact_toDelete", "", 0);
    for (int t = 1; t <= prm.n_threads; t++) ok = ok && write_file(inexact_out + "_preCompact_toDelete_t" + std::to_string(t), "", 0);
    if (!ok) { std::cout << "ERROR: Can't write to file: " << exact_out << " / " << inexact_out << std::endl; return 68; }

    cora_hip_stats st;
    cora_hip_job_stats(job, &st);
    std::cout << "k-mers " << st.kmers << "  representatives " << st.representatives << "  exact nodes " << st.exact_nodes
              << "  candidate pairs " << st.candidate_pairs << "  homology pairs " << st.hom_pairs << "  inexact nodes "
              << st.inexact_nodes << "  (tied lists " << st.dirty_nodes << ", lists > 16 items " << st.big_nodes << ")\n"
              << "ms: pack+keys " << st.ms_pack << "  exact " << st.ms_exact << "  seeds " << st.ms_seeds << "  items " << st.ms_items
              << "  host " << st.ms_host_finish << "  total " << st.ms_total << std::endl;
    // process start -> the four files closed (SURVEY.md section 8d's definition of the end-to-end time), cumulative marks
    std::cout << "e2e_ms: args+splits+mmap " << ms_parse << "  device_init " << ms_init - ms_parse << "  read+upload+scan " << ms_load - ms_init
              << "  build(incl. tables to host) " << ms_build - ms_load << "  write_files " << ms_since(t_start) - ms_build
              << "  total " << ms_since(t_start) << std::endl;
    cora_hip_fasta_destroy(fa);
    cora_hip_job_destroy(job);
    if (getenv("CORA_TRACE")) std::cerr << "[cora] job destroyed at " << ms_since(t_start) << " ms" << std::endl;
    cora_hip_shutdown();
    if (getenv("CORA_TRACE")) std::cerr << "[cora] shut down at " << ms_since(t_start) << " ms" << std::endl;
    // (Leaving through _exit here, to skip the HIP / HSA runtime's own teardown, makes the process take 1.08 instead of 0.89 s: the
    //  kernel driver then tears the process's GPU state down itself, which is the slower way -- measured with profiles/e2e_variants.py.)
    return 0;
}
