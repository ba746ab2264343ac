// io_probe.cpp -- how fast the GPU box's temporary directory takes the <Inexact> table (894 MB) and hands back the FASTA image (3.1 GB):
// round 3's choices for the executable's file I/O rest on these numbers.   io_probe <dir>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t n = (size_t)894 << 20;
    char *src = (char *)mmap(0, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(src, 7, n);
    {
        double t = now();
        FILE *f = fopen((dir + "/probe_f").c_str(), "wb");
        if (!f || fwrite(src, 1, n, f) != n) return 1;
        fclose(f);
        printf("write 894 MB: fwrite %.1f ms\n", now() - t);
    }
    for (int nt : {1, 4, 8}) {
        double t = now();
        int fd = open((dir + "/probe_m").c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || ftruncate(fd, n)) return 1;
        char *p = (char *)mmap(0, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double t1 = now();
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++) th.emplace_back([=] { for (size_t o = n * k / nt; o < n * (k + 1) / nt; o += 4096) p[o] = 0; });
        for (auto &x : th) x.join();
        double t2 = now();
        std::vector<std::thread> th2;
        for (int k = 0; k < nt; k++) th2.emplace_back([=] { memcpy(p + n * k / nt, src + n * k / nt, n * (k + 1) / nt - n * k / nt); });
        for (auto &x : th2) x.join();
        double t3 = now();
        munmap(p, n);
        if (ftruncate(fd, n - 12345)) return 1;
        close(fd);
        printf("write 894 MB: shared mapping, %d threads: set-up %.1f, first touch %.1f, copy into touched pages %.1f, unmap + truncate + close %.1f ms\n", nt, t1 - t,
               t2 - t1, t3 - t2, now() - t3);
    }
    for (int nt : {1, 4}) {
        double t = now();
        int fd = open((dir + "/probe_p").c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++)
            th.emplace_back([=] { size_t a = n * k / nt, b = n * (k + 1) / nt; while (a < b) { ssize_t w = pwrite(fd, src + a, std::min((size_t)(8 << 20), b - a), a); if (w <= 0) break; a += w; } });
        for (auto &x : th) x.join();
        close(fd);
        printf("write 894 MB: pwrite, %d threads %.1f ms\n", nt, now() - t);
    }
    // read back 3 x the file as one 2.7 GB "image": just written (not synced), then again
    const std::string big = dir + "/probe_big";
    {
        FILE *f = fopen(big.c_str(), "wb");
        for (int k = 0; k < 3; k++) if (fwrite(src, 1, n, f) != n) return 1;
        fclose(f);
    }
    const size_t N = 3 * n, CH = 8u << 20;
    char *dst = (char *)mmap(0, (size_t)16 * 2 * CH, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(dst, 1, (size_t)16 * 2 * CH);
    for (int round = 0; round < 2; round++)
        for (int mode = 0; mode < 2; mode++)
            for (int nt : {6, 12}) {
                int fd = open(big.c_str(), O_RDONLY);
                const char *img = mode == 0 ? (const char *)mmap(0, N, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
                double t = now();
                std::vector<std::thread> th;
                for (int k = 0; k < nt; k++)
                    th.emplace_back([=] {
                        char *buf = dst + (size_t)k * 2 * CH;
                        int b = 0;
                        for (size_t c = k; c * CH < N; c += nt, b ^= 1) {
                            const size_t off = c * CH, len = std::min(CH, N - off);
                            if (mode == 0) memcpy(buf + b * CH, img + off, len);
                            else { size_t a = 0; while (a < len) { ssize_t r = pread(fd, buf + b * CH + a, len - a, off + a); if (r <= 0) break; a += r; } }
                        }
                    });
                for (auto &x : th) x.join();
                printf("read 2.7 GB (%s) into staging buffers: %s, %d threads: %.1f ms = %.1f GB/s\n", round ? "second time" : "just written",
                       mode == 0 ? "memcpy from a private mapping" : "pread", nt, now() - t, N / (now() - t) / 1e6);
                if (img) munmap((void *)img, N);
                close(fd);
            }
    for (const char *f : {"/probe_f", "/probe_m", "/probe_p", "/probe_big"}) unlink((dir + f).c_str());
    return 0;
}
