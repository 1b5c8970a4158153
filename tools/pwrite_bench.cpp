// Can the output FASTQ be written faster than one write() stream?  1 GiB to one file: pwrite from N threads to disjoint ranges, and
// memcpy from N threads into a shared mapping.  On the GPU box (ext4): pwrite 10.6 / 8.4 / 13.3 GB/s with 1 / 4 / 16 threads, mmap
// 10.7 / 10.5 / 3.9 GB/s: neither scales (inode lock / page-cache tree), so bz_assign_files keeps its single ordered writer thread.
// g++ -O2 -pthread -o /tmp/pwrite_bench tools/pwrite_bench.cpp
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/pw_test.bin";
    const size_t total = (size_t)1 << 30, piece = (size_t)2 << 20;
    std::vector<char> src(piece, 'A');
    struct statfs sf; statfs("/tmp", &sf); printf("f_type 0x%lx\n", (long)sf.f_type);
    for (int threads : {1, 4, 16}) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back([&, t] {
            for (size_t off = (size_t)t * piece; off < total; off += (size_t)threads * piece) if (pwrite(fd, src.data(), piece, off) != (ssize_t)piece) perror("pwrite");
        });
        for (auto &x : ts) x.join();
        double dt = now() - t0; close(fd);
        printf("pwrite %2d threads: %.3f s  %.1f GB/s\n", threads, dt, total / dt / 1e9);
    }
    for (int threads : {1, 4, 16}) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (ftruncate(fd, total) != 0) perror("ftruncate");
        char *m = (char *)mmap(nullptr, total, PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back([&, t] {
            for (size_t off = (size_t)t * piece; off < total; off += (size_t)threads * piece) memcpy(m + off, src.data(), piece);
        });
        for (auto &x : ts) x.join();
        munmap(m, total);
        double dt = now() - t0; close(fd);
        printf("mmap   %2d threads: %.3f s  %.1f GB/s\n", threads, dt, total / dt / 1e9);
    }
    unlink(path);
}
