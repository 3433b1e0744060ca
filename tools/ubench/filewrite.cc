// tools/ubench/filewrite.cc -- how fast can SAM text go into ONE regular file?  (the text path's end-to-end ceiling)
//   a) one thread, write()                         b) N threads, pwrite() at their own offsets
//   c) N threads, memcpy into a MAP_SHARED mapping of the file (ftruncate'd first), blocks of 64 MiB dealt round-robin
//   g++ -O2 -pthread tools/ubench/filewrite.cc -o tools/ubench/filewrite ;  tools/ubench/filewrite <dir> <GiB> <threads>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 4.0) << 30, blk = 64u << 20;
    const int nt = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<char> src(blk);
    for (size_t i = 0; i < blk; ++i) src[i] = (char)('A' + i % 23);
    const std::string fn = dir + "/filewrite.tmp";
    auto run = [&](const char *name, auto body) {
        unlink(fn.c_str());
        const int fd = open(fn.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); exit(1); }
        const double t0 = now();
        body(fd);
        const double dt = now() - t0;
        close(fd);
        printf("%-46s %6.2f GB/s (%.2f s for %.1f GiB)\n", name, total / dt / 1e9, dt, total / 1073741824.0);
        fflush(stdout);
    };
    run("a) 1 thread write()", [&](int fd) { for (size_t o = 0; o < total; o += blk) if (write(fd, src.data(), blk) != (ssize_t)blk) { perror("write"); exit(1); } });
    char nm[96];
    snprintf(nm, sizeof nm, "b) %d threads pwrite()", nt);
    run(nm, [&](int fd) {
        std::atomic<size_t> next{ 0 };
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&] { for (;;) { const size_t o = next.fetch_add(blk); if (o >= total) break; if (pwrite(fd, src.data(), blk, (off_t)o) != (ssize_t)blk) { perror("pwrite"); exit(1); } } });
        for (auto &x : th) x.join();
    });
    for (int pre = 0; pre < 2; ++pre) {
        snprintf(nm, sizeof nm, "c) %d threads memcpy into mmap%s", nt, pre ? " (fallocate first)" : " (ftruncate)");
        run(nm, [&](int fd) {
            if (pre ? posix_fallocate(fd, 0, (off_t)total) : ftruncate(fd, (off_t)total)) { perror("size"); exit(1); }
            char *m = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); exit(1); }
            std::atomic<size_t> next{ 0 };
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&] { for (;;) { const size_t o = next.fetch_add(blk); if (o >= total) break; memcpy(m + o, src.data(), blk); } });
            for (auto &x : th) x.join();
            munmap(m, total);
        });
    }
    snprintf(nm, sizeof nm, "d) %d threads pwrite(), fallocate first", nt);
    run(nm, [&](int fd) {
        if (posix_fallocate(fd, 0, (off_t)total)) { perror("fallocate"); exit(1); }
        std::atomic<size_t> next{ 0 };
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&] { for (;;) { const size_t o = next.fetch_add(blk); if (o >= total) break; if (pwrite(fd, src.data(), blk, (off_t)o) != (ssize_t)blk) { perror("pwrite"); exit(1); } } });
        for (auto &x : th) x.join();
    });
    unlink(fn.c_str());
    return 0;
}
