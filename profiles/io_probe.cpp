// What the file system of a box gives the deviceingest=t pipeline of bbduk_cli: page-cache read and write rates against thread count.
//   g++ -O2 -pthread profiles/io_probe.cpp -o /tmp/io_probe && /tmp/io_probe /tmp/io_probe.dat 8
// (GiB to use, default 8).  One JSON line.  Reads use pread into malloc'ed memory touched beforehand; writes use pwrite (one file, disjoint ranges),
// then O_DIRECT pwrite, then mmap + memcpy (one file, disjoint ranges) since buffered writes to one file take the inode lock on most file systems.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double timed(int nt, F f) { const double t0 = now(); std::vector<std::thread> th; for (int i = 0; i < nt; i++) th.emplace_back(f, i); for (auto& t : th) t.join(); return now() - t0; }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/io_probe.dat";
    const int64_t total = (argc > 2 ? atoll(argv[2]) : 8) << 30, piece = 64 << 20;
    uint8_t* buf = (uint8_t*)malloc((size_t)piece * 16);
    memset(buf, 'A', (size_t)piece * 16);
    printf("{\"bytes\": %lld", (long long)total);
    for (int nt : {1, 4, 8}) {                                     // pwrite
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        const double s = timed(nt, [&](int me) { for (int64_t off = (int64_t)me * piece; off < total; off += (int64_t)nt * piece) if (pwrite(fd, buf + (size_t)me * piece, piece, off) != piece) abort(); });
        close(fd);
        printf(", \"pwrite_%d_GBps\": %.2f", nt, total / s / 1e9);
    }
    {                                                              // O_DIRECT pwrite (page-aligned source, as pinned host memory is)
        void* ab = nullptr;
        if (posix_memalign(&ab, 4096, (size_t)piece * 8) == 0) {
            memset(ab, 'A', (size_t)piece * 8);
            for (int nt : {1, 4, 8}) {
                const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_DIRECT, 0644);
                if (fd < 0) { printf(", \"odirect\": \"not supported\""); break; }
                bool bad = false;
                const double s = timed(nt, [&](int me) { for (int64_t off = (int64_t)me * piece; off < total; off += (int64_t)nt * piece) if (pwrite(fd, (uint8_t*)ab + (size_t)me * piece, piece, off) != piece) { bad = true; return; } });
                close(fd);
                if (bad) { printf(", \"odirect_pwrite_%d_GBps\": null", nt); } else printf(", \"odirect_pwrite_%d_GBps\": %.2f", nt, total / s / 1e9);
            }
            free(ab);
        }
    }
    for (int nt : {4, 8, 16}) {                                    // mmap + memcpy into a file grown by ftruncate
        const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        const double t0 = now();
        if (ftruncate(fd, total) != 0) abort();
        uint8_t* m = (uint8_t*)mmap(nullptr, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { printf(", \"mmap\": \"failed\""); close(fd); break; }
        timed(nt, [&](int me) { for (int64_t off = (int64_t)me * piece; off < total; off += (int64_t)nt * piece) memcpy(m + off, buf + (size_t)me * piece, piece); });
        munmap(m, (size_t)total); close(fd);
        printf(", \"mmap_write_%d_GBps\": %.2f", nt, total / (now() - t0) / 1e9);
    }
    for (int nt : {1, 4, 8, 16}) {                                 // pread from the page cache
        const int fd = open(path, O_RDONLY);
        const double s = timed(nt, [&](int me) { for (int64_t off = (int64_t)me * piece; off < total; off += (int64_t)nt * piece) if (pread(fd, buf + (size_t)me * piece, piece, off) != piece) abort(); });
        close(fd);
        printf(", \"pread_%d_GBps\": %.2f", nt, total / s / 1e9);
    }
    printf("}\n");
    unlink(path);
    return 0;
}
