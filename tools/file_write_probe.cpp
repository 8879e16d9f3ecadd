// file_write_probe.cpp -- how fast can ONE regular file take bytes from memory on this box?
// `SVDSS smooth` writes ~16 GB per million reads to its stdout; four pwrite threads reach ~7 GB/s (the inode lock
// serialises buffered writes).  Modes: pwrite by T threads (chunks of 64 MB dealt round-robin), a shared mapping of the
// file written by T threads (page faults are per page, not per inode), O_DIRECT pwrite by T threads.
//   g++ -O2 -pthread -o file_write_probe file_write_probe.cpp;  ./file_write_probe DIR GIB
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

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const size_t total = (size_t)(argc > 2 ? atof(argv[2]) : 8.0) << 30;
  const size_t chunk = (size_t)64 << 20;
  const size_t n_chunks = total / chunk;
  // source: 1 GiB of page-aligned memory, touched (the product writes from page-locked buffers)
  const size_t src_n = (size_t)1 << 30;
  uint8_t* src = (uint8_t*)aligned_alloc(4096, src_n);
  for (size_t i = 0; i < src_n; i += 8) *(uint64_t*)(src + i) = i * 0x9e3779b97f4a7c15ull;
  const std::string path = dir + "/file_write_probe.bin";
  auto run = [&](const char* what, int T, int mode) {
    unlink(path.c_str());
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC | (mode == 2 ? O_DIRECT : 0), 0644);
    if (fd < 0) { printf("%-28s T=%2d: open failed\n", what, T); return; }
    const double t0 = now();
    if (mode == 1 || mode == 3) { if (ftruncate(fd, (off_t)total) != 0) { printf("ftruncate failed\n"); close(fd); return; } }
    if (mode == 3 || mode == 4) { if (posix_fallocate(fd, 0, (off_t)total) != 0) printf("(fallocate failed) "); }
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&] {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= n_chunks) return;
          const uint8_t* p = src + (k * chunk) % src_n;
          if (mode == 1 || mode == 3) {
            void* m = mmap(nullptr, chunk, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(k * chunk));
            if (m == MAP_FAILED) { bad = 1; return; }
            memcpy(m, p, chunk);
            munmap(m, chunk);
          } else {
            size_t n = chunk; off_t at = (off_t)(k * chunk);
            while (n) { const ssize_t w = pwrite(fd, p, n, at); if (w <= 0) { bad = 1; return; } p += w; n -= (size_t)w; at += w; }
          }
        }
      });
    for (auto& x : th) x.join();
    const double t1 = now();
    close(fd);
    printf("%-28s T=%2d: %6.2f GB/s%s\n", what, T, (double)total / (t1 - t0) * 1e-9, bad ? "  (FAILED)" : "");
    fflush(stdout);
  };
  printf("# %s, %.1f GiB per run\n", dir.c_str(), (double)total / (1 << 30));
  for (int T : {1, 4, 8}) run("pwrite", T, 0);
  for (int T : {4, 8, 16}) run("mmap (ftruncate first)", T, 1);
  for (int T : {4, 8, 16}) run("mmap (fallocate first)", T, 3);
  for (int T : {4, 8}) run("pwrite (fallocate first)", T, 4);
  for (int T : {4, 8}) run("pwrite O_DIRECT", T, 2);
  unlink(path.c_str());
  return 0;
}
