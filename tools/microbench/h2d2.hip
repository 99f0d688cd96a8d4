// h2d2.hip -- costs around the text upload: hipMalloc / hipFree of a GB, mmap with and without pre-faulting
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
int main(int argc, char **argv) {
    double t0 = now(); CK(hipFree(0)); printf("hip init %.3f s\n", now() - t0);
    int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st); size_t n = st.st_size;
    char *d;
    for (int i = 0; i < 3; i++) {
        t0 = now(); CK(hipMalloc(&d, n + (i << 22))); double ta = now() - t0;
        t0 = now(); CK(hipFree(d)); printf("hipMalloc %.2f GB: %.1f ms, hipFree %.1f ms\n", n / 1e9, ta * 1e3, (now() - t0) * 1e3);
    }
    CK(hipMalloc(&d, n));
    for (int mode = 0; mode < 4; mode++) {
        t0 = now();
        const char *m = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (mode == 1 ? MAP_POPULATE : 0), fd, 0);
        double tm = now() - t0, tp = 0;
        if (mode == 2) { t0 = now(); int r = madvise((void *)m, n, MADV_POPULATE_READ); tp = now() - t0; printf("  madvise(POPULATE_READ) rc %d\n", r); }
        if (mode == 3) {  // populate with 8 threads
            t0 = now(); std::thread th[8];
            for (int t = 0; t < 8; t++) th[t] = std::thread([&, t] { size_t a = n / 8 * t, b = t == 7 ? n : n / 8 * (t + 1); madvise((void *)(m + a), b - a, MADV_POPULATE_READ); });
            for (auto &x : th) x.join(); tp = now() - t0;
        }
        t0 = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); double tc = now() - t0;
        printf("mode %d (0 plain, 1 MAP_POPULATE, 2 madvise populate, 3 the same with 8 threads): mmap %.1f ms, populate %.1f ms, copy %.1f ms = %.1f GB/s\n",
               mode, tm * 1e3, tp * 1e3, tc * 1e3, n / tc / 1e9);
        munmap((void *)m, n);
    }
    return 0;
}
