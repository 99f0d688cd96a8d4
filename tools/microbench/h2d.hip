// h2d.hip -- how fast can 1.2 GB of page-cache-backed file text reach the device?  (experiment behind the driver's upload path)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d tools/microbench/h2d.hip -lpthread && /tmp/h2d <file>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    size_t n = st.st_size;
    const char *m = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    double t0 = now(); CK(hipFree(0)); printf("hip init %.3f s\n", now() - t0);
    char *d; CK(hipMalloc(&d, n));
    hipStream_t s; CK(hipStreamCreate(&s));
    // (a) pageable copy straight from the mapping
    t0 = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); double ta = now() - t0;
    printf("pageable hipMemcpy from mmap: %.3f s = %.1f GB/s\n", ta, n / ta / 1e9);
    t0 = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); ta = now() - t0;
    printf("  again: %.3f s = %.1f GB/s\n", ta, n / ta / 1e9);
    // (b) pinned allocation cost + copy speed
    for (size_t mb : {32, 128}) {
        t0 = now(); void *p; CK(hipHostMalloc(&p, mb << 20)); double tp = now() - t0;
        t0 = now(); memcpy(p, m, mb << 20); double tm = now() - t0;
        t0 = now(); CK(hipMemcpy(d, p, mb << 20, hipMemcpyHostToDevice)); double tc = now() - t0;
        printf("hipHostMalloc %zu MB: %.1f ms; memcpy into it %.1f ms (%.1f GB/s); H2D %.1f ms (%.1f GB/s)\n", mb, tp * 1e3, tm * 1e3,
               (mb << 20) / tm / 1e9, tc * 1e3, (mb << 20) / tc / 1e9);
        CK(hipHostFree(p));
    }
    // (c) register the mapping itself
    t0 = now(); hipError_t e = hipHostRegister((void *)m, n, hipHostRegisterReadOnly); double tr = now() - t0;
    printf("hipHostRegister(read-only) of the mapping: %s, %.1f ms\n", hipGetErrorString(e), tr * 1e3);
    if (e == hipSuccess) {
        t0 = now(); CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice)); ta = now() - t0;
        printf("  copy from the registered mapping: %.3f s = %.1f GB/s\n", ta, n / ta / 1e9);
        hipHostUnregister((void *)m);
    } else { (void)hipGetLastError(); }
    // (d) ring of pinned buffers filled by T threads, async copies
    for (int T : {4, 8, 16}) {
        const size_t CH = 16u << 20; const int NB = 2 * T;
        std::vector<void *> pin(NB); std::vector<hipEvent_t> ev(NB);
        t0 = now();
        for (int i = 0; i < NB; i++) { CK(hipHostMalloc(&pin[i], CH)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        double talloc = now() - t0;
        t0 = now();
        size_t nch = (n + CH - 1) / CH;
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            (void)hipSetDevice(0);
            hipStream_t ms; (void)hipStreamCreateWithFlags(&ms, hipStreamNonBlocking);
            int k = 0;
            for (size_t c = t; c < nch; c += T, k ^= 1) {
                const int b = 2 * t + k;
                (void)hipEventSynchronize(ev[b]);
                const size_t off = c * CH, len = std::min(CH, n - off);
                memcpy(pin[b], m + off, len);
                (void)hipMemcpyAsync(d + off, pin[b], len, hipMemcpyHostToDevice, ms);
                (void)hipEventRecord(ev[b], ms);
            }
            (void)hipStreamSynchronize(ms); (void)hipStreamDestroy(ms);
        });
        for (auto &x : th) x.join();
        double tt = now() - t0;
        printf("ring, %d threads x 2 x 16 MB pinned: alloc %.1f ms, copy %.3f s = %.1f GB/s\n", T, talloc * 1e3, tt, n / tt / 1e9);
        for (int i = 0; i < NB; i++) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
    }
    return 0;
}
