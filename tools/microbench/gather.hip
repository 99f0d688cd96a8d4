// Ceiling probe for k_tile's access pattern: N reads of 150 bytes at stride 150 in one array, fetched in
// a RANDOM order (the order k_tile sees them: grouped by window, i.e. random with respect to memory),
// 8 lanes x 32 bytes per read, nothing else.  Prints achieved GB/s in algorithmic bytes (150 B per read).
//   hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

template <int GW, bool ALIGNED>
__global__ __launch_bounds__(1024) void k_gather(const uint8_t *seq, const uint32_t *order, uint32_t n, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    constexpr uint32_t IPP = 64 / GW;
    const uint32_t g = GW == 8 ? lane >> 3 : (GW == 5 ? (lane * 52u) >> 8 : (lane * 43u) >> 8), s = lane - GW * g;
    uint32_t acc = 0;
    for (uint32_t first = wave_global * IPP; first < n; first += n_waves * IPP) {
        const uint32_t j = first + g;
        if (g < IPP && j < n) {
            const uint8_t *rp = seq + (uint64_t)order[j] * 150u;
            const uint32_t mis = ALIGNED ? (uint32_t)((uintptr_t)rp & 31u) : 0u;
            if (32u * s < mis + 150u) {
                const uint8_t *p = rp - mis + 32u * s;
                uint4 a, b;
                __builtin_memcpy(&a, p, 16);
                __builtin_memcpy(&b, p + 16, 16);
                acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint32_t n = 6666666;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::mt19937 rng(1);
    std::shuffle(order.begin(), order.end(), rng);
    uint8_t *seq; uint32_t *d_order, *out;
    hipMalloc(&seq, (size_t)n * 150 + 4096); hipMemset(seq, 1, (size_t)n * 150 + 4096);
    hipMalloc(&d_order, n * 4); hipMalloc(&out, 4);
    hipMemcpy(d_order, order.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kernel, bool sequential) {
        if (sequential) { std::vector<uint32_t> o(n); for (uint32_t i = 0; i < n; i++) o[i] = i; hipMemcpy(d_order, o.data(), n * 4, hipMemcpyHostToDevice); }
        else hipMemcpy(d_order, order.data(), n * 4, hipMemcpyHostToDevice);
        for (int blocks : {512, 2048}) {
            float best = 1e9;
            for (int r = 0; r < 6; r++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), 0, 0, seq, d_order, n, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("%-34s blocks %4d  %.3f ms  %.0f GB/s (150 B/read)\n", name, blocks, best, n * 150.0 / best / 1e6);
        }
    };
    run("random, 8 lanes aligned", k_gather<8, true>, false);
    run("random, 6 lanes aligned", k_gather<6, true>, false);
    run("random, 5 lanes unaligned", k_gather<5, false>, false);
    run("random, 8 lanes unaligned", k_gather<8, false>, false);
    run("sequential, 8 lanes aligned", k_gather<8, true>, true);
    run("sequential, 5 lanes unaligned", k_gather<5, false>, true);
    return 0;
}
