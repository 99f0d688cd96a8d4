// Fourth probe: which 16-byte pieces a lane loads.  "paired": lane s takes bytes [32s, 32s+32) with two loads
// (each instruction then touches every line of the read); "split": the first instruction covers bytes
// [0, 16*GW) (lane s: piece s), the second the rest (lane s: piece GW + s) -- each instruction touches only
// its half of the read's lines.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

template <int GW, int SPLIT>
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
            const uint32_t oa = SPLIT ? 16u * s : 32u * s, ob = SPLIT ? 16u * (GW + s) : 32u * s + 16u;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (oa < 150u) __builtin_memcpy(&a, rp + oa, 16);
            if (ob < 150u) __builtin_memcpy(&b, rp + ob, 16);
            acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
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
    (void)hipMalloc(&seq, (size_t)n * 150 + 4096); (void)hipMemset(seq, 1, (size_t)n * 150 + 4096);
    (void)hipMalloc(&d_order, n * 4); (void)hipMalloc(&out, 4);
    (void)hipMemcpy(d_order, order.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char *name, auto kernel) {
        float best = 1e9;
        for (int r = 0; r < 8; r++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3(2048), dim3(1024), 0, 0, seq, d_order, n, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        printf("%-40s %.3f ms\n", name, best);
    };
    run("5 lanes, paired 32-byte chunks", k_gather<5, 0>);
    run("5 lanes, split halves", k_gather<5, 1>);
    run("8 lanes, paired", k_gather<8, 0>);
    run("8 lanes, split halves", k_gather<8, 1>);
    return 0;
}
