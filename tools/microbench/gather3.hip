// Third probe: start from the window-organised gather (0.277 ms) and add k_tile's ingredients one by one.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

template <int TAIL, int LDSKB, int ITEMS16>
__global__ __launch_bounds__(1024, 8) void k_win(const uint8_t *seq, const uint32_t *order, const uint4 *items,
                                                 const uint32_t *win_off, uint32_t *out, uint32_t nwin) {
    __shared__ uint32_t lds[LDSKB * 256 + 1];
    uint32_t per = gridDim.x >> 3;
    uint32_t w = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (w >= nwin) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (LDSKB) { for (uint32_t i = threadIdx.x; i < LDSKB * 256; i += 1024) lds[i] = 0; __syncthreads(); }
    const uint32_t e0 = win_off[w], e1 = win_off[w + 1];
    const uint32_t per_wave = ((e1 - e0 + 15u) / 16u + 7u) / 8u * 8u;
    const uint32_t lo = min(e1, e0 + wave * per_wave), hi = min(e1, lo + per_wave);
    const uint32_t g = lane >> 3, s = lane & 7u;
    uint32_t acc = 0;
    if (!ITEMS16) {
        for (uint32_t first = lo; first < hi; first += 8) {
            const uint32_t j = first + g;
            if (j < hi) {
                const uint8_t *rp = seq + (uint64_t)order[j] * 150u;
                const uint32_t mis = (uint32_t)((uintptr_t)rp & 31u);
                if (TAIL) { uint32_t t; __builtin_memcpy(&t, rp + 146, 4); acc ^= t; }
                if (32u * s < mis + 150u) {
                    const uint8_t *p = rp - mis + 32u * s;
                    uint4 a, b;
                    __builtin_memcpy(&a, p, 16);
                    __builtin_memcpy(&b, p + 16, 16);
                    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
                }
            }
        }
    } else {
        // k_tile's way: a coalesced batch of 64 16-byte items per wave, fields by ds_bpermute
        for (uint32_t eb = lo; eb < hi; eb += 64) {
            const uint32_t nb = min(64u, hi - eb);
            const uint4 my = items[eb + min(lane, nb - 1u)];
            for (uint32_t first = 0; first < nb; first += 8) {
                const uint32_t j = first + g;
                const int src = (int)(min(j, nb - 1u) << 2);
                const uint32_t ex = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)my.x), ey = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)my.y);
                if (j < nb) {
                    const uint8_t *rp = seq + ((uint64_t)ex | ((uint64_t)(ey & 0xFFu) << 32));
                    const uint32_t mis = (uint32_t)((uintptr_t)rp & 31u);
                    if (TAIL) { uint32_t t; __builtin_memcpy(&t, rp + 146, 4); acc ^= t; }
                    if (32u * s < mis + 150u) {
                        const uint8_t *p = rp - mis + 32u * s;
                        uint4 a, b;
                        __builtin_memcpy(&a, p, 16);
                        __builtin_memcpy(&b, p + 16, 16);
                        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
                    }
                }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc + (LDSKB ? lds[lane] : 0);
}

int main() {
    const uint32_t n = 6666666, nwin = 2442;
    std::mt19937 rng(1);
    std::vector<uint32_t> win_of(n);
    for (uint32_t i = 0; i < n; i++) win_of[i] = rng() % nwin;
    std::vector<uint32_t> win_off(nwin + 1, 0);
    for (uint32_t i = 0; i < n; i++) win_off[win_of[i] + 1]++;
    for (uint32_t w = 0; w < nwin; w++) win_off[w + 1] += win_off[w];
    std::vector<uint32_t> sorted(n), cur(win_off.begin(), win_off.end() - 1);
    for (uint32_t i = 0; i < n; i++) sorted[cur[win_of[i]]++] = i;
    std::vector<uint4> items(n);
    for (uint32_t i = 0; i < n; i++) { uint64_t so = (uint64_t)sorted[i] * 150u; items[i] = make_uint4((uint32_t)so, (uint32_t)(so >> 32) | (150u << 24), 5, i); }
    uint8_t *seq; uint32_t *d_order, *d_off, *out; uint4 *d_items;
    hipMalloc(&seq, (size_t)n * 150 + 4096); hipMemset(seq, 1, (size_t)n * 150 + 4096);
    hipMalloc(&d_order, n * 4); hipMalloc(&d_off, (nwin + 1) * 4); hipMalloc(&out, 4); hipMalloc(&d_items, (size_t)n * 16);
    hipMemcpy(d_off, win_off.data(), (nwin + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_order, sorted.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_items, items.data(), (size_t)n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kernel) {
        float best = 1e9;
        for (int r = 0; r < 8; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3((nwin + 7) / 8 * 8), dim3(1024), 0, 0, seq, d_order, d_items, d_off, out, nwin);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        printf("%-52s %.3f ms\n", name, best);
    };
    run("base (order[] index, 8 per pass)", k_win<0, 0, 0>);
    run("+ tail dword load", k_win<1, 0, 0>);
    run("+ 75 KB LDS zeroed per block", k_win<0, 75, 0>);
    run("+ 16-byte items + bpermute", k_win<0, 0, 1>);
    run("+ items + tail + LDS", k_win<1, 75, 1>);
    return 0;
}
