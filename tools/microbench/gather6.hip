// Sixth probe (round 3): does anything make the FILE-ORDER gather of k_tile's plain class move fewer bytes?
// k_tile moves 2.14 GB for 1.14 GB of algorithmic bytes at 5.3 TB/s -- 85 % of the chip's practical copy rate: the kernel is
// HBM-bound on the bytes the memory system actually fetches, 128-byte lines for 150-byte reads at arbitrary offsets (2.16
// lines each).  Tried here, same pattern as gather5 (one block per window, 5 lanes x 2 x 16 bytes per read):
//   cache-policy bits on the loads (nt / sc0 / sc1 / all)  -- would a streaming or L1-bypassing load fetch sectors, not lines?
//   the read's pitch (150 packed | 160 = 32-byte aligned | 192 = 64-byte aligned | 256 = line aligned)
//   4-bit reads in file order (75 bytes at pitch 80 | pitch 128 = one line per read)
// Run plain for the times; under `rocprofv3 --pmc FETCH_SIZE` for the bytes (one kernel name per variant).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

constexpr uint32_t N = 6666666, NWIN = 2442;

template <int MODE>
__device__ __forceinline__ void load32(const uint8_t *p, uint4 &a, uint4 &b) {
    if (MODE == 0) { __builtin_memcpy(&a, p, 16); __builtin_memcpy(&b, p + 16, 16); }
    else if (MODE == 1) asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %2, off offset:16 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else if (MODE == 4) asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1 nt\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}

// MODE: cache-policy bits; BPL: bytes per lane (32: two loads, 16: one); items = byte offset of every read, by window
template <int MODE, int BPL>
__global__ __launch_bounds__(1024, 8) void k_gather(const uint8_t *seq, const uint64_t *items, const uint32_t *w0, uint32_t len, uint32_t *out) {
    const uint32_t b = blockIdx.x, per = gridDim.x >> 3, w = (b & 7u) * per + (b >> 3);
    if (w >= NWIN) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t g = (lane * 52u) >> 8, s = lane - 5u * g;  // 12 groups of 5 lanes
    const uint32_t e0 = w0[w], e1 = w0[w + 1];
    uint32_t acc = 0;
    for (uint32_t first = e0 + wave * 12u; first < e1; first += 16u * 12u) {
        const uint32_t j = first + g;
        if (g < 12u && j < e1) {
            const uint8_t *rp = seq + items[j];
            const uint32_t o = BPL * s;
            if (o < len) {
                uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
                if (BPL == 32) load32<MODE>(rp + o, a, c);
                else __builtin_memcpy(&a, rp + o, 16);
                acc ^= a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// one LANE per read: NCH 16-byte loads from the lane's own read (what k_tile's wide4_pass does with the 4-bit mirror)
template <int NCH>
__global__ __launch_bounds__(1024, 8) void k_gather_wide(const uint8_t *seq, const uint64_t *items, const uint32_t *w0, uint32_t *out) {
    const uint32_t b = blockIdx.x, per = gridDim.x >> 3, w = (b & 7u) * per + (b >> 3);
    if (w >= NWIN) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e0 = w0[w], e1 = w0[w + 1];
    const uint32_t per_wave = (e1 - e0 + 15u) / 16u, lo = min(e1, e0 + wave * per_wave), hi = min(e1, lo + per_wave);
    uint32_t acc = 0;
    for (uint32_t first = lo; first < hi; first += 64u) {
        const uint32_t j = first + lane;
        if (j < hi) {
            const uint8_t *rp = seq + items[j];
            uint4 a[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) __builtin_memcpy(&a[c], rp + 16 * c, 16);
#pragma unroll
            for (int c = 0; c < NCH; c++) acc ^= a[c].x ^ a[c].y ^ a[c].z ^ a[c].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    std::mt19937_64 rng(1);
    std::vector<uint32_t> win(N);
    for (auto &x : win) x = (uint32_t)(rng() % NWIN);
    std::vector<uint32_t> w0(NWIN + 1, 0);
    for (uint32_t i = 0; i < N; i++) w0[win[i] + 1]++;
    for (uint32_t w = 0; w < NWIN; w++) w0[w + 1] += w0[w];
    std::vector<uint32_t> slot_of(N), cur(w0.begin(), w0.end() - 1);
    for (uint32_t i = 0; i < N; i++) slot_of[i] = cur[win[i]]++;
    const size_t bytes = (size_t)N * 256 + 4096;
    uint8_t *seq; uint64_t *d_items; uint32_t *d_w0, *out;
    (void)hipMalloc(&seq, bytes); (void)hipMemset(seq, 1, bytes);
    (void)hipMalloc(&d_items, (size_t)N * 8); (void)hipMalloc(&d_w0, (NWIN + 1) * 4); (void)hipMalloc(&out, 4);
    (void)hipMemcpy(d_w0, w0.data(), (NWIN + 1) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = ((NWIN + 7) / 8) * 8;
    std::vector<uint64_t> items(N);
    auto place = [&](uint32_t pitch) {  // read i at i * pitch (file order), listed by window
        for (uint32_t i = 0; i < N; i++) items[slot_of[i]] = (uint64_t)i * pitch;
        (void)hipMemcpy(d_items, items.data(), (size_t)N * 8, hipMemcpyHostToDevice);
    };
    auto time_it = [&](const char *name, auto launch, double read_bytes) {
        float best = 1e9;
        for (int r = 0; r < 6; r++) {
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        printf("%-58s %.3f ms  (%.2f TB/s of read bytes)\n", name, best, read_bytes / best / 1e9);
    };
#define G(MODE, BPL, LEN) [&] { hipLaunchKernelGGL((k_gather<MODE, BPL>), dim3(grid), dim3(1024), 0, 0, seq, d_items, d_w0, (uint32_t)(LEN), out); }
    const double rb = (double)N * 150;
    place(150);
    time_it("150 B reads, pitch 150, plain loads          [mode 0]", G(0, 32, 150), rb);
    time_it("150 B reads, pitch 150, nt                   [mode 1]", G(1, 32, 150), rb);
    time_it("150 B reads, pitch 150, sc0                  [mode 2]", G(2, 32, 150), rb);
    time_it("150 B reads, pitch 150, sc1                  [mode 3]", G(3, 32, 150), rb);
    time_it("150 B reads, pitch 150, sc0 sc1              [mode 4]", G(4, 32, 150), rb);
    time_it("150 B reads, pitch 150, sc0 sc1 nt           [mode 5]", G(5, 32, 150), rb);
    place(160);
    time_it("150 B reads, pitch 160 (32-byte aligned)", G(0, 32, 150), rb);
    place(192);
    time_it("150 B reads, pitch 192 (64-byte aligned)", G(0, 32, 150), rb);
    place(256);
    time_it("150 B reads, pitch 256 (line aligned)", G(0, 32, 150), rb);
    place(80);
    time_it("4-bit reads (75 B), pitch 80", G(0, 16, 80), rb);
    place(128);
    time_it("4-bit reads (75 B), pitch 128 (one line per read)", G(0, 16, 80), rb);
#define GW_(NCH) [&] { hipLaunchKernelGGL((k_gather_wide<NCH>), dim3(grid), dim3(1024), 0, 0, seq, d_items, d_w0, out); }
    place(75);
    time_it("4-bit reads, pitch 75, lane groups (5 x 16 B per read)", G(0, 16, 75), rb);
    time_it("4-bit reads, pitch 75, ONE LANE per read (5 loads)", GW_(5), rb);
    place(150);
    time_it("150 B reads, pitch 150, ONE LANE per read (10 loads)", GW_(10), rb);
    {   // window-grouped: read i of the sorted order at slot * 75
        for (uint32_t i = 0; i < N; i++) items[slot_of[i]] = (uint64_t)slot_of[i] * 75;
        (void)hipMemcpy(d_items, items.data(), (size_t)N * 8, hipMemcpyHostToDevice);
        time_it("4-bit reads, window-grouped, lane groups", G(0, 16, 75), rb);
        time_it("4-bit reads, window-grouped, ONE LANE per read", GW_(5), rb);
    }
    return 0;
}
