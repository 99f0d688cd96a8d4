// Fifth probe (round 3): what a window-grouped SEQ layout would buy k_tile.  One block per 2048-position window, as k_tile;
// every window's ~2,900 reads of 150 bases are fetched with the lane-group pattern of k_tile's plain class (5 lanes x 32
// bytes, two 16-byte loads per lane), nothing else is done with them.
//   file order      the reads of a window are scattered through the seq array (what the C ABI delivers: SAM order)
//   grouped         the reads of a window are adjacent, packed back to back (150 B each), windows in order
//   grouped+aligned the same, every window's stretch starting on a 128-byte line
//   grouped 4-bit   two bases per byte (75 -> 80 bytes per read, 16-byte pieces: 5 lanes x 16 bytes, one load per lane)
//   stream          the grouped array read as a plain stream (lane i: 16 bytes at base + 16 i): the floor of any layout
// The gate of VERDICT r2 item 3: go on with a window-grouped batch flavour only if the grouped probes reach <= 0.25 ms.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

constexpr uint32_t N = 6666666, NWIN = 2442, L = 150;

// items[w0[w] .. w0[w+1]) = byte offsets of the window's reads
template <int BYTES_PER_LANE>
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
            const uint32_t o = BYTES_PER_LANE * s;
            if (o < len) {
                uint4 a;
                __builtin_memcpy(&a, rp + o, 16);
                acc ^= a.x ^ a.y ^ a.z ^ a.w;
                if (BYTES_PER_LANE == 32) {
                    __builtin_memcpy(&a, rp + o + 16, 16);
                    acc ^= a.x ^ a.y ^ a.z ^ a.w;
                }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(1024, 8) void k_stream(const uint8_t *seq, uint64_t bytes, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t o = ((uint64_t)blockIdx.x * 1024 + threadIdx.x) * 16; o + 16 <= bytes; o += (uint64_t)gridDim.x * 1024 * 16) {
        const uint4 a = *(const uint4 *)(seq + o);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    std::mt19937_64 rng(1);
    // reads: uniform window; file order = random order of reads
    std::vector<uint32_t> win(N);
    for (auto &x : win) x = (uint32_t)(rng() % NWIN);
    std::vector<uint32_t> w0(NWIN + 1, 0);
    for (uint32_t i = 0; i < N; i++) w0[win[i] + 1]++;
    for (uint32_t w = 0; w < NWIN; w++) w0[w + 1] += w0[w];
    std::vector<uint64_t> it_file(N), it_grp(N), it_aln(N), it_4bit(N);
    std::vector<uint32_t> cur(w0.begin(), w0.end() - 1);
    std::vector<uint64_t> aln_base(NWIN + 1, 0);
    for (uint32_t w = 0; w < NWIN; w++) aln_base[w + 1] = (aln_base[w] + (uint64_t)(w0[w + 1] - w0[w]) * L + 127) / 128 * 128;
    for (uint32_t i = 0; i < N; i++) {
        const uint32_t w = win[i], slot = cur[w]++;
        it_file[slot] = (uint64_t)i * L;                     // read i sits at its file position
        it_grp[slot] = (uint64_t)slot * L;                   // grouped: slot order
        it_aln[slot] = aln_base[w] + (uint64_t)(slot - w0[w]) * L;
        it_4bit[slot] = (uint64_t)slot * 80;
    }
    const size_t bytes = (size_t)N * L + (size_t)NWIN * 128 + 4096;
    uint8_t *seq; uint64_t *d_items; uint32_t *d_w0, *out;
    (void)hipMalloc(&seq, bytes); (void)hipMemset(seq, 1, bytes);
    (void)hipMalloc(&d_items, (size_t)N * 8); (void)hipMalloc(&d_w0, (NWIN + 1) * 4); (void)hipMalloc(&out, 4);
    (void)hipMemcpy(d_w0, w0.data(), (NWIN + 1) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = ((NWIN + 7) / 8) * 8;
    auto run = [&](const char *name, const std::vector<uint64_t> *items, int mode) {
        if (items) (void)hipMemcpy(d_items, items->data(), (size_t)N * 8, hipMemcpyHostToDevice);
        float best = 1e9;
        for (int r = 0; r < 8; r++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<32>, dim3(grid), dim3(1024), 0, 0, seq, d_items, d_w0, L, out);
            else if (mode == 1) hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(1024), 0, 0, seq, d_items, d_w0, 80u, out);
            else hipLaunchKernelGGL(k_stream, dim3(2048), dim3(1024), 0, 0, seq, (uint64_t)N * L, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        printf("%-44s %.3f ms  (%.2f TB/s of the 1.0 GB of read bytes)\n", name, best, (double)N * L / best / 1e9);
    };
    run("file order (scattered reads)", &it_file, 0);
    run("window-grouped, packed", &it_grp, 0);
    run("window-grouped, windows 128-byte aligned", &it_aln, 0);
    run("window-grouped, 4-bit (80 B per read)", &it_4bit, 1);
    run("plain stream over the grouped array", nullptr, 2);
    return 0;
}
