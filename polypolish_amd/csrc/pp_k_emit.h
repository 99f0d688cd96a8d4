// pp_k_emit.h -- k_compact / k_finalize: emit codes -> polished bytes.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// emission: code bytes -> polished bytes
// =============================================================================================
__device__ __forceinline__ u32 code_len(u8 c, u32 gp, const MultiEnt *multi, u32 n_multi) {
    if (c == 0) return 0;
    if (c < 0x80u) return 1;
    if (c != 0xFFu) return c & 0x7Fu;
    for (u32 i = 0; i < n_multi; i++)
        if (multi[i].pos == gp) return multi[i].eff;
    return 0;
}

__global__ __launch_bounds__(TILE_THREADS) void k_compact(const u8 *__restrict__ code, u64 G,
                                                          const u64 *__restrict__ win_out,
                                                          const MultiEnt *__restrict__ multi,
                                                          const u32 *__restrict__ counters,
                                                          u8 *__restrict__ out, const u64 *__restrict__ status) {
    __shared__ u32 wsum[TILE_THREADS / 64];
    if (*status != ~0ull) return;
    const u32 w = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const u64 p0 = (u64)w * TILE + 2ull * t;
    const u32 n_multi = counters[1];
    const u8 c0 = (p0 < G) ? code[p0] : 0, c1 = (p0 + 1 < G) ? code[p0 + 1] : 0;
    const u32 l0 = code_len(c0, (u32)p0, multi, n_multi), l1 = code_len(c1, (u32)(p0 + 1), multi, n_multi);
    const u32 s = l0 + l1;
    u32 inc = s;  // inclusive scan within the wave
    for (int o = 1; o < 64; o <<= 1) {
        u32 v = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    u32 base = 0;
    for (u32 i = 0; i < wave; i++) base += wsum[i];
    const u64 off = win_out[w] + base + (inc - s);
    if (c0 && c0 < 0x80u) out[off] = c0;
    if (c1 && c1 < 0x80u) out[off + l0] = c1;
}

// threads [0, n_multi): copy a multi-byte winner into its reserved gap;
// threads [n_multi, n_multi + n_contigs]: output offset of each contig start (and the total)
__global__ __launch_bounds__(64) void k_finalize(const u8 *__restrict__ code, u64 G,
                                                 const u64 *__restrict__ win_out, u32 nwin,
                                                 const MultiEnt *__restrict__ multi,
                                                 const u32 *__restrict__ counters,
                                                 const u8 *__restrict__ seq,
                                                 const u64 *__restrict__ contig_off, u32 n_contigs,
                                                 u8 *__restrict__ out, u64 *__restrict__ ctg_out,
                                                 const u64 *__restrict__ status) {
    if (*status != ~0ull) return;
    const u32 n_multi = counters[1];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_multi + n_contigs + 1u) return;
    u64 gp;
    if (t < n_multi) gp = multi[t].pos; else gp = contig_off[t - n_multi];
    u64 off;
    if (gp >= G) {
        off = win_out[nwin];
    } else {
        const u32 w = (u32)(gp / TILE);
        off = win_out[w];
        for (u64 q = (u64)w * TILE; q < gp; q++) off += code_len(code[q], (u32)q, multi, n_multi);
    }
    if (t < n_multi) {
        const u8 *s = seq + multi[t].off;
        for (u32 b = 0; b < multi[t].len; b++)
            if (s[b] != (u8)'-') out[off++] = s[b];
    } else {
        ctg_out[t - n_multi] = off;
    }
}

}  // namespace pp
