// pp_devtext.h -- device-side text utilities shared by the SAM tokenizers (pp_tokenize.hip: polish ingest;
// pp_filter_dev.hip: the filter's quick parse): newline index, multi-block exclusive scan, tab search,
// integer parsing, and small host helpers.  Everything lives in an anonymous namespace: each translation
// unit gets its own copies of the kernels.
#pragma once
#include "pp_internal.h"

#include <algorithm>

namespace {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

constexpr u32 NL_BLOCK = 1024 * 64;  // bytes of text per block of the newline kernels (64 per thread)
__device__ __forceinline__ void report(u64 *status, u64 key) { atomicMin(status, key); }

// ---- newline index -------------------------------------------------------------------------------
__device__ __forceinline__ u32 count_nl16(uint4 v) {
    u32 n = 0;
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 x = w[i] ^ 0x0A0A0A0Au;  // zero bytes where the text has '\n'
        n += (u32)__popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u);
    }
    return n;
}

// text is padded with zeros up to a multiple of NL_BLOCK.  blk_cnt[gridDim.x] (zeroed by the caller) is set when a byte
// outside ASCII shows up anywhere: the reference refuses lines that are not valid UTF-8 (BufRead::lines), and which
// ones are is then left to the host parsers -- SAM text is ASCII in practice.
__global__ __launch_bounds__(1024) void k_nl_count(const u8 *__restrict__ text, u32 *__restrict__ blk_cnt) {
    __shared__ u32 s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const uint4 *p = (const uint4 *)(text + (u64)blockIdx.x * NL_BLOCK + (u64)threadIdx.x * 64u);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const u32 high = (q0.x | q0.y | q0.z | q0.w | q1.x | q1.y | q1.z | q1.w | q2.x | q2.y | q2.z | q2.w | q3.x | q3.y | q3.z | q3.w) &
                     0x80808080u;
    if (__ballot(high != 0) && (threadIdx.x & 63u) == 0) atomicOr(&blk_cnt[gridDim.x], 1u);
    const u32 n = count_nl16(q0) + count_nl16(q1) + count_nl16(q2) + count_nl16(q3);
    u32 v = n;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_sum;
}

__global__ __launch_bounds__(1024) void k_nl_write(const u8 *__restrict__ text, const u64 *__restrict__ blk_off,
                                                   u64 *__restrict__ nl_pos) {
    __shared__ u32 s_w[16];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 base = (u64)blockIdx.x * NL_BLOCK + (u64)threadIdx.x * 64u;
    const uint4 *p = (const uint4 *)(text + base);
    const u32 n = count_nl16(p[0]) + count_nl16(p[1]) + count_nl16(p[2]) + count_nl16(p[3]);
    u32 inc = n;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u32 before = inc - n;
    for (u32 i = 0; i < wave; i++) before += s_w[i];
    if (!n) return;
    // the positions come out of the registers: bit 7 of every newline byte, dword by dword (no second look at the text)
    u64 out = blk_off[blockIdx.x] + before;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const uint4 q = p[v];
        const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const u32 x = w[i] ^ 0x0A0A0A0Au;
            u32 m = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
            while (m) {
                const u32 b = ((u32)__ffs((int)m) - 1u) >> 3;  // byte index inside the dword
                m &= m - 1u;
                nl_pos[out++] = base + (u64)(16 * v + 4 * i) + b;
            }
        }
    }
}

// ---- exclusive scan: u32 in -> T out (n + 1 entries) -- block sums, a single-block scan of the sums,
// then every block scans its own 8192 elements on top of its base ------------------------------------
constexpr u32 SCAN_PER_BLOCK = 1024 * 8;

__global__ __launch_bounds__(1024) void k_scan_sums(const u32 *__restrict__ in, u64 n, u32 *__restrict__ sums) {
    __shared__ u32 s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * SCAN_PER_BLOCK + (u64)threadIdx.x * 8u;
    u32 v = 0;
#pragma unroll
    for (u32 i = 0; i < 8; i++)
        if (base + i < n) v += in[base + i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_sum;
}

template <typename T>
__global__ __launch_bounds__(1024) void k_tscan(const u32 *__restrict__ in, u64 n, T *__restrict__ out) {
    __shared__ u64 part[1024];
    const u32 t = threadIdx.x;
    const u64 per = (n + 1023) / 1024;
    const u64 lo = min(n, (u64)t * per), hi = min(n, lo + per);
    u64 s = 0;
    for (u64 i = lo; i < hi; i++) s += in[i];
    part[t] = s;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        const u64 v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 run = part[t] - s;
    for (u64 i = lo; i < hi; i++) {
        out[i] = (T)run;
        run += in[i];
    }
    if (t == 1023) out[n] = (T)part[1023];
}

template <typename T>
__global__ __launch_bounds__(1024) void k_scan_apply(const u32 *__restrict__ in, u64 n, const u64 *__restrict__ sums_off,
                                                     T *__restrict__ out) {
    __shared__ u32 s_w[16];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 base = (u64)blockIdx.x * SCAN_PER_BLOCK + (u64)threadIdx.x * 8u;
    u32 v[8], sum = 0;
#pragma unroll
    for (u32 i = 0; i < 8; i++) {
        v[i] = base + i < n ? in[base + i] : 0u;
        sum += v[i];
    }
    u32 inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    u64 run = sums_off[blockIdx.x] + (inc - sum);
    for (u32 i = 0; i < wave; i++) run += s_w[i];
#pragma unroll
    for (u32 i = 0; i < 8; i++) {
        if (base + i < n) out[base + i] = (T)run;
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = (T)sums_off[gridDim.x];
}

__device__ __forceinline__ bool parse_u(const u8 *s, u32 n, u64 max, u64 &out) {  // str::parse::<uN>()
    u32 i = 0;
    if (n == 0) return false;
    if (s[0] == (u8)'+') { i = 1; if (n == 1) return false; }
    u64 v = 0;
    for (; i < n; i++) {
        if (s[i] < (u8)'0' || s[i] > (u8)'9') return false;
        const u64 d = (u64)(s[i] - (u8)'0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// index of the first '\t' in L[from, n), or n -- eight bytes per step (the text buffer is padded, so a load
// may run past the line; matches beyond n are cut off)
__device__ __forceinline__ u32 find_tab(const u8 *L, u32 from, u32 n) {
    u32 i = from;
    while (i < n) {
        u64 w;
        __builtin_memcpy(&w, L + i, 8);
        const u64 x = w ^ 0x0909090909090909ull;
        const u64 m = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;  // bit 7 of the zero bytes (first one exact)
        if (m) return min(n, i + ((u32)__ffsll((long long)m) - 1u) / 8u);
        i += 8;
    }
    return n;
}

// ---- the lines of one wave, through LDS -------------------------------------------------------------
// One lane per line, but the text comes through LDS: the 64 lines of a wave are one contiguous stretch of the file
// (~35 KB of a 150-bp SAM with QUAL), copied with coalesced 16-byte loads -- every 128-byte line of HBM once -- and
// then walked byte by byte at LDS speed.  One lane per line straight from HBM (round 1) made every load instruction
// touch 64 different cache lines: 245 GB/s.  Lines that do not fit the staging (very long reads) are read from HBM.
// Launch with 64 threads per workgroup; `stage` = TOK_STAGE + 32 bytes of LDS, 16-byte aligned.  Returns false for
// the lanes past the last line; otherwise *L points at the lane's line (LDS or HBM), *n is its length without
// "\n" / "\r\n", *li its number.  The text buffer is padded with >= 64 zero bytes past `size`.
// The stage comes in three sizes, picked per file from its average line length (64 lines + 15 %): the smaller it is, the
// more one-wave workgroups share a CU (10 / 6 / 4) and hide each other's LDS and memory latency.
constexpr u32 TOK_STAGE_S = 16 * 1024 - 64, TOK_STAGE_M = 26 * 1024 - 64, TOK_STAGE_L = 40 * 1024 - 64;
inline u32 tok_stage_for(u64 text_bytes, u64 n_lines) {
    const u64 need = n_lines ? (text_bytes / n_lines + 1) * 64 * 115 / 100 : 0;
    return need <= TOK_STAGE_S ? TOK_STAGE_S : (need <= TOK_STAGE_M ? TOK_STAGE_M : TOK_STAGE_L);
}

template <u32 TOK_STAGE>
__device__ __forceinline__ bool stage_wave_lines(const u8 *__restrict__ text, u64 size, const u64 *__restrict__ nl_pos,
                                                 u64 n_nl, u64 n_lines, u8 *stage, const u8 **L, u32 *n, u64 *li) {
    const u32 lane = threadIdx.x;
    const u64 l0 = (u64)blockIdx.x * 64u, l1 = min(n_lines, l0 + 64u);  // this wave's lines [l0, l1)
    const u64 s0 = l0 ? nl_pos[l0 - 1] + 1 : 0;
    const u64 e1 = (l1 - 1 < n_nl) ? nl_pos[l1 - 1] : size;              // end of its last line
    const u64 a0 = s0 & ~15ull;
    // the copy runs 16 bytes past the last line: find_tab looks at eight bytes at a time
    const u64 want = e1 - a0 + 16;
    const u32 span = (u32)min((u64)TOK_STAGE + 16u, (want + 15ull) & ~15ull);
    // Eight 1 KB rows of the stretch per trip, their loads all out before the first is stored: with one row per trip the
    // compiler waits for every load before its store, and a wave -- alone in its workgroup -- spent twenty-two memory round
    // trips on its ~22 KB (k_tok_parse: 1.3 TB/s of text).  A lane past the end of the stretch copies the stretch's last 16
    // bytes once more, unconditionally: a store under a condition takes its load along into the branch, one wait each.
    const u32 n_rows = (span + 1023u) >> 10;  // (wave-uniform; span is a multiple of 16 and at least 16)
    for (u32 row = 0; row < n_rows; row += 8u) {
        uint4 v[8];
#pragma unroll
        for (u32 j = 0; j < 8u; j++) v[j] = *(const uint4 *)(text + a0 + min(((row + j) << 10) + lane * 16u, span - 16u));
#pragma unroll
        for (u32 j = 0; j < 8u; j++) *(uint4 *)(stage + min(((row + j) << 10) + lane * 16u, span - 16u)) = v[j];
    }
    __syncthreads();
    *li = l0 + lane;
    if (*li >= n_lines) return false;
    const u64 ls = *li ? nl_pos[*li - 1] + 1 : 0, le = *li < n_nl ? nl_pos[*li] : size;
    u32 len = (u32)(le - ls);
    const bool staged = le - a0 + 8 <= (u64)TOK_STAGE + 16u;
    const u8 *p = staged ? (const u8 *)stage + (ls - a0) : text + ls;
    if (len > 0 && p[len - 1] == (u8)'\r') len--;
    *L = p;
    *n = len;
    return true;
}

__device__ __forceinline__ int op_code(u8 c) {
    switch (c) {
    case 'M': return PP_OP_M; case 'I': return PP_OP_I; case 'D': return PP_OP_D; case 'N': return PP_OP_N;
    case 'S': return PP_OP_S; case 'H': return PP_OP_H; case 'P': return PP_OP_P; case '=': return PP_OP_EQ;
    case 'X': return PP_OP_X; default: return -1;
    }
}

// grow a device buffer keeping its first `used` bytes
int dev_grow(pp_ctx *ctx, pp::DevBuf &b, size_t need, size_t used) {
    if (need == 0) need = 16;
    if (b.cap >= need) return PP_OK;
    const size_t want = need + need / 4 + 256;
    void *q = nullptr;
    PP_HIPCHK(ctx, hipMalloc(&q, want));
    if (b.p && used) PP_HIPCHK(ctx, hipMemcpyAsync(q, b.p, used, hipMemcpyDeviceToDevice, ctx->stream));
    if (b.p) {
        PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        PP_HIPCHK(ctx, hipFree(b.p));
    }
    b.p = q;
    b.cap = want;
    return PP_OK;
}

// out[0..n] = exclusive scan of in[0..n); scratch: two small device buffers for the block sums
template <typename T>
int scan_u32(pp_ctx *ctx, pp::DevBuf &b_sums, pp::DevBuf &b_sums_off, const u32 *in, u64 n, T *out) {
    const u64 nb = std::max<u64>(1, (n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    if (int rc = pp::dev_ensure(ctx, b_sums, nb * 4)) return rc;
    if (int rc = pp::dev_ensure(ctx, b_sums_off, (nb + 1) * 8)) return rc;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, in, n, (u32 *)b_sums.p);
    hipLaunchKernelGGL(k_tscan<u64>, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)b_sums.p, nb, (u64 *)b_sums_off.p);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, in, n, (const u64 *)b_sums_off.p, out);
    return PP_OK;
}

template <typename T>
int fetch(pp_ctx *ctx, const void *dev, T *host, size_t n = 1) {
    PP_HIPCHK(ctx, hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

// The newline index of `size` bytes of text on the device (padded with zeros to a multiple of NL_BLOCK): *n_nl positions in
// d_nl, *not_ascii when a byte outside ASCII was seen (there is no index then).  Count, scan, write: two passes over the
// text.  (Round 5 tried ONE pass -- a chained scan over the workgroups, each publishing its count and then its prefix, the
// look-back a wave wide: 0.89 ms per 680 MB file against the 0.33 ms of the two passes.  Every hop of the chain is a
// device-scope load that another XCD's store has to reach through memory.)
int newline_index(pp_ctx *ctx, const u8 *d_text, u64 size, pp::DevBuf &d_blk, pp::DevBuf &d_blkoff, pp::DevBuf &d_nl, u64 *n_nl,
                  u32 *not_ascii) {
    hipStream_t st = ctx->stream;
    const u64 n_blk = (size + NL_BLOCK - 1) / NL_BLOCK;
    *n_nl = 0;
    *not_ascii = 0;
    if (!n_blk) return PP_OK;
    int rc;
    if ((rc = pp::dev_ensure(ctx, d_blk, (n_blk + 1) * 4)) || (rc = pp::dev_ensure(ctx, d_blkoff, (n_blk + 1) * 8))) return rc;
    PP_HIPCHK(ctx, hipMemsetAsync((u32 *)d_blk.p + n_blk, 0, 4, st));
    hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_blk), dim3(1024), 0, st, d_text, (u32 *)d_blk.p);
    hipLaunchKernelGGL(k_tscan<u64>, dim3(1), dim3(1024), 0, st, (const u32 *)d_blk.p, n_blk, (u64 *)d_blkoff.p);
    if ((rc = fetch(ctx, (const u64 *)d_blkoff.p + n_blk, n_nl))) return rc;
    if ((rc = fetch(ctx, (const u32 *)d_blk.p + n_blk, not_ascii))) return rc;
    if (*not_ascii) return PP_OK;
    if ((rc = pp::dev_ensure(ctx, d_nl, std::max<u64>(1, *n_nl) * 8))) return rc;
    hipLaunchKernelGGL(k_nl_write, dim3((unsigned)n_blk), dim3(1024), 0, st, d_text, (const u64 *)d_blkoff.p, (u64 *)d_nl.p);
    return PP_OK;
}

}  // namespace
