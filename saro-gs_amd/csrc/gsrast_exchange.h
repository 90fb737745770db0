// gsrast_exchange.h -- the kernels of the ALL-GATHER gradient exchange (view_parallel.exchange_gradients(sparse="gather"), round 5).
//
// One view per rank: a rank's gradient rows are exactly zero for every Gaussian its view did not blend (~95 % of a 3 M scene).  Instead of
// reducing arrays the size of the UNION of the views' rows (sparse=True: MAX all-reduce of P flag bytes, all-reduce of 11 floats and
// all-gather of 3 floats per union row, every rank sending zeros for the union rows it did not touch), every rank sends ITS OWN touched rows
// and nothing else, in ONE all-gather:
//     row = 16 words = 64 bytes: { Gaussian index | 11 dense gradient floats (mean 3, opacity 1, scale 3, rotation 4) | dL/dsh factor 3 | 0 }
//     a rank's chunk = header row { count, campos.x, campos.y, campos.z, 0... } + cap rows (cap = the largest count of the step)
// and every rank adds the W chunks into its own arrays IN RANK ORDER (one launch per chunk, no atomics: the indices inside a chunk are
// distinct), so all ranks compute bit-identical means.  xGMI is point-to-point: what a replica must receive in any scheme is the other
// ranks' non-zero rows -- 3 M Gaussians, 8 ranks, 0.12-0.16 M rows per view (their union 0.43 M: the bench's ring of views overlaps):
// 7 x 0.15 M x 64 B = 67 MB per rank and step in one collective; the union form moves 75 MB in three, the dense factor form 483 MB.
#pragma once
#include "gsrast_preprocess.h"

namespace gsrast {

constexpr int GROW_WORDS = 16;          // words per row
__device__ __host__ constexpr int grow_width(int k) { return k == 0 ? 3 : k == 1 ? 1 : k == 2 ? 3 : 4; }     // mean | opacity | scale | rotation
struct GradRowArrays {
    float* dense[4];                    // the dense gradient arrays, [P][grow_width(k)] each: 11 floats per Gaussian
    float* sh;                          // dL/dsh [P][M][3] or null
    float* dc; float* rest;             // or split: [P][1][3] + [P][M-1][3]
    int M;
};

// my touched rows -> rows[1 + pos], pos handed out by one returning atomic per workgroup on the header's count word (rows[0], zeroed by the
// caller); a row past `cap` is dropped (the count still says how many there were).  A workgroup takes GROW_PACK consecutive Gaussians, 16
// per lane (one 16-byte load of flags): 733 atomics on the one word at 3 M -- one per 256 Gaussians, 11.7 k of them, serialise at the
// memory side: the kernel took 0.13 ms.
constexpr int GROW_PER = 16, GROW_PACK = 256 * GROW_PER;
__global__ void __launch_bounds__(256)
grad_rows_pack_kernel(int P, const unsigned char* __restrict__ touched, GradRowArrays a, const float* __restrict__ factor /* [P][3] */,
                      uint32_t* __restrict__ rows, uint32_t cap)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const size_t i0 = (size_t)blockIdx.x * GROW_PACK + (size_t)threadIdx.x * GROW_PER;
    uint32_t bits = 0;                                       // bit k: Gaussian i0 + k is touched
    if (i0 + GROW_PER <= (size_t)P && ((uintptr_t)(touched + i0) & 15) == 0) {
        const uint4 f = *reinterpret_cast<const uint4*>(touched + i0);
        const uint32_t w[4] = { f.x, f.y, f.z, f.w };
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) bits |= ((w[q] >> (8 * e)) & 0xFFu) ? 1u << (4 * q + e) : 0u;
    } else {
        for (int k = 0; k < GROW_PER; k++) if (i0 + k < (size_t)P && touched[i0 + k]) bits |= 1u << k;
    }
    const uint32_t mine = (uint32_t)__builtin_popcount(bits);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= (unsigned)d) incl += o; }
    if (lane == 63u) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine, total = 0;
#pragma unroll
    for (unsigned w = 0; w < 4; w++) { if (w < wave) before += s_w[w]; total += s_w[w]; }
    if (total == 0u) return;                                  // (uniform)
    if (threadIdx.x == 0) s_base = atomicAdd(rows, total);
    __syncthreads();
    uint32_t pos = s_base + before;
    while (bits) {
        const int k = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (pos >= cap) break;
        const size_t i = i0 + (size_t)k;
        uint32_t w[GROW_WORDS];
        w[0] = (uint32_t)i;
        int o = 1;
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int f = 0; f < grow_width(q); f++) w[o++] = __float_as_uint(a.dense[q][i * grow_width(q) + f]);
        w[12] = __float_as_uint(factor[3 * i]); w[13] = __float_as_uint(factor[3 * i + 1]); w[14] = __float_as_uint(factor[3 * i + 2]);
        w[15] = 0u;
        uint4* dst = reinterpret_cast<uint4*>(rows + (size_t)(1u + pos) * GROW_WORDS);
#pragma unroll
        for (int q = 0; q < 4; q++) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        pos++;
    }
}

// zero the rows a set of chunks names: the dense arrays' (what = 1), the SH arrays' (2), both (3).  16 lanes per row: lane 0..11 one
// 16-byte piece of the SH row each, lane 12 the 11 dense floats.
__global__ void __launch_bounds__(256)
grad_rows_clear_kernel(const uint32_t* __restrict__ chunks, int n_chunks, size_t chunk_words, uint32_t cap, GradRowArrays a, int what, uint32_t P)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t part = (uint32_t)(q & 15u);
    const size_t rj = q >> 4;
    const uint32_t r = (uint32_t)(rj / cap), j = (uint32_t)(rj - (size_t)r * cap);
    if (r >= (uint32_t)n_chunks) return;
    const uint32_t* ch = chunks + (size_t)r * chunk_words;
    const uint32_t count = ch[0] < cap ? ch[0] : cap;
    if (j >= count) return;
    const size_t i = ch[(size_t)(1u + j) * GROW_WORDS];
    if (i >= (size_t)P) return;                               // (an index received from a peer: never trusted past the arrays' end)
    const int L = a.M * 3;
    if ((what & 2) && (int)part * 4 < L) {
        if (a.sh) *reinterpret_cast<float4*>(a.sh + i * L + part * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.dc) {
            if (part == 0u) { a.dc[i * 3] = 0.f; a.dc[i * 3 + 1] = 0.f; a.dc[i * 3 + 2] = 0.f; if (L > 3) a.rest[i * (L - 3)] = 0.f; }
            else { float* d = a.rest + i * (L - 3) + (part * 4 - 3); d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f; }
        }
    }
    if ((what & 1) && part == 12u)
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int f = 0; f < grow_width(k); f++) a.dense[k][i * grow_width(k) + f] = 0.f;
}

// one rank's chunk added into the arrays: dense[idx] += scale * row.dense, dL/dsh[idx] += scale * w(dir(mean[idx] - campos)) (x) row.factor.
// One thread per row computes, the workgroup then updates its 256 SH rows with 16-byte accesses, consecutive lanes along a row.
__global__ void __launch_bounds__(PP_THREADS)
grad_rows_add_kernel(const uint32_t* __restrict__ chunk, uint32_t cap, GradRowArrays a, const float* __restrict__ means3D, int D, float scale, uint32_t P)
{
    __shared__ float sh_lds[PP_THREADS * PP_SH_STRIDE];
    __shared__ uint32_t s_idx[PP_THREADS];
    const uint32_t count = chunk[0] < cap ? chunk[0] : cap;
    const uint32_t j0 = blockIdx.x * PP_THREADS;
    if (j0 >= count) return;                                  // (uniform)
    const uint32_t j = j0 + threadIdx.x;
    const int L = a.M * 3;
    float acc[PP_SH_MAX];
#pragma unroll
    for (int k = 0; k < PP_SH_MAX; k++) acc[k] = 0.0f;
    uint32_t i = 0xFFFFFFFFu;
    if (j < count) {
        const uint4* src = reinterpret_cast<const uint4*>(chunk + (size_t)(1u + j) * GROW_WORDS);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        const uint32_t w[GROW_WORDS] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
        i = w[0] < P ? w[0] : 0xFFFFFFFFu;                    // (an index received from a peer: a row past the arrays' end is skipped)
        int o = 1;
        if (i != 0xFFFFFFFFu)
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int f = 0; f < grow_width(k); f++) { float* d = a.dense[k] + (size_t)i * grow_width(k) + f; *d = *d + scale * __uint_as_float(w[o]); o++; }
        if (i != 0xFFFFFFFFu && (a.sh || a.dc)) {
            const float pos[3] = { means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2] };
            sh_factor_term(acc, pos, D, __uint_as_float(chunk[1]), __uint_as_float(chunk[2]), __uint_as_float(chunk[3]),
                           __uint_as_float(w[12]), __uint_as_float(w[13]), __uint_as_float(w[14]));
        }
    }
    if (!a.sh && !a.dc) return;
    s_idx[threadIdx.x] = i;
    float* my_lds = sh_lds + threadIdx.x * PP_SH_STRIDE;
#pragma unroll
    for (int k = 0; k < PP_SH_MAX; k++) if (k < L) my_lds[k] = acc[k] * scale;
    __syncthreads();
    const int q4 = L >> 2, n_here = (int)min((uint32_t)PP_THREADS, count - j0);
    for (int q = threadIdx.x; q < n_here * q4; q += PP_THREADS) {
        const int r = q / q4, part = q - r * q4;
        const float* src = sh_lds + r * PP_SH_STRIDE + part * 4;
        if (s_idx[r] == 0xFFFFFFFFu) continue;
        const size_t g = (size_t)s_idx[r];
        if (a.sh) {
            float4* d = reinterpret_cast<float4*>(a.sh + g * L + part * 4);
            float4 v = *d;
            v.x += src[0]; v.y += src[1]; v.z += src[2]; v.w += src[3];
            *d = v;
        }
        if (a.dc) {
            if (part == 0) { a.dc[g * 3] += src[0]; a.dc[g * 3 + 1] += src[1]; a.dc[g * 3 + 2] += src[2]; a.rest[g * (L - 3)] += src[3]; }
            else { float* d = a.rest + g * (L - 3) + (part * 4 - 3); d[0] += src[0]; d[1] += src[1]; d[2] += src[2]; d[3] += src[3]; }
        }
    }
}

}  // namespace gsrast
