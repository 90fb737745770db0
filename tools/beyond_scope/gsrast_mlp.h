// gsrast_mlp.h -- weight / bias gradient of a Linear layer over ~1e6 rows on the fp32 matrix cores.
// Beyond SURVEY.md 8f (all of its rows are built): this is the measured bottleneck of the DYNAMIC-stage iteration once they are
// (DESIGN.md 8).  The reference's deformation heads (/root/reference/scene/saro_gaussian.py:104-110: four 3-layer nn.Linear
// stacks, hidden width 128, evaluated for every Gaussian, :779-812) need, per layer and step,
//     dW[N1][N2] = sum_r G[r][n1] * X[r][n2]     (G = gradient at the layer's output, X = its input; r over all P Gaussians)
//     db[N1]     = sum_r G[r][n1]
// -- a GEMM with K = P ~ 1e6 and a 128x128 (or smaller) result, which the BLAS library runs as 16 workgroups (10.5 + 4 ms of
// the 24 ms the three heads take in torch at P = 1e6).  Here the rows are split over ~1000 workgroups (split-K); a wave owns
// one 32-row block of dW and up to four 32-column blocks, streams two rows per v_mfma_f32_32x32x2_f32 straight from global
// memory (the operand layout -- lane l holds element [l & 31] of row [l >> 5] -- IS a coalesced 128-byte read of each row, so no
// LDS staging), and adds its partial block with float atomics at the end.  fp32 in, fp32 accumulate: an fmaf chain, like
// the reference's fp32 Linear layers.
#pragma once
#include "gsrast_common.h"

namespace gsrast {

typedef float mlp_f32x16 __attribute__((ext_vector_type(16)));
constexpr int MLP_MAX_N = 128;          // layer widths up to 128 (the reference: 41 / 32 -> 128 -> 128 | 64 -> 3 | 7 | 48 | 1)

// NJ = 32-column blocks of dW per wave (1, 2 or 4); STEPS = k-steps (row pairs) whose operands are in flight together: the
// loop is one memory latency per STEPS MFMA groups, so the skinny layers (NJ = 1: 64 cycles of MFMA per row pair) fly more.
template <int NJ, int STEPS>
__global__ void __launch_bounds__(256)
mlp_wgrad_kernel(const float* __restrict__ G, const float* __restrict__ X, int M, int N1, int N2, int rows_per_chunk,
                 float* __restrict__ dW, float* __restrict__ db)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb1 = (N1 + 31) >> 5, nb2 = (N2 + 31) >> 5;
    const int nb1p = nb1 <= 1 ? 1 : (nb1 == 2 ? 2 : 4);           // waves per workgroup spent on row blocks of dW
    const int bi = wave % nb1p, jg = wave / nb1p, jstep = 4 / nb1p;
    if (bi >= nb1 || jg >= nb2) return;
    const int col = lane & 31, half = lane >> 5;
    const int n1 = bi * 32 + col;
    const bool a_ok = n1 < N1;
    int bj[NJ]; bool b_ok[NJ]; int nj = 0;
#pragma unroll
    for (int t = 0; t < NJ; t++) { bj[t] = jg + t * jstep; b_ok[t] = false; if (bj[t] < nb2) { nj = t + 1; b_ok[t] = bj[t] * 32 + col < N2; } }
    mlp_f32x16 acc[NJ];
#pragma unroll
    for (int t = 0; t < NJ; t++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[t][v] = 0.0f;
    float bsum = 0.0f;
    const long long r_begin = (long long)blockIdx.x * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < M ? r_begin + rows_per_chunk : M;
    // double-buffered: the operands of the next STEPS row pairs are requested before the current ones enter the matrix core.
    // The loads are unconditional (indices clamped, values masked on use): a branch around a load would make the compiler wait
    // for ALL outstanding loads before the first MFMA, which undoes the double buffering.
    const int n1c = a_ok ? n1 : N1 - 1;
    int n2c[NJ];
#pragma unroll
    for (int t = 0; t < NJ; t++) n2c[t] = b_ok[t] ? bj[t] * 32 + col : N2 - 1;
    float a[2][STEPS], b[2][STEPS][NJ];
    auto fetch = [&](long long r0, float (&fa)[STEPS], float (&fb)[STEPS][NJ]) {
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            long long r = r0 + 2 * s + half;
            r = r < r_end ? r : r_end - 1;
            fa[s] = G[r * N1 + n1c];
#pragma unroll
            for (int t = 0; t < NJ; t++) fb[s][t] = X[r * N2 + n2c[t]];
        }
    };
    auto consume = [&](long long r0, const float (&fa)[STEPS], const float (&fb)[STEPS][NJ]) {
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const bool rok = r0 + 2 * s + half < r_end;
            const float av = (rok && a_ok) ? fa[s] : 0.0f;
            bsum += av;
#pragma unroll
            for (int t = 0; t < NJ; t++) if (t < nj) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_ok[t] ? fb[s][t] : 0.0f, acc[t], 0, 0, 0);
        }
    };
    fetch(r_begin, a[0], b[0]);
    for (long long r0 = r_begin; r0 < r_end; r0 += 4 * STEPS) {
        fetch(r0 + 2 * STEPS, a[1], b[1]);
        consume(r0, a[0], b[0]);
        fetch(r0 + 4 * STEPS, a[0], b[0]);
        consume(r0 + 2 * STEPS, a[1], b[1]);
    }
    // C/D layout of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < NJ; t++) {
        if (t >= nj) break;
        const int n2 = bj[t] * 32 + col;
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int row = bi * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (row < N1 && n2 < N2) atomicAdd(dW + (size_t)row * N2 + n2, acc[t][v]);
        }
    }
    if (db && jg == 0) {
        const float tot = bsum + __shfl_xor(bsum, 32);
        if (half == 0 && a_ok) atomicAdd(db + n1, tot);
    }
}

} // namespace gsrast
