// gsrast_adam.h -- Adam step for the per-Gaussian parameter groups with a PER-ROW learning rate, all groups in one
// launch (third item).  Reference behaviour restated (paths relative to /root/reference/):
//   scene/saro_gaussian.py:306-323  param groups xyz / f_dc / f_rest / opacity / scaling / rotation / temporal_pos,
//                                   torch.optim.Adam(l, lr=0.0, eps=1e-15, fused=True)
//   scene/saro_gaussian.py:345-398  update_learning_rate: param_group['lr'] = lr * self.inv_intergral -- a [P,1] tensor,
//                                   i.e. one learning rate per Gaussian (row)
// torch.optim.Adam (amsgrad=False, maximize=False, weight_decay=0), step t = 1, 2, ...:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Traffic per element: p, g, m, v in, p, m, v out = 28 B; 60 floats per Gaussian = 1.68 kB per Gaussian and step.
#pragma once
#include "gsrast_common.h"

namespace gsrast {

constexpr int ADAM_MAX_GROUPS = 8;
struct AdamGroup {
    float* p; const float* g; float* m; float* v;
    const float* lr_rows;       // [rows] per-row learning rate, or null
    float lr;                   // scalar learning rate (multiplied with lr_rows[i] when that is given)
    unsigned width;             // floats per row
    unsigned long long n;       // rows * width
    unsigned long long first_block, n_blocks;   // this group's slice of the grid
};
struct AdamArgs { AdamGroup grp[ADAM_MAX_GROUPS]; int n_groups; float b1, b2, omb1, omb2, eps, inv_bc1, inv_sqrt_bc2; };   // omb = 1 - beta, rounded once from fp64 (1 - 0.999f is off by 5e-5)

constexpr int ADAM_THREADS = 256, ADAM_PER_THREAD = 4;

__global__ void __launch_bounds__(ADAM_THREADS)
adam_step_kernel(AdamArgs a)
{
    int gi = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_GROUPS; k++) if (k < a.n_groups && blockIdx.x >= a.grp[k].first_block) gi = k;
    const AdamGroup G = a.grp[gi];
    const unsigned long long e0 = ((unsigned long long)(blockIdx.x - G.first_block) * ADAM_THREADS + threadIdx.x) * ADAM_PER_THREAD;
    if (e0 >= G.n) return;
    const bool vec = e0 + ADAM_PER_THREAD <= G.n && ((((uintptr_t)G.p | (uintptr_t)G.g | (uintptr_t)G.m | (uintptr_t)G.v) & 15) == 0);
    float p[4], g[4], m[4], v[4];
    const int cnt = (int)((G.n - e0) < (unsigned long long)ADAM_PER_THREAD ? (G.n - e0) : ADAM_PER_THREAD);
    if (vec) {
        const float4 P4 = *reinterpret_cast<const float4*>(G.p + e0), G4 = *reinterpret_cast<const float4*>(G.g + e0);
        const float4 M4 = *reinterpret_cast<const float4*>(G.m + e0), V4 = *reinterpret_cast<const float4*>(G.v + e0);
        p[0] = P4.x; p[1] = P4.y; p[2] = P4.z; p[3] = P4.w; g[0] = G4.x; g[1] = G4.y; g[2] = G4.z; g[3] = G4.w;
        m[0] = M4.x; m[1] = M4.y; m[2] = M4.z; m[3] = M4.w; v[0] = V4.x; v[1] = V4.y; v[2] = V4.z; v[3] = V4.w;
    } else {
        for (int k = 0; k < cnt; k++) { p[k] = G.p[e0 + k]; g[k] = G.g[e0 + k]; m[k] = G.m[e0 + k]; v[k] = G.v[e0 + k]; }
    }
#pragma unroll
    for (int k = 0; k < ADAM_PER_THREAD; k++) {
        if (k >= cnt) break;
        float lr = G.lr;
        if (G.lr_rows) lr *= G.lr_rows[(e0 + k) / G.width];
        m[k] = a.b1 * m[k] + a.omb1 * g[k];
        v[k] = a.b2 * v[k] + a.omb2 * g[k] * g[k];
        const float denom = sqrtf(v[k]) * a.inv_sqrt_bc2 + a.eps;
        p[k] -= (lr * a.inv_bc1) * (m[k] / denom);
    }
    if (vec) {
        *reinterpret_cast<float4*>(G.p + e0) = make_float4(p[0], p[1], p[2], p[3]);
        *reinterpret_cast<float4*>(G.m + e0) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(G.v + e0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int k = 0; k < cnt; k++) { G.p[e0 + k] = p[k]; G.m[e0 + k] = m[k]; G.v[e0 + k] = v[k]; }
    }
}

} // namespace gsrast
