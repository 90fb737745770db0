/*
 * gsrast_oracle.c -- CPU restatement of the SaRO-GS differentiable Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under saro-gs_amd/ may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY PINNING STATUS: "parity unpinned" against a RUNNING reference -- pinned piecewise against everything reachable.
 *   The reference implementation of this path exists only as CUDA (.cu) sources that include
 *   <cuda.h>, "cuda_runtime.h", <cooperative_groups.h> and <cub/cub.cuh>; none of those headers
 *   exist in this image and the reference ships no tests or golden vectors for the path
 *   (SURVEY.md section 4).  It is therefore unbuildable here (no oracle/_ref) and this file is an independent
 *   restatement of the published algorithm, following the reference file:line cited at every
 *   function.  What pins it (DESIGN.md section 5 has the table):
 *     - against the reference's OWN Python, imported by tests/golden/make_golden.py: SH colour (utils/sh_utils.py eval_sh), the
 *       camera-matrix conventions and the point projection (utils/graphics_utils.py, scene/cameras.py), cov3D -- packing,
 *       quaternion convention, R S S^T R^T, scale_modifier (utils/general_utils.py build_scaling_rotation / strip_symmetric as
 *       composed in scene/saro_gaussian.py:33-37)                                      -> tests/test_oracle_golden.py;
 *     - against an independent derivation (tests/math_renderer.py: torch float64 + autograd, written from the mathematics,
 *       no code or derivation shared with this file): cov2D / conic / radius / tile rectangle, the compositing recurrence,
 *       median depth, and ALL hand-derived backward formulas (this file's gradients equal autograd's to 1e-7 relative)
 *                                                                                      -> tests/test_oracle_independent.py;
 *     - finite differences of the forward reproduce the backward                      -> tests/test_oracle_grad.py.
 *   Not pinnable here: that the CUDA binary evaluates these formulas in exactly this association (FMA contraction).
 *
 * RST = /root/reference/submodules/gaussian_rasterization_ch3/cuda_rasterizer
 *
 * The file is compiled twice (oracle/Makefile):
 *   -DORC_REAL=float  -DORC_PFX=orc32_   arithmetic exactly as the reference's fp32 kernels, in a
 *                                         fixed evaluation order (no FMA contraction; FMAs only
 *                                         where written as fmaf) so that the HIP kernels can be
 *                                         compared with it bit for bit.
 *   -DORC_REAL=double -DORC_PFX=orc64_   "truth" build: every continuous quantity in double while
 *                                         every DISCRETE decision (culling, radii, clamp flags,
 *                                         alpha / transmittance thresholds) is replayed from the
 *                                         fp32 build, so both builds traverse the same control
 *                                         flow and differ by rounding only.
 *
 * Matrix storage: viewmatrix / projmatrix are the reference's transposed 4x4 (flat index
 * 4*c + r holds math M[r][c]); 3x3 helper matrices below use m[c][r] (column c, row r) like the
 * glm library the reference kernels are written against.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_REAL
#define ORC_REAL float
#endif
#ifndef ORC_PFX
#define ORC_PFX orc32_
#endif
#define ORC_CAT2(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT2(a, b)
#define FN(name) ORC_CAT(ORC_PFX, name)

typedef ORC_REAL real;

#define TILE_X 16 /* RST/config.h:16 */
#define TILE_Y 16 /* RST/config.h:17 */

/* ------------------------------------------------------------------------------------------ */
/* exp().  RST/forward.cu:349 and RST/backward.cu:498 call exp(float).  Two selectable
 * implementations: mode 0 = a fixed-sequence fp32 exp (Cody-Waite reduction + degree-5
 * polynomial, every step an exactly rounded IEEE operation), reproducible bit for bit on the
 * GPU; mode 1 = the C library's expf.  Accuracy of mode 0 is checked against libm in
 * tests/test_oracle_units.py. */
static int g_exp_mode = 0;
void FN(set_exp_mode)(int mode) { g_exp_mode = mode; }

static inline float det_expf(float x)
{
    if (x < -80.0f) return 0.0f;
    if (x > 80.0f) x = 80.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    return ldexpf(y, (int)n);
}
static inline float ctl_expf(float x) { return g_exp_mode == 0 ? det_expf(x) : expf(x); }
float FN(expf)(float x) { return ctl_expf(x); }

static inline real real_exp(real x, float xf)
{
    if (sizeof(real) == sizeof(float)) return (real)ctl_expf(xf);
    return (real)exp((double)x);
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { real m[3][3]; } mat3; /* m[c][r] */

static inline mat3 mat3_mul(const mat3* a, const mat3* b)
{ /* glm operator*(mat3,mat3): res[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] + a[2][r]*b[c][2] */
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            o.m[c][r] = a->m[0][r] * b->m[c][0] + a->m[1][r] * b->m[c][1] + a->m[2][r] * b->m[c][2];
    return o;
}
static inline mat3 mat3_t(const mat3* a)
{
    mat3 o;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) o.m[c][r] = a->m[r][c];
    return o;
}

/* RST/auxiliary.h:58-77 */
static inline void xform4x3(const real p[3], const float* M, real o[3])
{
    for (int r = 0; r < 3; r++)
        o[r] = (real)M[r] * p[0] + (real)M[4 + r] * p[1] + (real)M[8 + r] * p[2] + (real)M[12 + r];
}
static inline void xform4x4(const real p[3], const float* M, real o[4])
{
    for (int r = 0; r < 4; r++)
        o[r] = (real)M[r] * p[0] + (real)M[4 + r] * p[1] + (real)M[8 + r] * p[2] + (real)M[12 + r];
}

/* RST/auxiliary.h:41-44: evaluated in double because of the 1.0 / 0.5 literals, then narrowed */
static inline real ndc2pix(real v, int S) { return (real)((((double)v + 1.0) * S - 1.0) * 0.5); }

/* RST/auxiliary.h:46-56 */
static inline void get_rect(float px, float py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    int a;
    a = (int)((px - max_radius) / TILE_X); a = a > 0 ? a : 0; rmin[0] = gx < a ? gx : a;
    a = (int)((py - max_radius) / TILE_Y); a = a > 0 ? a : 0; rmin[1] = gy < a ? gy : a;
    a = (int)((px + max_radius + TILE_X - 1) / TILE_X); a = a > 0 ? a : 0; rmax[0] = gx < a ? gx : a;
    a = (int)((py + max_radius + TILE_Y - 1) / TILE_Y); a = a > 0 ? a : 0; rmax[1] = gy < a ? gy : a;
}

/* RST/auxiliary.h:22-39 */
static const float kSH0 = 0.28209479177387814f;
static const float kSH1 = 0.4886025119029199f;
static const float kSH2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f };
static const float kSH3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f };

/* RST/forward.cu:20-71.  out[3] is the colour BEFORE the max(.,0) clamp and after the +0.5. */
static void sh_to_rgb(int deg, const real pos[3], const float* campos, const float* sh /*[M][3]*/, real out[3])
{
    real d[3] = { pos[0] - (real)campos[0], pos[1] - (real)campos[1], pos[2] - (real)campos[2] };
    real len = (real)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    real x = d[0] / len, y = d[1] / len, z = d[2] / len;
    for (int c = 0; c < 3; c++) {
#define SHC(k) ((real)sh[(k) * 3 + c])
        real res = (real)kSH0 * SHC(0);
        if (deg > 0) {
            res = res - (real)kSH1 * y * SHC(1) + (real)kSH1 * z * SHC(2) - (real)kSH1 * x * SHC(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + (real)kSH2[0] * xy * SHC(4) + (real)kSH2[1] * yz * SHC(5) +
                      (real)kSH2[2] * ((real)2.0 * zz - xx - yy) * SHC(6) + (real)kSH2[3] * xz * SHC(7) +
                      (real)kSH2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    res = res + (real)kSH3[0] * y * ((real)3.0 * xx - yy) * SHC(9) +
                          (real)kSH3[1] * xy * z * SHC(10) +
                          (real)kSH3[2] * y * ((real)4.0 * zz - xx - yy) * SHC(11) +
                          (real)kSH3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy) * SHC(12) +
                          (real)kSH3[4] * x * ((real)4.0 * zz - xx - yy) * SHC(13) +
                          (real)kSH3[5] * z * (xx - yy) * SHC(14) +
                          (real)kSH3[6] * x * (xx - (real)3.0 * yy) * SHC(15);
                }
            }
        }
#undef SHC
        out[c] = res + (real)0.5;
    }
}
/* exported for the golden-vector test against the reference's utils/sh_utils.py:eval_sh */
void FN(sh_to_rgb)(int n, int deg, int M, const float* pos, const float* campos, const float* sh, real* out)
{
    for (int i = 0; i < n; i++) {
        real p[3] = { (real)pos[3 * i], (real)pos[3 * i + 1], (real)pos[3 * i + 2] };
        sh_to_rgb(deg, p, campos, sh + (size_t)i * M * 3, out + 3 * i);
    }
}

/* RST/forward.cu:118-152 (quaternion is used as given, normalisation is commented out at :127) */
static void cov3d_from_scale_rot(const real s_in[3], real mod, const real q[4], real cov6[6], mat3* Mout)
{
    real r = q[0], x = q[1], y = q[2], z = q[3];
    mat3 R;
    R.m[0][0] = (real)1.0 - (real)2.0 * (y * y + z * z); R.m[0][1] = (real)2.0 * (x * y - r * z); R.m[0][2] = (real)2.0 * (x * z + r * y);
    R.m[1][0] = (real)2.0 * (x * y + r * z); R.m[1][1] = (real)1.0 - (real)2.0 * (x * x + z * z); R.m[1][2] = (real)2.0 * (y * z - r * x);
    R.m[2][0] = (real)2.0 * (x * z - r * y); R.m[2][1] = (real)2.0 * (y * z + r * x); R.m[2][2] = (real)1.0 - (real)2.0 * (x * x + y * y);
    real s[3] = { mod * s_in[0], mod * s_in[1], mod * s_in[2] };
    mat3 M; /* M = S * R with S diagonal: M[c][r] = s[r] * R[c][r] */
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) M.m[c][rr] = s[rr] * R.m[c][rr];
    /* Sigma = transpose(M) * M : Sigma[c][r] = sum_k M[r][k] * M[c][k] */
#define SIG(c, r) (M.m[r][0] * M.m[c][0] + M.m[r][1] * M.m[c][1] + M.m[r][2] * M.m[c][2])
    cov6[0] = SIG(0, 0); cov6[1] = SIG(0, 1); cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1); cov6[4] = SIG(1, 2); cov6[5] = SIG(2, 2);
#undef SIG
    if (Mout) *Mout = M;
}

/* shared by forward (RST/forward.cu:74-113) and backward (RST/backward.cu:165-197): the clamped
 * view-space mean t, the 2x3 part of T = W*J (T.m[c][r], c<2) and cov2D (a,b,c) incl. +0.3. */
typedef struct { real t[3]; real txtz, tytz, limx, limy; mat3 T; mat3 W; mat3 V; real a, b, c; } cov2d_t;

static void cov2d_eval(const real mean[3], real fx, real fy, real tanx, real tany, const real cov6[6],
                       const float* view, cov2d_t* o)
{
    real t[3];
    xform4x3(mean, view, t);
    o->limx = (real)1.3f * tanx; o->limy = (real)1.3f * tany;
    o->txtz = t[0] / t[2]; o->tytz = t[1] / t[2];
    real cx = o->txtz < -o->limx ? -o->limx : o->txtz; cx = o->limx < cx ? o->limx : cx; /* min(lim,max(-lim,.)) */
    real cy = o->tytz < -o->limy ? -o->limy : o->tytz; cy = o->limy < cy ? o->limy : cy;
    t[0] = cx * t[2]; t[1] = cy * t[2];
    o->t[0] = t[0]; o->t[1] = t[1]; o->t[2] = t[2];
    real J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    real J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    mat3 W;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) W.m[c][r] = (real)view[4 * r + c]; /* W[c] = (v[c], v[4+c], v[8+c]) */
    mat3 T;
    for (int r = 0; r < 3; r++) {
        T.m[0][r] = W.m[0][r] * J00 + W.m[2][r] * J02;
        T.m[1][r] = W.m[1][r] * J11 + W.m[2][r] * J12;
        T.m[2][r] = (real)0.0;
    }
    mat3 V;
    V.m[0][0] = cov6[0]; V.m[0][1] = cov6[1]; V.m[0][2] = cov6[2];
    V.m[1][0] = cov6[1]; V.m[1][1] = cov6[3]; V.m[1][2] = cov6[4];
    V.m[2][0] = cov6[2]; V.m[2][1] = cov6[4]; V.m[2][2] = cov6[5];
    /* cov = transpose(T) * transpose(V) * T, only the upper-left 2x2 is needed:
     * A[k][r] = T[r][0]*V[0][k] + T[r][1]*V[1][k] + T[r][2]*V[2][k]   (A = Tt * Vt)
     * cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2] */
    real A[3][2];
    for (int k = 0; k < 3; k++)
        for (int r = 0; r < 2; r++)
            A[k][r] = T.m[r][0] * V.m[0][k] + T.m[r][1] * V.m[1][k] + T.m[r][2] * V.m[2][k];
    real c00 = A[0][0] * T.m[0][0] + A[1][0] * T.m[0][1] + A[2][0] * T.m[0][2];
    real c01 = A[0][1] * T.m[0][0] + A[1][1] * T.m[0][1] + A[2][1] * T.m[0][2];
    real c11 = A[0][1] * T.m[1][0] + A[1][1] * T.m[1][1] + A[2][1] * T.m[1][2];
    o->a = c00 + (real)0.3f; o->b = c01; o->c = c11 + (real)0.3f;
    o->T = T; o->W = W; o->V = V;
}

/* ------------------------------------------------------------------------------------------ */
/* K1: RST/forward.cu:155-256 (+ in_frustum RST/auxiliary.h:139-164).
 * forced_radii / forced_clamped: NULL in the fp32 build.  In the fp64 build they carry the fp32
 * build's discrete decisions (radius incl. culling, colour clamp flags). */
int FN(preprocess)(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                   const float* rotations, const float* opacities, const float* shs,
                   const float* cov3D_precomp, const float* colors_precomp, const float* view,
                   const float* proj, const float* campos, int W, int H, float tanfovx, float tanfovy,
                   const int* forced_radii, const uint8_t* forced_clamped, int* radii, real* means2D,
                   real* depths, real* cov3D, real* rgb, real* conic_opacity, uint8_t* clamped,
                   uint32_t* tiles_touched)
{
    const real fy = (real)H / ((real)2.0 * (real)tanfovy); /* RST/rasterizer_impl.cu:222-223 */
    const real fx = (real)W / ((real)2.0 * (real)tanfovx);
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        real p[3] = { (real)means3D[3 * i], (real)means3D[3 * i + 1], (real)means3D[3 * i + 2] };
        real ph[4], pv[3];
        xform4x4(p, proj, ph);
        real pw = (real)1.0 / (ph[3] + (real)0.0000001f);
        real pp[3] = { ph[0] * pw, ph[1] * pw, ph[2] * pw };
        xform4x3(p, view, pv);
        if (forced_radii) { if (!(forced_radii[i] > 0)) continue; }
        else if (pv[2] <= (real)0.2f) continue;

        real c6[6];
        if (cov3D_precomp) for (int k = 0; k < 6; k++) c6[k] = (real)cov3D_precomp[6 * i + k];
        else {
            real s[3] = { (real)scales[3 * i], (real)scales[3 * i + 1], (real)scales[3 * i + 2] };
            real q[4] = { (real)rotations[4 * i], (real)rotations[4 * i + 1], (real)rotations[4 * i + 2], (real)rotations[4 * i + 3] };
            cov3d_from_scale_rot(s, (real)scale_modifier, q, c6, 0);
            for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c6[k];
        }
        cov2d_t cv;
        cov2d_eval(p, fx, fy, (real)tanfovx, (real)tanfovy, c6, view, &cv);
        real det = cv.a * cv.c - cv.b * cv.b;
        if (!forced_radii && det == (real)0.0) continue;
        real det_inv = (real)1.0 / det;
        real conic[3] = { cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv };
        real mid = (real)0.5 * (cv.a + cv.c);
        real disc = mid * mid - det; disc = disc < (real)0.1f ? (real)0.1f : disc;
        real l1 = mid + (real)sqrt(disc), l2 = mid - (real)sqrt(disc);
        real lmax = l1 < l2 ? l2 : l1;
        real my_radius = (real)ceil((real)3.0 * (real)sqrt(lmax));
        real px = ndc2pix(pp[0], W), py = ndc2pix(pp[1], H);
        int rad = forced_radii ? forced_radii[i] : (int)my_radius;
        int rmin[2], rmax[2];
        get_rect((float)px, (float)py, rad, gx, gy, rmin, rmax);
        if (!forced_radii && (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (!colors_precomp) {
            real col[3];
            sh_to_rgb(D, p, campos, shs + (size_t)i * M * 3, col);
            for (int c = 0; c < 3; c++) {
                int cl = forced_clamped ? forced_clamped[3 * i + c] : (col[c] < (real)0.0);
                clamped[3 * i + c] = (uint8_t)cl;
                rgb[3 * i + c] = forced_clamped ? (cl ? (real)0.0 : col[c]) : (col[c] < (real)0.0 ? (real)0.0 : col[c]);
            }
        }
        depths[i] = pv[2];
        radii[i] = rad;
        means2D[2 * i] = px; means2D[2 * i + 1] = py;
        conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2]; conic_opacity[4 * i + 3] = (real)opacities[i];
        tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
    return 0;
}

/* K0: RST/rasterizer_impl.cu:54-66 */
void FN(mark_visible)(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    (void)proj;
    for (int i = 0; i < P; i++) {
        real p[3] = { (real)means3D[3 * i], (real)means3D[3 * i + 1], (real)means3D[3 * i + 2] }, pv[3];
        xform4x3(p, view, pv);
        present[i] = pv[2] > (real)0.2f;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Binning -- integer work, fp32 build only semantics (inputs are the fp32 means2D / depths).
 * C1 scan RST/rasterizer_impl.cu:277; getHigherMsb :35-50; K2 duplicateWithKeys :70-111;
 * C2 stable ascending sort on the low 32+bit key bits :301-309 (CUB radix sort == any stable
 * sort); K3 identifyTileRanges :116-138 after the memset at :311. */
uint32_t FN(higher_msb)(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/* returns R = number of (Gaussian, tile) instances; fills point_offsets (inclusive scan) */
int64_t FN(bin_count)(int P, const uint32_t* tiles_touched, uint32_t* point_offsets)
{
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return (int64_t)acc;
}

static void stable_sort_pairs(uint64_t* k, uint32_t* v, uint64_t* k2, uint32_t* v2, size_t n, int nbits)
{ /* LSD radix, 8 bits per pass: stable, ascending on the low nbits */
    for (int shift = 0; shift < nbits; shift += 8) {
        size_t cnt[257] = { 0 };
        for (size_t i = 0; i < n; i++) cnt[((k[i] >> shift) & 0xFF) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; i++) { size_t d = cnt[(k[i] >> shift) & 0xFF]++; k2[d] = k[i]; v2[d] = v[i]; }
        memcpy(k, k2, n * sizeof(uint64_t)); memcpy(v, v2, n * sizeof(uint32_t));
    }
}

void FN(bin_fill)(int P, int W, int H, const float* means2D, const float* depths, const int* radii,
                  const uint32_t* point_offsets, int64_t R, uint64_t* keys_unsorted, uint64_t* keys_sorted,
                  uint32_t* point_list, uint32_t* ranges /* [T][2] */)
{
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    uint32_t* vals = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        uint32_t off = i == 0 ? 0 : point_offsets[i - 1];
        int rmin[2], rmax[2];
        get_rect(means2D[2 * i], means2D[2 * i + 1], radii[i], gx, gy, rmin, rmax);
        uint32_t dbits; memcpy(&dbits, &depths[i], 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32; key |= dbits;
                keys_unsorted[off] = key; vals[off] = (uint32_t)i; off++;
            }
    }
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
    uint32_t* v2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
    memcpy(keys_sorted, keys_unsorted, sizeof(uint64_t) * (size_t)R);
    int bit = (int)FN(higher_msb)((uint32_t)(gx * gy));
    stable_sort_pairs(keys_sorted, vals, k2, v2, (size_t)R, 32 + bit);
    memcpy(point_list, vals, sizeof(uint32_t) * (size_t)R);
    free(k2); free(v2); free(vals);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (int64_t i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Canonical per-(pixel, Gaussian) arithmetic of the blend kernels.  The reference writes
 *   power = -0.5f*(con.x*d.x*d.x + con.z*d.y*d.y) - con.y*d.x*d.y      (RST/forward.cu:338)
 * and leaves FMA contraction to nvcc; this build fixes one contraction so that CPU and GPU agree:
 *   q = fma(con.z*d.y, d.y, (con.x*d.x)*d.x);  power = fma(-0.5, q, -((con.y*d.x)*d.y)). */
static inline float power_f(float cx, float cy, float cz, float dx, float dy)
{
    float q = fmaf(cz * dy, dy, (cx * dx) * dx);
    return fmaf(-0.5f, q, -((cy * dx) * dy));
}
static inline real power_r(real cx, real cy, real cz, real dx, real dy)
{
    if (sizeof(real) == sizeof(float)) return (real)power_f((float)cx, (float)cy, (float)cz, (float)dx, (float)dy);
    real q = (cz * dy) * dy + (cx * dx) * dx;
    return (real)-0.5 * q - (cy * dx) * dy;
}

/* K4: RST/forward.cu:261-393.  Arrays suffixed 32 are the fp32 build's geometry: they drive the
 * control flow (skip / terminate / median tests) in both builds. */
void FN(blend_forward)(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                       const real* means2D, const real* rgb, const real* conic_opacity, const real* depths,
                       const float* means2D32, const float* conic_opacity32, const float* bg,
                       real* out_color, real* out_depth, real* final_T, uint32_t* n_contrib,
                       uint32_t* pair_hash /* optional [H*W]: signature of which list entries contributed */)
{
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE_Y; ly++)
            for (int lx = 0; lx < TILE_X; lx++) {
                const int px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
                if (px >= W || py >= H) continue;
                const float pxf = (float)px, pyf = (float)py;
                real T = (real)1.0, C[3] = { 0, 0, 0 }, Dm = (real)15.0f; /* :308 median-depth default */
                float Tc = 1.0f;
                uint32_t contributor = 0, last = 0, hsh = 0;
                for (uint32_t j = r0; j < r1; j++) {
                    contributor++;
                    const uint32_t g = point_list[j];
                    /* control (fp32) */
                    float dxc = means2D32[2 * g] - pxf, dyc = means2D32[2 * g + 1] - pyf;
                    const float* co = conic_opacity32 + 4 * g;
                    float pwc = power_f(co[0], co[1], co[2], dxc, dyc);
                    if (pwc > 0.0f) continue;
                    float ac = co[3] * ctl_expf(pwc); ac = ac < 0.99f ? ac : 0.99f;
                    if (ac < 1.0f / 255.0f) continue;
                    float test_Tc = Tc * (1.0f - ac);
                    if (test_Tc < 0.0001f) break; /* done = true; :352-356 */
                    /* values */
                    real dx = means2D[2 * g] - (real)pxf, dy = means2D[2 * g + 1] - (real)pyf;
                    const real* cr = conic_opacity + 4 * g;
                    real pw = power_r(cr[0], cr[1], cr[2], dx, dy);
                    real a = cr[3] * real_exp(pw, pwc); a = a < (real)0.99f ? a : (real)0.99f;
                    real test_T = T * ((real)1.0 - a);
                    /* forward.cu:361  C[ch] += features[...] * alpha * T  -- the source's association is (feature * alpha) * T;
                       the add is contracted with the last multiply (the one fixed contraction of this statement) */
                    for (int ch = 0; ch < 3; ch++) {
                        if (sizeof(real) == sizeof(float)) C[ch] = (real)fmaf((float)rgb[3 * g + ch] * (float)a, (float)T, (float)C[ch]);
                        else C[ch] += rgb[3 * g + ch] * a * T;
                    }
                    if (Tc > 0.5f && test_Tc < 0.5f) Dm = depths[g]; /* :368-372 */
                    T = test_T; Tc = test_Tc;
                    last = contributor;
                    hsh = hsh * 31u + contributor * 2654435761u;
                }
                const size_t pid = (size_t)W * py + px;
                final_T[pid] = T; n_contrib[pid] = last;
                if (pair_hash) pair_hash[pid] = hsh;
                for (int ch = 0; ch < 3; ch++) {
                    if (sizeof(real) == sizeof(float)) out_color[(size_t)ch * H * W + pid] = (real)fmaf((float)T, bg[ch], (float)C[ch]);
                    else out_color[(size_t)ch * H * W + pid] = C[ch] + T * (real)bg[ch];
                }
                out_depth[pid] = Dm;
            }
    }
}

/* K5: RST/backward.cu:399-557.  Accumulates per instance (tile-local, pixel-major order), then
 * folds instances into per-Gaussian sums in sorted-list order, so the fp32 result is
 * deterministic.  Outputs must be zero on entry: dL_dmean2D [P][3] (.x,.y written),
 * dL_dconic [P][4] (.x,.y,.w written), dL_dopacity [P], dL_dcolors [P][3]. */
void FN(blend_backward)(int P, int W, int H, int64_t R, const uint32_t* ranges, const uint32_t* point_list,
                        const float* bg, const real* means2D, const real* conic_opacity, const real* colors,
                        const float* means2D32, const float* conic_opacity32, const real* final_T,
                        const uint32_t* n_contrib, const float* dL_dpix, real* dL_dmean2D, real* dL_dconic,
                        real* dL_dopacity, real* dL_dcolors)
{
    (void)P;
    const int gx = (W + TILE_X - 1) / TILE_X, gy = (H + TILE_Y - 1) / TILE_Y;
    real* inst = (real*)calloc((size_t)(R > 0 ? R : 1) * 9, sizeof(real));
    const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < TILE_Y; ly++)
            for (int lx = 0; lx < TILE_X; lx++) {
                const int px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
                if (px >= W || py >= H) continue;
                const size_t pid = (size_t)W * py + px;
                const float pxf = (float)px, pyf = (float)py;
                const real T_final = final_T[pid];
                real T = T_final;
                const uint32_t last = n_contrib[pid];
                real accum[3] = { 0, 0, 0 }, last_color[3] = { 0, 0, 0 }, last_alpha = 0;
                real dpx[3];
                for (int ch = 0; ch < 3; ch++) dpx[ch] = (real)dL_dpix[(size_t)ch * H * W + pid];
                real bg_dot = 0;
                for (int ch = 0; ch < 3; ch++) bg_dot += (real)bg[ch] * dpx[ch];
                /* back to front; entries at list position >= last are skipped (:486-488) */
                for (uint32_t jj = (last < r1 - r0 ? last : r1 - r0); jj-- > 0;) {
                    const uint32_t j = r0 + jj;
                    const uint32_t g = point_list[j];
                    float dxc = means2D32[2 * g] - pxf, dyc = means2D32[2 * g + 1] - pyf;
                    const float* co = conic_opacity32 + 4 * g;
                    float pwc = power_f(co[0], co[1], co[2], dxc, dyc);
                    if (pwc > 0.0f) continue;
                    float Gc = ctl_expf(pwc);
                    float ac = co[3] * Gc; ac = ac < 0.99f ? ac : 0.99f;
                    if (ac < 1.0f / 255.0f) continue;

                    real dx = means2D[2 * g] - (real)pxf, dy = means2D[2 * g + 1] - (real)pyf;
                    const real* cr = conic_opacity + 4 * g;
                    real pw = power_r(cr[0], cr[1], cr[2], dx, dy);
                    real G = real_exp(pw, pwc);
                    real a = cr[3] * G; a = a < (real)0.99f ? a : (real)0.99f;
                    T = T / ((real)1.0 - a);
                    const real dch = a * T;
                    real dL_dalpha = 0;
                    real* o = inst + (size_t)j * 9;
                    for (int ch = 0; ch < 3; ch++) {
                        const real c = colors[3 * g + ch];
                        accum[ch] = last_alpha * last_color[ch] + ((real)1.0 - last_alpha) * accum[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum[ch]) * dpx[ch];
                        o[6 + ch] += dch * dpx[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = a;
                    dL_dalpha += (-T_final / ((real)1.0 - a)) * bg_dot;
                    const real dL_dG = cr[3] * dL_dalpha;
                    const real gdx = G * dx, gdy = G * dy;
                    const real dG_ddelx = -gdx * cr[0] - gdy * cr[1];
                    const real dG_ddely = -gdy * cr[2] - gdx * cr[1];
                    o[0] += dL_dG * dG_ddelx * ddelx_dx;
                    o[1] += dL_dG * dG_ddely * ddely_dy;
                    o[2] += (real)-0.5 * gdx * dx * dL_dG;
                    o[3] += (real)-0.5 * gdx * dy * dL_dG;
                    o[4] += (real)-0.5 * gdy * dy * dL_dG;
                    o[5] += G * dL_dalpha;
                }
            }
    }
    for (int64_t j = 0; j < R; j++) {
        const uint32_t g = point_list[j];
        const real* o = inst + (size_t)j * 9;
        dL_dmean2D[3 * g] += o[0]; dL_dmean2D[3 * g + 1] += o[1];
        dL_dconic[4 * g] += o[2]; dL_dconic[4 * g + 1] += o[3]; dL_dconic[4 * g + 3] += o[4];
        dL_dopacity[g] += o[5];
        dL_dcolors[3 * g] += o[6]; dL_dcolors[3 * g + 1] += o[7]; dL_dcolors[3 * g + 2] += o[8];
    }
    free(inst);
}

/* ------------------------------------------------------------------------------------------ */
/* RST/backward.cu:20-139 -- SH backward; writes dL_dsh[M][3] (first (deg+1)^2 rows) and ADDS the
 * view-direction part to dL_dmean. */
static void sh_backward(int deg, const real pos[3], const float* campos, const float* sh,
                        const uint8_t* clamped3, const real dL_dcolor[3], real dL_dmean[3], real* dL_dsh)
{
    real dorig[3] = { pos[0] - (real)campos[0], pos[1] - (real)campos[1], pos[2] - (real)campos[2] };
    real len = (real)sqrt(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
    real x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
    real g[3];
    for (int c = 0; c < 3; c++) g[c] = dL_dcolor[c] * (clamped3[c] ? (real)0.0 : (real)1.0);
    real dx[3] = { 0, 0, 0 }, dy[3] = { 0, 0, 0 }, dz[3] = { 0, 0, 0 };
#define SHV(k, c) ((real)sh[(k) * 3 + (c)])
#define PUT(k, w) for (int c = 0; c < 3; c++) dL_dsh[(k) * 3 + c] = (w) * g[c]
    PUT(0, (real)kSH0);
    if (deg > 0) {
        PUT(1, -(real)kSH1 * y); PUT(2, (real)kSH1 * z); PUT(3, -(real)kSH1 * x);
        for (int c = 0; c < 3; c++) {
            dx[c] = -(real)kSH1 * SHV(3, c); dy[c] = -(real)kSH1 * SHV(1, c); dz[c] = (real)kSH1 * SHV(2, c);
        }
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            PUT(4, (real)kSH2[0] * xy); PUT(5, (real)kSH2[1] * yz);
            PUT(6, (real)kSH2[2] * ((real)2.0 * zz - xx - yy));
            PUT(7, (real)kSH2[3] * xz); PUT(8, (real)kSH2[4] * (xx - yy));
            for (int c = 0; c < 3; c++) {
                dx[c] += (real)kSH2[0] * y * SHV(4, c) + (real)kSH2[2] * (real)2.0 * -x * SHV(6, c) +
                         (real)kSH2[3] * z * SHV(7, c) + (real)kSH2[4] * (real)2.0 * x * SHV(8, c);
                dy[c] += (real)kSH2[0] * x * SHV(4, c) + (real)kSH2[1] * z * SHV(5, c) +
                         (real)kSH2[2] * (real)2.0 * -y * SHV(6, c) + (real)kSH2[4] * (real)2.0 * -y * SHV(8, c);
                dz[c] += (real)kSH2[1] * y * SHV(5, c) + (real)kSH2[2] * (real)2.0 * (real)2.0 * z * SHV(6, c) +
                         (real)kSH2[3] * x * SHV(7, c);
            }
            if (deg > 2) {
                PUT(9, (real)kSH3[0] * y * ((real)3.0 * xx - yy));
                PUT(10, (real)kSH3[1] * xy * z);
                PUT(11, (real)kSH3[2] * y * ((real)4.0 * zz - xx - yy));
                PUT(12, (real)kSH3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy));
                PUT(13, (real)kSH3[4] * x * ((real)4.0 * zz - xx - yy));
                PUT(14, (real)kSH3[5] * z * (xx - yy));
                PUT(15, (real)kSH3[6] * x * (xx - (real)3.0 * yy));
                for (int c = 0; c < 3; c++) {
                    dx[c] += ((real)kSH3[0] * SHV(9, c) * (real)3.0 * (real)2.0 * xy +
                              (real)kSH3[1] * SHV(10, c) * yz +
                              (real)kSH3[2] * SHV(11, c) * (real)-2.0 * xy +
                              (real)kSH3[3] * SHV(12, c) * (real)-3.0 * (real)2.0 * xz +
                              (real)kSH3[4] * SHV(13, c) * ((real)-3.0 * xx + (real)4.0 * zz - yy) +
                              (real)kSH3[5] * SHV(14, c) * (real)2.0 * xz +
                              (real)kSH3[6] * SHV(15, c) * (real)3.0 * (xx - yy));
                    dy[c] += ((real)kSH3[0] * SHV(9, c) * (real)3.0 * (xx - yy) +
                              (real)kSH3[1] * SHV(10, c) * xz +
                              (real)kSH3[2] * SHV(11, c) * ((real)-3.0 * yy + (real)4.0 * zz - xx) +
                              (real)kSH3[3] * SHV(12, c) * (real)-3.0 * (real)2.0 * yz +
                              (real)kSH3[4] * SHV(13, c) * (real)-2.0 * xy +
                              (real)kSH3[5] * SHV(14, c) * (real)-2.0 * yz +
                              (real)kSH3[6] * SHV(15, c) * (real)-3.0 * (real)2.0 * xy);
                    dz[c] += ((real)kSH3[1] * SHV(10, c) * xy +
                              (real)kSH3[2] * SHV(11, c) * (real)4.0 * (real)2.0 * yz +
                              (real)kSH3[3] * SHV(12, c) * (real)3.0 * ((real)2.0 * zz - xx - yy) +
                              (real)kSH3[4] * SHV(13, c) * (real)4.0 * (real)2.0 * xz +
                              (real)kSH3[5] * SHV(14, c) * (xx - yy));
                }
            }
        }
    }
#undef PUT
#undef SHV
    real dd[3] = { dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2], dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2],
                   dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2] };
    /* dnormvdv, RST/auxiliary.h:107-117 */
    real sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
    real inv32 = (real)1.0 / (real)sqrt(sum2 * sum2 * sum2);
    dL_dmean[0] += ((+sum2 - dorig[0] * dorig[0]) * dd[0] - dorig[1] * dorig[0] * dd[1] - dorig[2] * dorig[0] * dd[2]) * inv32;
    dL_dmean[1] += (-dorig[0] * dorig[1] * dd[0] + (sum2 - dorig[1] * dorig[1]) * dd[1] - dorig[2] * dorig[1] * dd[2]) * inv32;
    dL_dmean[2] += (-dorig[0] * dorig[2] * dd[0] - dorig[1] * dorig[2] * dd[1] + (sum2 - dorig[2] * dorig[2]) * dd[2]) * inv32;
}

/* K6 + K7: RST/backward.cu:144-274 (computeCov2DCUDA), :346-396 (preprocessCUDA),
 * :278-341 (computeCov3D).  Every output row of a culled Gaussian is left untouched (callers
 * zero-fill, RST/../rasterize_points.cu:150-158). */
void FN(preprocess_backward)(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rotations,
                             float scale_modifier, const real* cov3D /* [P][6] */, const float* view,
                             const float* proj, int W, int H, float tanfovx, float tanfovy,
                             const float* campos, const real* dL_dmean2D, const real* dL_dconic,
                             const real* dL_dcolor, real* dL_dmeans3D, real* dL_dcov3D, real* dL_dsh,
                             real* dL_dscale, real* dL_drot)
{
    const real fy = (real)H / ((real)2.0 * (real)tanfovy), fx = (real)W / ((real)2.0 * (real)tanfovx);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        real mean[3] = { (real)means3D[3 * i], (real)means3D[3 * i + 1], (real)means3D[3 * i + 2] };
        /* ---- K6 ---- */
        cov2d_t cv;
        cov2d_eval(mean, fx, fy, (real)tanfovx, (real)tanfovy, cov3D + 6 * i, view, &cv);
        /* clamp masks: decided in fp32 in both builds */
        float xgm, ygm;
        {
            float mf[3] = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] }, tf[3];
            for (int r = 0; r < 3; r++) tf[r] = view[r] * mf[0] + view[4 + r] * mf[1] + view[8 + r] * mf[2] + view[12 + r];
            float lx = 1.3f * tanfovx, ly = 1.3f * tanfovy, a = tf[0] / tf[2], b = tf[1] / tf[2];
            xgm = (a < -lx || a > lx) ? 0.f : 1.f; ygm = (b < -ly || b > ly) ? 0.f : 1.f;
        }
        const real a = cv.a, b = cv.b, c = cv.c;
        const real dcx = dL_dconic[4 * i], dcy = dL_dconic[4 * i + 1], dcz = dL_dconic[4 * i + 3];
        const real denom = a * c - b * b;
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        const real denom2inv = (real)1.0 / ((denom * denom) + (real)0.0000001f);
        const mat3* T = &cv.T; const mat3* V = &cv.V; const mat3* Wm = &cv.W;
        real* dcov = dL_dcov3D + 6 * i;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + (real)2.0 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + (real)2.0 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * (real)2.0 * (b * c * dcx - (denom + (real)2.0 * b * b) * dcy + a * b * dcz);
            dcov[0] = (T->m[0][0] * T->m[0][0] * dL_da + T->m[0][0] * T->m[1][0] * dL_db + T->m[1][0] * T->m[1][0] * dL_dc);
            dcov[3] = (T->m[0][1] * T->m[0][1] * dL_da + T->m[0][1] * T->m[1][1] * dL_db + T->m[1][1] * T->m[1][1] * dL_dc);
            dcov[5] = (T->m[0][2] * T->m[0][2] * dL_da + T->m[0][2] * T->m[1][2] * dL_db + T->m[1][2] * T->m[1][2] * dL_dc);
            dcov[1] = (real)2.0 * T->m[0][0] * T->m[0][1] * dL_da + (T->m[0][0] * T->m[1][1] + T->m[0][1] * T->m[1][0]) * dL_db + (real)2.0 * T->m[1][0] * T->m[1][1] * dL_dc;
            dcov[2] = (real)2.0 * T->m[0][0] * T->m[0][2] * dL_da + (T->m[0][0] * T->m[1][2] + T->m[0][2] * T->m[1][0]) * dL_db + (real)2.0 * T->m[1][0] * T->m[1][2] * dL_dc;
            dcov[4] = (real)2.0 * T->m[0][2] * T->m[0][1] * dL_da + (T->m[0][1] * T->m[1][2] + T->m[0][2] * T->m[1][1]) * dL_db + (real)2.0 * T->m[1][1] * T->m[1][2] * dL_dc;
        } else {
            for (int k = 0; k < 6; k++) dcov[k] = 0;
        }
        /* dL/dT (upper 2x3) */
        real dT[2][3];
        for (int k = 0; k < 3; k++) {
            real r0 = T->m[0][0] * V->m[k][0] + T->m[0][1] * V->m[k][1] + T->m[0][2] * V->m[k][2];
            real r1 = T->m[1][0] * V->m[k][0] + T->m[1][1] * V->m[k][1] + T->m[1][2] * V->m[k][2];
            dT[0][k] = (real)2.0 * r0 * dL_da + r1 * dL_db;
            dT[1][k] = (real)2.0 * r1 * dL_dc + r0 * dL_db;
        }
        real dJ00 = Wm->m[0][0] * dT[0][0] + Wm->m[0][1] * dT[0][1] + Wm->m[0][2] * dT[0][2];
        real dJ02 = Wm->m[2][0] * dT[0][0] + Wm->m[2][1] * dT[0][1] + Wm->m[2][2] * dT[0][2];
        real dJ11 = Wm->m[1][0] * dT[1][0] + Wm->m[1][1] * dT[1][1] + Wm->m[1][2] * dT[1][2];
        real dJ12 = Wm->m[2][0] * dT[1][0] + Wm->m[2][1] * dT[1][1] + Wm->m[2][2] * dT[1][2];
        real tz = (real)1.0 / cv.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real dtx = (real)xgm * -fx * tz2 * dJ02;
        real dty = (real)ygm * -fy * tz2 * dJ12;
        real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + ((real)2.0 * fx * cv.t[0]) * tz3 * dJ02 + ((real)2.0 * fy * cv.t[1]) * tz3 * dJ12;
        /* transformVec4x3Transpose, RST/auxiliary.h:89-97 */
        real dmean[3] = { (real)view[0] * dtx + (real)view[1] * dty + (real)view[2] * dtz,
                          (real)view[4] * dtx + (real)view[5] * dty + (real)view[6] * dtz,
                          (real)view[8] * dtx + (real)view[9] * dty + (real)view[10] * dtz };
        /* ---- K7 ---- */
        real mh[4];
        xform4x4(mean, proj, mh);
        real mw = (real)1.0 / (mh[3] + (real)0.0000001f);
        real mul1 = ((real)proj[0] * mean[0] + (real)proj[4] * mean[1] + (real)proj[8] * mean[2] + (real)proj[12]) * mw * mw;
        real mul2 = ((real)proj[1] * mean[0] + (real)proj[5] * mean[1] + (real)proj[9] * mean[2] + (real)proj[13]) * mw * mw;
        const real g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
        real add[3];
        add[0] = ((real)proj[0] * mw - (real)proj[3] * mul1) * g2x + ((real)proj[1] * mw - (real)proj[3] * mul2) * g2y;
        add[1] = ((real)proj[4] * mw - (real)proj[7] * mul1) * g2x + ((real)proj[5] * mw - (real)proj[7] * mul2) * g2y;
        add[2] = ((real)proj[8] * mw - (real)proj[11] * mul1) * g2x + ((real)proj[9] * mw - (real)proj[11] * mul2) * g2y;
        for (int k = 0; k < 3; k++) dmean[k] += add[k];
        if (shs) sh_backward(D, mean, campos, shs + (size_t)i * M * 3, clamped + 3 * i, dL_dcolor + 3 * i, dmean, dL_dsh + (size_t)i * M * 3);
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = dmean[k];
        if (scales) {
            real s[3] = { (real)scales[3 * i], (real)scales[3 * i + 1], (real)scales[3 * i + 2] };
            real q[4] = { (real)rotations[4 * i], (real)rotations[4 * i + 1], (real)rotations[4 * i + 2], (real)rotations[4 * i + 3] };
            real c6[6]; mat3 Mm;
            cov3d_from_scale_rot(s, (real)scale_modifier, q, c6, &Mm);
            real sm[3] = { (real)scale_modifier * s[0], (real)scale_modifier * s[1], (real)scale_modifier * s[2] };
            mat3 R; /* recover R from the same formulas (M = S*R) */
            {
                real r = q[0], x = q[1], y = q[2], z = q[3];
                R.m[0][0] = (real)1.0 - (real)2.0 * (y * y + z * z); R.m[0][1] = (real)2.0 * (x * y - r * z); R.m[0][2] = (real)2.0 * (x * z + r * y);
                R.m[1][0] = (real)2.0 * (x * y + r * z); R.m[1][1] = (real)1.0 - (real)2.0 * (x * x + z * z); R.m[1][2] = (real)2.0 * (y * z - r * x);
                R.m[2][0] = (real)2.0 * (x * z - r * y); R.m[2][1] = (real)2.0 * (y * z + r * x); R.m[2][2] = (real)1.0 - (real)2.0 * (x * x + y * y);
            }
            mat3 dSig;
            dSig.m[0][0] = dcov[0]; dSig.m[0][1] = (real)0.5 * dcov[1]; dSig.m[0][2] = (real)0.5 * dcov[2];
            dSig.m[1][0] = (real)0.5 * dcov[1]; dSig.m[1][1] = dcov[3]; dSig.m[1][2] = (real)0.5 * dcov[4];
            dSig.m[2][0] = (real)0.5 * dcov[2]; dSig.m[2][1] = (real)0.5 * dcov[4]; dSig.m[2][2] = dcov[5];
            mat3 M2;
            for (int cc = 0; cc < 3; cc++) for (int rr = 0; rr < 3; rr++) M2.m[cc][rr] = (real)2.0 * Mm.m[cc][rr];
            mat3 dM = mat3_mul(&M2, &dSig);
            mat3 Rt = mat3_t(&R), dMt = mat3_t(&dM);
            real* ds = dL_dscale + 3 * i;
            for (int k = 0; k < 3; k++) ds[k] = Rt.m[k][0] * dMt.m[k][0] + Rt.m[k][1] * dMt.m[k][1] + Rt.m[k][2] * dMt.m[k][2];
            for (int k = 0; k < 3; k++) for (int rr = 0; rr < 3; rr++) dMt.m[k][rr] *= sm[k];
            real r = q[0], x = q[1], y = q[2], z = q[3];
            real* dq = dL_drot + 4 * i;
#define D_(c, r_) dMt.m[c][r_]
            dq[0] = (real)2.0 * z * (D_(0, 1) - D_(1, 0)) + (real)2.0 * y * (D_(2, 0) - D_(0, 2)) + (real)2.0 * x * (D_(1, 2) - D_(2, 1));
            dq[1] = (real)2.0 * y * (D_(1, 0) + D_(0, 1)) + (real)2.0 * z * (D_(2, 0) + D_(0, 2)) + (real)2.0 * r * (D_(1, 2) - D_(2, 1)) - (real)4.0 * x * (D_(2, 2) + D_(1, 1));
            dq[2] = (real)2.0 * x * (D_(1, 0) + D_(0, 1)) + (real)2.0 * r * (D_(2, 0) - D_(0, 2)) + (real)2.0 * z * (D_(1, 2) + D_(2, 1)) - (real)4.0 * y * (D_(2, 2) + D_(0, 0));
            dq[3] = (real)2.0 * r * (D_(0, 1) - D_(1, 0)) + (real)2.0 * x * (D_(2, 0) + D_(0, 2)) + (real)2.0 * y * (D_(1, 2) + D_(2, 1)) - (real)4.0 * z * (D_(1, 1) + D_(0, 0));
#undef D_
        }
    }
}
