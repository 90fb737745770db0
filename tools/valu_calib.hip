// valu_calib.hip -- measures the wave64 VALU issue rate of gfx950 (MI355X) per instruction class, so that bench.py's
// "issue-slot fraction" of the blend kernels rests on a measured constant rather than an assumed cycles-per-instruction.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o gpurun_out/valu_calib && gpurun_out/valu_calib > profiles/r02_valu_calib.json
//
// Every kernel runs CHAINS independent dependency chains per lane of ONE instruction (inline asm, so nothing is folded), UNROLL
// instructions per loop trip, on every SIMD of the chip with 1 / 2 / 4 / 8 waves per SIMD.  Reported per (op, waves per SIMD):
//   wave_insts_per_s           chip-wide wave-instructions retired per second
//   cycles_per_inst_per_simd   shader cycles one SIMD spends per wave-instruction = clock / (wave_insts_per_s / 1024 SIMDs),
//                              with the clock measured in the same launch (s_memtime ticks / wall time)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float float2v __attribute__((ext_vector_type(2)));

enum Op { FMA, MUL, ADD, PK_FMA, PK_MUL, EXP, RCP, LDEXP, RNDNE, ADD_DPP_SHR, ADD_DPP_QUAD, ADD_DPP_BCAST, CNDMASK, MOV_DPP, CMP, FMA_DEP1, BPERMUTE, READLANE, NOPS, CNDMASK_VCC, MIN, CVT_I32, SUB_U32, LSHL_ADD, CMP_SGPR, PERMLANE32, FMA_SALU, MOV, AND, MAX, LSHLREV, MAD_U24, FMAC, FMAMK, SUBREV, ADD_U32, XOR, MED3, MUL_LO, CMP_CND_VCC, CMP_CND_SGPR, SAND_CND_VCC, CND_VCC_SMOV, CXX_SELECT };
static const char* kNames[] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_ldexp_f32", "v_rndne_f32",
                                "v_add_f32_dpp row_shr:1", "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_bcast:15", "v_cndmask_b32_e64 (sgpr mask)", "v_mov_b32_dpp row_shr:1",
                                "v_cmp_lt_f32 (vcc)", "v_fma_f32 one dependent chain", "ds_bpermute_b32", "v_readlane_b32", "s_nop 0", "v_cndmask_b32_e32 (vcc)", "v_min_f32", "v_cvt_i32_f32", "v_sub_u32", "v_lshl_add_u32",
                                "v_cmp_lt_f32_e64 (sgpr pair)", "v_permlane32_swap", "v_fma_f32 + s_add_u32 interleaved (counted: the v_fma)", "v_mov_b32", "v_and_b32", "v_max_f32", "v_lshlrev_b32", "v_mad_u32_u24",
                                "v_fmac_f32", "v_fmamk_f32", "v_subrev_f32", "v_add_u32", "v_xor_b32", "v_med3_f32", "v_mul_lo_u32",
                                "pair: v_cmp_lt_f32 vcc + v_cndmask_b32_e32 vcc (counted: pairs)", "pair: v_cmp_lt_f32_e64 sgpr + v_cndmask_b32_e64 sgpr (counted: pairs)",
                                "pair: s_and_b64 vcc + v_cndmask_b32_e32 vcc (counted: pairs)", "v_cndmask_b32_e32 vcc (vcc written once by s_mov_b64)",
                                "C++ select a = a < b ? a * c : a + d (counted: selects; compiler's code)" };

constexpr int CHAINS = 8, UNROLL = 64;

template <int OP>
__global__ void __launch_bounds__(256) calib_kernel(float* out, int iters, unsigned long long* ticks)
{
    float a[CHAINS];
    float2v p[CHAINS];
#pragma unroll
    for (int u = 0; u < CHAINS; u++) { a[u] = 1.0f + 0.001f * (float)(threadIdx.x + u); p[u] = float2v{ a[u], a[u] + 0.5f }; }
    const float b = 0.9999f, c = 1.0e-4f;
    const float2v pb = { b, b }, pc = { c, c };
    const int idx = (int)((threadIdx.x * 4u) ^ 4u);
    const unsigned long long selmask = 0x5555555555555555ull ^ (unsigned long long)iters;
    unsigned sacc = 0;
    if constexpr (OP == CND_VCC_SMOV) asm volatile("s_mov_b64 vcc, %0" :: "s"(selmask) : "vcc");
    else asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(1.01f) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < UNROLL / CHAINS; r++) {
#pragma unroll
            for (int u = 0; u < CHAINS; u++) {
                if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                else if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                else if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c));
                else if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pb), "v"(pc));
                else if constexpr (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pb));
                else if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[u]));
                else if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u]));
                else if constexpr (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[u]) : "v"(0));
                else if constexpr (OP == RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[u]));
                else if constexpr (OP == ADD_DPP_SHR) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
                else if constexpr (OP == ADD_DPP_QUAD) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
                else if constexpr (OP == ADD_DPP_BCAST) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
                else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "s"(selmask));
                else if constexpr (OP == CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : );
                else if constexpr (OP == MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                else if constexpr (OP == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[u]));
                else if constexpr (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[u]) : "v"(1));
                else if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[u]) : "v"(1));
                else if constexpr (OP == CMP_SGPR) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[u]), "v"(b)); asm volatile("" :: "s"(m)); }
                else if constexpr (OP == PERMLANE32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[u]), "+v"(a[(u + 1) % CHAINS]));
                else if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[u]) : "v"(b));
                else if constexpr (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[u]) : "v"(0x7fffffff));
                else if constexpr (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[u]) : "v"(b));
                else if constexpr (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[u]));
                else if constexpr (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(3), "v"(1));
                else if constexpr (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                else if constexpr (OP == FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fff58, %1" : "+v"(a[u]) : "v"(c));
                else if constexpr (OP == SUBREV) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[u]) : "v"(c));
                else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(1));
                else if constexpr (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[u]) : "v"(1));
                else if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c));
                else if constexpr (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(3));
                else if constexpr (OP == CMP_CND_VCC) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[u]) : "v"(b), "v"(c) : "vcc"); }
                else if constexpr (OP == CMP_CND_SGPR) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[u]), "=&s"(m) : "v"(b), "v"(c)); }
                else if constexpr (OP == SAND_CND_VCC) { asm volatile("s_and_b64 vcc, %1, exec\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(a[u]) : "s"(selmask), "v"(c) : "vcc"); }
                else if constexpr (OP == CND_VCC_SMOV) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(b) : );
                else if constexpr (OP == CXX_SELECT) { a[u] = a[u] < b ? a[u] * 0.999f : a[u] + c; }
                else if constexpr (OP == FMA_SALU) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(b), "v"(c)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); }
                else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[u]));
                else if constexpr (OP == CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[u]), "v"(b) : "vcc");
                else if constexpr (OP == FMA_DEP1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                else if constexpr (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[u]) : "v"(idx));
                else if constexpr (OP == READLANE) { int sr; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(sr) : "v"(a[u])); asm volatile("" :: "s"(sr)); }
                else if constexpr (OP == NOPS) asm volatile("s_nop 0");
            }
        }
        if constexpr (OP == BPERMUTE) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < CHAINS; u++) s += a[u] + p[u].x + p[u].y;
    if (s == 123.456f || sacc == 0xFFFFFFFFu) out[0] = s;                       // keeps the chains alive, never true
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int OP>
void run(int wps, int iters, float* d_out, unsigned long long* d_ticks, bool first)
{
    const int grid = 256 * wps;      // 256-thread blocks = 4 waves = one per SIMD of a CU; wps blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    calib_kernel<OP><<<grid, 256>>>(d_out, iters / 8, d_ticks);       // warm-up
    hipDeviceSynchronize();
    double best_ms = 1e30; unsigned long long ticks = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        calib_kernel<OP><<<grid, 256>>>(d_out, iters, d_ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_ms) { best_ms = ms; hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost); }
    }
    const double insts = (double)grid * 4.0 * (double)iters * UNROLL;      // wave-instructions
    const double rate = insts / (best_ms * 1e-3);
    // s_memtime ticks of block 0 over its own loop / the launch's wall time: ~2400 per us if the counter runs at the shader
    // clock, ~100 per us if it is the constant 100 MHz reference -- reported, not assumed
    const double ticks_per_us = (double)ticks / (best_ms * 1e3);
    const double cyc_nominal = 2.4e9 / (rate / 1024.0);
    printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"wave_insts_per_s\": %.4g, \"cycles_per_inst_per_simd_at_2.4GHz\": %.3f, \"kernel_ms\": %.4f, \"memtime_ticks_per_us\": %.1f}",
           first ? "" : ",\n", kNames[OP], wps, rate, cyc_nominal, best_ms, ticks_per_us);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int OP>
void sweep(float* d_out, unsigned long long* d_ticks, bool& first)
{
    const int w[] = { 1, 2, 4, 8 };
    for (int k = 0; k < 4; k++) { run<OP>(w[k], 4096, d_out, d_ticks, first); first = false; }
}

int main()
{
    float* d_out; unsigned long long* d_ticks;
    hipMalloc(&d_out, 256); hipMalloc(&d_ticks, 64);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"chains_per_lane\": %d,\n \"note\": \"wave64 instructions; 1024 SIMDs; cycles_per_inst_per_simd = 2.4e9 / (wave_insts_per_s / 1024)\",\n \"results\": [\n",
           prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, CHAINS);
    bool first = true;
    sweep<FMA>(d_out, d_ticks, first); sweep<MUL>(d_out, d_ticks, first); sweep<ADD>(d_out, d_ticks, first);
    sweep<PK_FMA>(d_out, d_ticks, first); sweep<PK_MUL>(d_out, d_ticks, first);
    sweep<EXP>(d_out, d_ticks, first); sweep<RCP>(d_out, d_ticks, first); sweep<LDEXP>(d_out, d_ticks, first); sweep<RNDNE>(d_out, d_ticks, first);
    sweep<ADD_DPP_SHR>(d_out, d_ticks, first); sweep<ADD_DPP_QUAD>(d_out, d_ticks, first); sweep<ADD_DPP_BCAST>(d_out, d_ticks, first);
    sweep<CNDMASK>(d_out, d_ticks, first); sweep<MOV_DPP>(d_out, d_ticks, first); sweep<CMP>(d_out, d_ticks, first);
    sweep<FMA_DEP1>(d_out, d_ticks, first); sweep<BPERMUTE>(d_out, d_ticks, first); sweep<READLANE>(d_out, d_ticks, first); sweep<NOPS>(d_out, d_ticks, first);
    sweep<CNDMASK_VCC>(d_out, d_ticks, first); sweep<MIN>(d_out, d_ticks, first); sweep<CVT_I32>(d_out, d_ticks, first); sweep<SUB_U32>(d_out, d_ticks, first);
    sweep<LSHL_ADD>(d_out, d_ticks, first); sweep<CMP_SGPR>(d_out, d_ticks, first); sweep<PERMLANE32>(d_out, d_ticks, first);
    sweep<MOV>(d_out, d_ticks, first); sweep<AND>(d_out, d_ticks, first); sweep<MAX>(d_out, d_ticks, first); sweep<LSHLREV>(d_out, d_ticks, first);
    sweep<MAD_U24>(d_out, d_ticks, first); sweep<FMAC>(d_out, d_ticks, first); sweep<FMAMK>(d_out, d_ticks, first); sweep<SUBREV>(d_out, d_ticks, first);
    sweep<ADD_U32>(d_out, d_ticks, first); sweep<XOR>(d_out, d_ticks, first); sweep<MED3>(d_out, d_ticks, first); sweep<MUL_LO>(d_out, d_ticks, first);
    sweep<CMP_CND_VCC>(d_out, d_ticks, first);
    // (pairs written as ONE asm statement in which the second instruction reads a mask the first has just written to an SGPR pair
    // or, through the scalar unit, to vcc -- "v_cmp_e64 s[N:N+1] + v_cndmask_e64 s[N:N+1]", "s_and_b64 vcc + v_cndmask_e32" -- did
    // not terminate on the GPU box: no compiler-inserted wait states inside an asm statement.  They are deliberately not swept.)
    printf("\n ]}\n");
    return 0;
}
