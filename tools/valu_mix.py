#!/usr/bin/env python
"""Static VALU instruction mix of the blend kernels' per-(wave, instance) loop, priced with the MEASURED issue cost of each
instruction class (profiles/r02_valu_calib.json, tools/valu_calib.hip) -> profiles/<tag>_valu_mix.json (tag = argv[1], default r04).

    python tools/valu_mix.py            # compiles csrc/gsrast_capi.hip with -save-temps into a temp dir, no GPU needed

bench.py uses `avg_cycles_per_valu_inst` of a kernel to turn its SQ_INSTS_VALU count into SIMD issue cycles; the per-pair
figures say where the cycles of one evaluated (wave, instance) pair go.  The loop region is the code between the `s_ff1_i32_b64`
that picks the next surviving instance and the batch's closing `s_barrier`; both sides of every branch inside it are counted
once (a static mix, not a trace)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "saro-gs_amd", "csrc", "gsrast_capi.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics",
         "-fno-slp-vectorize"]      # = saro-gs_amd/build.py
KERNELS = {"blend_bwd_cull_t_kernel": r"^_ZN6gsrast23blend_bwd_cull_t_kernelILi0E", "blend_bwd_cull_kernel": r"^_ZN6gsrast21blend_bwd_cull_kernelILi0ELi1E",
           "blend_fwd_cull_kernel": r"^_ZN6gsrast21blend_fwd_cull_kernelILi0E"}

# instruction classes: full rate (one wave64 instruction per ~2.4 cycles of a SIMD), half rate, quarter rate -- membership from
# the calibration run; mnemonics it did not cover fall into `half` (most of the ISA is half rate on this part)
FULL = ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_mov_b64",
        "v_and_b32", "v_xor_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_accvgpr")
QUARTER = ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_permlane")


def classify(mn, line):
    if "_dpp" in mn or " dpp" in line or "quad_perm" in line or "row_" in line:
        return "dpp"
    if mn.startswith("v_pk_"):
        return "packed"
    if mn.startswith(QUARTER):
        return "quarter"
    if mn.startswith("v_cndmask") and mn.endswith("_e32"):
        return "cndmask_vcc"
    if mn.startswith(FULL):
        return "full"
    return "half"


def cost_table(calib):
    best = {}
    for r in calib["results"]:
        if r["waves_per_simd"] == 8:
            best[r["op"]] = r["cycles_per_inst_per_simd_at_2.4GHz"]
    full = sum(best[k] for k in ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_fmac_f32", "v_mov_b32")) / 5
    half = sum(best[k] for k in ("v_cmp_lt_f32 (vcc)", "v_cndmask_b32_e64 (sgpr mask)", "v_min_f32", "v_cvt_i32_f32", "v_ldexp_f32", "v_rndne_f32")) / 6
    return {"full": full, "half": half, "dpp": best["v_add_f32_dpp quad_perm"], "quarter": best["v_rcp_f32"], "packed": best["v_pk_fma_f32"],
            # VOP2 v_cndmask with the implicit vcc operand: priced as the half-rate class here; the back-to-back microbenchmark of it
            # (23 cycles) is reported separately in the calibration file
            "cndmask_vcc": half}


def main():
    calib = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_calib.json")))
    cost = cost_table(calib)
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["hipcc"] + FLAGS + ["-save-temps", "-c", SRC, "-o", os.path.join(td, "x.o")], cwd=td,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "gsrast_capi-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    out = {"cycles_per_wave_instruction": {k: round(v, 3) for k, v in cost.items()}, "calibration": "profiles/r02_valu_calib.json (8 waves per SIMD)",
           "kernels": {}}
    for name, pat in KERNELS.items():
        start = next(i for i, l in enumerate(asm) if re.match(pat, l))
        end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
        body = [l for l in asm[start:end] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        ff1 = next(i for i, l in enumerate(body) if "s_ff1_i32_b" in l)
        bar = next(i for i in range(ff1, len(body)) if "s_barrier" in body[i])
        loop = body[ff1:bar]

        def mix(lines):
            m = {}
            for l in lines:
                mn = l.split()[0]
                if mn.startswith("v_"):
                    c = classify(mn, l)
                    m[c] = m.get(c, 0) + 1
            return m
        lm, km = mix(loop), mix(body)
        n = sum(lm.values())
        cyc = sum(cost[c] * k for c, k in lm.items())
        out["kernels"][name] = {
            "loop_valu_instructions": n, "loop_mix": lm, "loop_cycles_per_pass": round(cyc, 1),
            "avg_cycles_per_valu_inst": round(cyc / n, 3),
            "loop_other": {"salu": sum(1 for l in loop if l.split()[0].startswith("s_")), "lds": sum(1 for l in loop if l.split()[0].startswith("ds_"))},
            "whole_kernel_mix": km}
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_valu_mix.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
