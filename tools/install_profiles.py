#!/usr/bin/env python
"""Copy a gpurun_out/profiles_<tag>/ set into profiles/ and derive profiles/pmc_blend_bwd.json
(the per-launch HBM traffic bench.py reports as roofline.traffic)."""
import json, os, re, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
suffix = sys.argv[2] if len(sys.argv) > 2 else ""          # e.g. "_3M": profiles/pmc_blend_bwd_3M.json (bench.py picks the file by #Gaussians)
src = os.path.join("gpurun_out", f"profiles_{tag}")
os.makedirs("profiles", exist_ok=True)
for f in os.listdir(src):
    if f.endswith((".txt", ".json", ".csv")) and os.path.getsize(os.path.join(src, f)) > 0:
        shutil.copy(os.path.join(src, f), os.path.join("profiles", f))
def per_launch(counter, kernel):
    p = os.path.join(src, f"{tag}_pmc_{counter}.txt")
    for line in open(p):
        if kernel in line:
            return float(line.split()[-2])          # (kernel, calls, value, all)
    return None
for kernel, pattern, stem in (("blend_bwd_cull_t_kernel", "blend_bwd_cull", "pmc_blend_bwd"), ("blend_fwd_cull_kernel", "blend_fwd_cull", "pmc_blend_fwd")):
    fetch_kb, write_kb = per_launch("FETCH_SIZE", pattern), per_launch("WRITE_SIZE", pattern)
    if fetch_kb is None or write_kb is None:
        continue
    d = {"kernel": kernel, "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
         "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> doubled "
                       "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported, in KB (calibrated in round 5 on late_rows_zero_kernel's "
                       "streaming fill of the same pass: 692 MB reported, 707 MB by arithmetic)",
         "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024), "source": f"profiles/{tag}_pmc_FETCH_SIZE.txt, {tag}_pmc_WRITE_SIZE.txt"}
    # SQ counters of the same kernel (separate --pmc pass): VALU instructions issued per launch
    sq = os.path.join(src, f"{tag}_pmc_SQ.txt")
    if os.path.exists(sq):
        lines = open(sq).read().splitlines()
        cols = lines[0].split()
        for line in lines[1:]:
            if pattern in line:
                vals = line.split()[-(len(cols) - 2):]            # the numeric columns after kernel name and calls (the last one, `all`, is a count)
                named = dict(zip(cols[2:], (float(v) for v in vals)))
                d["valu_wave_insts_per_launch"] = named.get("SQ_INSTS_VALU")
                d["salu_wave_insts_per_launch"] = named.get("SQ_INSTS_SALU")
                d["lds_wave_insts_per_launch"] = named.get("SQ_INSTS_LDS")
                d["sq_source"] = f"profiles/{tag}_pmc_SQ.txt"
                break
    ks = os.path.join(src, f"{tag}_kernel_stats.txt")
    if os.path.exists(ks):
        for line in open(ks):
            if pattern in line:
                m = re.findall(r"[-+]?\d*\.\d+|\d+", line)
                d["kernel_stats_line"] = line.strip()
                break
    bj = os.path.join(src, f"bench_{tag}_under_rocprof.json")
    try:
        d["gaussians"] = json.loads(open(bj).read().strip().splitlines()[-1])["config"]["gaussians"]
    except Exception:
        pass
    json.dump(d, open(os.path.join("profiles", f"{stem}{suffix}.json"), "w"), indent=1)
    print(d)
