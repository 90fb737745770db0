"""Dev helper: runs one fwd+bwd of the bench workload with the counter build (gpurun_variants/lib_counters.so)."""
import ctypes, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "saro-gs_amd", "diff_gaussian_rasterization_ch3", "libgsrast_hip.so")
shutil.copy(lib, "/tmp/orig.so"); shutil.copy(os.path.join(root, "gpurun_variants", "lib_counters.so"), lib)
try:
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd"))
    import torch, bench
    import diff_gaussian_rasterization_ch3 as rast
    import scenes
    from diff_gaussian_rasterization_ch3 import _C
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, 1, torch.device("cuda:0"))
    L = _C.lib()
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(int(os.environ.get('WARM', '1'))):
        wl.step(None, 1)
    torch.cuda.synchronize()
    L.gsrast_debug_counters(out, 1)
    import ctypes as C_
    init = (ctypes.c_ulonglong * 16)(); init[11] = 2**63
    wl.step(None, 1); torch.cuda.synchronize()
    L.gsrast_debug_counters(out, 1)
    v = list(out)
    print("RAW", v)
    print("fwd: survivor iterations %d, with a lane in range %d, with a contribution %d, contributing lanes %d (%.1f / iteration)" % (v[0], v[1], v[3], v[2], v[2] / max(v[1], 1)))
    print("bwd: survivor iterations %d, with a contribution %d, contributing lanes %d (%.1f / contributing iteration), lanes still in reach (pos < last) %.1f / iteration" % (v[4], v[5], v[6], v[6] / max(v[5], 1), v[7] / max(v[4], 1)))
    if v[10] and not os.environ.get("TIMING"):
        print("fwd, if one wave carried TWO instances (one per 8 x 4 half): %d iterations (%.2f of today's), FOUR (one per 8 x 2 quarter): %d (%.2f)" % (v[10], v[10] / max(v[0], 1), v[12], v[12] / max(v[0], 1)))
    elif v[10]:
        print("bwd blend: longest workgroup %.1f us (list prefix walked: %d entries), launch first start -> last end %.1f us" % (v[10] / 100.0, v[13], (v[12] - v[11]) / 100.0))
    if v[8]:
        print("bwd transposed phases %d, live instances in them %d (%.2f of 8); phases with ONE live instance %d (%.1f %%), with TWO %d (%.1f %%)" % (v[8], v[9], v[9] / v[8], v[14], 100.0 * v[14] / v[8], v[15], 100.0 * v[15] / v[8]))
finally:
    shutil.copy("/tmp/orig.so", lib)
