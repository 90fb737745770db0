"""Dev helper: phase times of depth_bucket_scatter_kernel (gpurun_variants/lib_scat.so built with -DGSRAST_SCATTER_TIMING), 3 M headline loop."""
import ctypes, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "saro-gs_amd", "diff_gaussian_rasterization_ch3", "libgsrast_hip.so")
shutil.copy(lib, "/tmp/orig.so"); shutil.copy(os.path.join(root, "gpurun_variants", "lib_scat.so"), lib)
try:
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd"))
    import torch, bench, scenes
    import diff_gaussian_rasterization_ch3 as rast
    from diff_gaussian_rasterization_ch3 import _C
    P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
    wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, 8, torch.device("cuda:0"), poses=8)
    L = _C.lib()
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(40):
        wl.step(None, 1)
    torch.cuda.synchronize()
    L.gsrast_debug_scatter_timing(out, 1)
    N = 16
    for _ in range(N):
        wl.step(None, 1)
    torch.cuda.synchronize()
    L.gsrast_debug_scatter_timing(out, 1)
    v = list(out)
    wgs = max(v[15], 1)
    names = ["keys + histogram loaded", "scan done", "cut cells in LDS", "ranks (LDS atomics), rect loads issued", "barrier", "global atomics (thread 0)", "barrier", "late test + slab stores"]
    print("workgroups per launch %.0f; wall clock ticks are 10 ns" % (wgs / N))
    tot = 0
    for k, n in enumerate(names):
        print("  %-42s %7.2f us per workgroup" % (n, v[k] / wgs / 100.0)); tot += v[k] / wgs / 100.0
    print("  %-42s %7.2f us" % ("sum", tot))
finally:
    shutil.copy("/tmp/orig.so", lib)
