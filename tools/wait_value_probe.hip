// wait_value_probe.hip -- does hipStreamWaitValue32 gate a stream on a word a KERNEL writes, and what does it cost next to a chain of
// empty dependent launches?  (development probe for the list cut's verified fallback; run under `timeout`)
//   hipcc --offload-arch=gfx950 -O2 tools/wait_value_probe.hip -o gpurun_out/wait_value_probe && timeout 60 gpurun_out/wait_value_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy_then_write(uint32_t* word, uint32_t value, long long spin_ticks, long long* t_written)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        *t_written = wall_clock64();
    }
}
__global__ void stamp(long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) *t = wall_clock64(); }
__global__ void empty_pred(const uint32_t* pred) { if (*pred == 0u) return; }

int main()
{
    int dev = 0, can = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
    uint32_t* sig = nullptr; uint32_t* plain = nullptr;
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    CK(hipMalloc((void**)&plain, 64));
    CK(hipMemset(plain, 0, 64));
    if (es == hipSuccess) CK(hipMemset(sig, 0, 8));
    long long* ts = nullptr;
    CK(hipHostMalloc((void**)&ts, 64 * sizeof(long long), hipHostMallocMapped));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, dev));
    const double tick_us = 1000.0 / clk_khz;
    printf("{\"can_use_stream_wait_value\": %d, \"signal_memory\": \"%s\", \"wall_clock_khz\": %d", can, es == hipSuccess ? "ok" : hipGetErrorString(es), clk_khz);
    for (int which = 0; which < 2; which++) {
        uint32_t* w = which == 0 ? sig : plain;
        if (!w) continue;
        double lat = 0; int n = 0;
        for (int it = 0; it < 20; it++) {
            const uint32_t v = 100u + it;
            busy_then_write<<<1, 64, 0, a>>>(w, v, (long long)(100.0 / tick_us), ts);          // writes after ~100 us
            hipError_t e = hipStreamWaitValue32(b, w, v, hipStreamWaitValueEq, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf(", \"wait_%s\": \"%s\"", which ? "plain" : "signal", hipGetErrorString(e)); break; }
            stamp<<<1, 64, 0, b>>>(ts + 1);
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            if (it >= 4) { lat += (ts[1] - ts[0]) * tick_us; n++; }
        }
        if (n) printf(", \"kernel_write_to_gated_kernel_start_us_%s\": %.2f", which ? "plain" : "signal", lat / n);
    }
    // a chain of ten predicated-off launches between two stamps, on one stream
    {
        double tot = 0; int n = 0;
        for (int it = 0; it < 20; it++) {
            stamp<<<1, 64, 0, a>>>(ts + 2);
            for (int k = 0; k < 10; k++) empty_pred<<<8192, 256, 0, a>>>(plain + 8);
            stamp<<<1, 64, 0, a>>>(ts + 3);
            CK(hipStreamSynchronize(a));
            if (it >= 4) { tot += (ts[3] - ts[2]) * tick_us; n++; }
        }
        printf(", \"ten_predicated_off_launches_us\": %.2f", tot / n);
    }
    // the same two stamps with one wait-value (already satisfied) between them
    if (sig) {
        double tot = 0; int n = 0;
        CK(hipMemset(sig, 0, 8));
        uint32_t one = 1; CK(hipMemcpy(sig, &one, 4, hipMemcpyHostToDevice));
        for (int it = 0; it < 20; it++) {
            stamp<<<1, 64, 0, a>>>(ts + 2);
            hipError_t e = hipStreamWaitValue32(a, sig, 1u, hipStreamWaitValueEq, 0xFFFFFFFFu);
            if (e != hipSuccess) break;
            stamp<<<1, 64, 0, a>>>(ts + 3);
            CK(hipStreamSynchronize(a));
            if (it >= 4) { tot += (ts[3] - ts[2]) * tick_us; n++; }
        }
        if (n) printf(", \"satisfied_wait_value_between_two_kernels_us\": %.2f", tot / n);
        // ... and two back-to-back kernels with nothing between them
        tot = 0; n = 0;
        for (int it = 0; it < 20; it++) {
            stamp<<<1, 64, 0, a>>>(ts + 2);
            stamp<<<1, 64, 0, a>>>(ts + 3);
            CK(hipStreamSynchronize(a));
            if (it >= 4) { tot += (ts[3] - ts[2]) * tick_us; n++; }
        }
        printf(", \"two_kernels_back_to_back_us\": %.2f", tot / n);
    }
    printf("}\n");
    return 0;
}
