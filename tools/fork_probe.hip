// fork_probe.hip -- what does forking work onto a second stream cost the FIRST stream?  A loop of { busy kernel A; [fork variant]; busy kernel B } on one stream,
// wall time per iteration, for: nothing in between; hipEventRecord alone; record + a second stream that waits for it and runs a kernel; the same fork
// done with hipStreamWaitValue32 on the second stream and a word kernel A's last thread writes (no event on the first stream).
//   hipcc --offload-arch=gfx950 -O2 tools/fork_probe.hip -o gpurun_out/fork_probe && timeout 60 gpurun_out/fork_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void busy(long long ticks, uint32_t* word, uint32_t value, uint32_t* word_at_start = nullptr)
{
    if (word_at_start && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(word_at_start, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (word && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main()
{
    CK(hipSetDevice(0));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const long long t30 = (long long)(30.0 * clk_khz / 1000.0);      // 30 us
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev, join;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    uint32_t* word = nullptr;
    CK(hipMalloc((void**)&word, 64));
    CK(hipMemset(word, 0, 64));
    const int N = 300;
    uint32_t seq = 0;
    for (int variant = 0; variant < 7; variant++) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                seq++;
                busy<<<256, 256, 0, a>>>(t30, variant == 3 || variant == 4 ? word : nullptr, seq);
                if (variant == 1) CK(hipEventRecord(ev, a));
                if (variant == 2) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); busy<<<64, 256, 0, b>>>(t30 / 2, nullptr, 0u); }
                if (variant == 3) { CK(hipStreamWaitValue32(b, word, seq, hipStreamWaitValueGte, 0xFFFFFFFFu)); busy<<<64, 256, 0, b>>>(t30 / 2, nullptr, 0u); }
                if (variant == 4) { CK(hipStreamWaitValue32(b, word, seq, hipStreamWaitValueGte, 0xFFFFFFFFu)); busy<<<64, 256, 0, b>>>(t30 / 2, nullptr, 0u);
                                    CK(hipEventRecord(join, b)); }
                if (variant == 5 || variant == 6) { CK(hipStreamWaitValue32(b, word, seq, hipStreamWaitValueGte, 0xFFFFFFFFu)); busy<<<64, 256, 0, b>>>(t30 / 2, nullptr, 0u);
                                                    if (variant == 6) CK(hipEventRecord(join, b)); }
                busy<<<256, 256, 0, a>>>(t30, nullptr, seq, variant == 5 || variant == 6 ? word : nullptr);
                if (variant == 4) CK(hipStreamWaitEvent(a, join, 0));
                if (variant == 6) { CK(hipStreamWaitEvent(a, join, 0)); busy<<<256, 256, 0, a>>>(t30, nullptr, 0u); }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (us < best) best = us;
        }
        const char* names[7] = { "A; B", "A; record; B", "A; record, other stream waits + kernel; B", "A writes word, other stream hipStreamWaitValue32 + kernel; B",
                                 "as before + join (B; first stream waits for the other's event)", "B writes the word at its start, other stream hipStreamWaitValue32 + kernel",
                                 "as before + join + a third 30 us kernel C behind it (subtract 30)" };
        printf("%-70s %.2f us per iteration (two 30 us kernels)\n", names[variant], best);
    }
    return 0;
}
