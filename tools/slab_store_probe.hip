// slab_store_probe.hip -- what do the bucket scatter's slab stores cost as 16-byte and as 8-byte elements?  (round 6 development probe)
// The pattern of depth_bucket_scatter_kernel in isolation: n elements, each to a random one of nb buckets, workgroup b writes the sub-slabs of
// XCD b mod 8 only, a returning atomic per element hands out the slot, the element is one 16-byte (uint4) or one 8-byte (uint2) store.
//   hipcc --offload-arch=gfx950 -O2 tools/slab_store_probe.hip -o gpurun_out/slab_store_probe && timeout 60 gpurun_out/slab_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITEMS = 16, CAPX = 128, XCD = 8;

template <int BYTES, bool STORE, bool ATOMIC>
__global__ void __launch_bounds__(256) scatter(const uint32_t* __restrict__ bucket_of, uint32_t n, uint32_t nb, uint32_t* __restrict__ gcount, void* __restrict__ slab)
{
    const uint32_t xcd = blockIdx.x & (XCD - 1);
    const uint32_t base = blockIdx.x * (256 * ITEMS);
    uint32_t b[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) { const uint32_t i = base + r * 256 + threadIdx.x; b[r] = i < n ? bucket_of[i] : 0xFFFFFFFFu; }
    uint32_t pos[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) pos[r] = (ATOMIC && b[r] != 0xFFFFFFFFu) ? atomicAdd(&gcount[(size_t)xcd * nb + b[r]], 1u) : (blockIdx.x >> 3) & (CAPX - 1);
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (b[r] == 0xFFFFFFFFu || pos[r] >= CAPX) continue;
        const size_t slot = ((size_t)b[r] * XCD + xcd) * CAPX + pos[r];
        const uint32_t i = base + r * 256 + threadIdx.x;
        if (STORE) {
            if (BYTES == 16) reinterpret_cast<uint4*>(slab)[slot] = make_uint4(i, b[r], pos[r], 7u);
            else reinterpret_cast<uint2*>(slab)[slot] = make_uint2(i, b[r]);
        }
    }
}

int main()
{
    const uint32_t n = 3000000, nb = 8192;
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % nb); }
    uint32_t *bucket_of, *gcount; void* slab;
    CK(hipMalloc((void**)&bucket_of, n * 4)); CK(hipMalloc((void**)&gcount, (size_t)XCD * nb * 4)); CK(hipMalloc(&slab, (size_t)nb * XCD * CAPX * 16));
    CK(hipMemcpy(bucket_of, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (n + 256 * ITEMS - 1) / (256 * ITEMS);
    auto run = [&](const char* name, void (*k)(const uint32_t*, uint32_t, uint32_t, uint32_t*, void*)) -> int {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 12; it++) {
            CK(hipMemsetAsync(gcount, 0, (size_t)XCD * nb * 4, 0));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, bucket_of, n, nb, gcount, slab);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-44s avg %7.1f us   best %7.1f us\n", name, sum / 10 * 1e3f, best * 1e3f);
        return 0;
    };
    if (run("atomics + 16-byte stores (as built)", scatter<16, true, true>)) return 1;
    if (run("atomics +  8-byte stores", scatter<8, true, true>)) return 1;
    if (run("atomics only", scatter<16, false, true>)) return 1;
    if (run("16-byte stores only (slot from the block id)", scatter<16, true, false>)) return 1;
    if (run(" 8-byte stores only", scatter<8, true, false>)) return 1;
    return 0;
}
