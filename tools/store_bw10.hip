// store_bw10.hip — the XCD-contiguous one-shot 8 KiB tile fill (tools/store_bw9.hip) with the eight XCDs' streams DE-PHASED:
// XCD x walks its eighth starting rot * x tiles into it (wrapping round), so that at any moment the eight streams sit at
// different offsets modulo the DRAM bank / channel interleave.  Run on several allocations: do the "slow" ones become fast?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_fill(uint8_t* out, size_t bytes, uint32_t rot_blocks, uint32_t mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = gridDim.x >> 3, x = blockIdx.x & 7;
    uint32_t q = blockIdx.x >> 3;
    if (mode == 1) { q += rot_blocks * x; q %= per; }          // rotation proportional to the XCD index
    else if (mode == 2) { q += rot_blocks * ((x * 5) & 7); q %= per; }  // a permuted rotation
    const size_t b = (size_t)x * per + q;
    const size_t t = b * 4 + wave;
    if (t * 8192 >= bytes) return;
    uint8_t* o = out + t * 8192;
    const size_t left = bytes - t * 8192;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const size_t off = (size_t)u * 1024 + (size_t)lane * 16;
        const uint32_t h = (uint32_t)(u * 64 + lane) * 7u;
        const u32x4 v = {h & 0x01010101u, (h >> 1) & 0x01010101u, (h >> 2) & 0x01010101u, (h >> 3) & 0x01010101u};
        if (off + 16 <= left) *(u32x4*)(o + off) = v;
    }
}
template <class F>
static float timeit(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float sum = 0;
    for (int rep = 0; rep < 24; rep++) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 8) sum += ms;
    }
    return sum / 16;
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)65536 * 25200;
    const int n = argc > 1 ? atoi(argv[1]) : 6;
    int grid = (int)((bytes / 8192 + 3) / 4) + 1; grid = (grid + 7) & ~7;
    for (int i = 0; i < n; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes + (64 << 20)) != hipSuccess) return 1;
        printf("alloc %d %p:", i, (void*)buf);
        const uint32_t rots[] = {0, 1, 2, 4, 8, 16, 32, 64, 128, 512, 2048, 3001};  // in blocks of 32 KiB
        for (uint32_t r : rots) {
            const float ms = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, bytes, r, r ? 1u : 0u); });
            printf("  r%u %.4f", r, ms);
        }
        const float ms2 = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, bytes, 16u, 2u); });
        printf("  perm16 %.4f", ms2);
        // the same buffer, shifted by a few MiB: is the kind a property of the base address?
        for (size_t shift : {(size_t)(1 << 20), (size_t)(3 << 20), (size_t)(17 << 20)}) {
            const float ms = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf + shift, bytes, 0u, 0u); });
            printf("  +%zuM %.4f", shift >> 20, ms);
        }
        printf("\n");
    }
    return 0;
}
