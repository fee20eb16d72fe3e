// store_bw12.hip — on a "slow" box every candidate buffer takes the one-shot 8 KiB tile fill (XCD-contiguous) 10 % slower than
// elsewhere, while a memset does not care.  Does a tile whose stores INTERLEAVE with its block-mates' (wave w's store u covers
// KiB u * NW + w of an NW * PER KiB block tile: the block writes NW dense KiB per round) behave like the memset?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NW, int PER, bool INTER>
__global__ void __launch_bounds__(NW * 64) k_fill(uint8_t* out, size_t bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t super = (size_t)NW * PER * 1024;
    const size_t b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous
    const size_t lo = b * super;
    if (lo >= bytes) return;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const size_t kib = INTER ? (size_t)(u * NW + wave) : (size_t)(wave * PER + u);
        const size_t off = lo + (kib << 10) + ((size_t)lane << 4);
        const uint32_t h = (uint32_t)(u * 64 + lane) * 7u;
        const u32x4 v = {h & 0x01010101u, (h >> 1) & 0x01010101u, (h >> 2) & 0x01010101u, (h >> 3) & 0x01010101u};
        if (off + 16 <= bytes) *(u32x4*)(out + off) = v;
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
    const int n = argc > 1 ? atoi(argv[1]) : 4;
    for (int i = 0; i < n; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        printf("alloc %d: memset %.4f", i, timeit([&] { (void)hipMemsetAsync(buf, 1, bytes, 0); }));
#define RUN(NW, PER, INTER, NAME)                                                                                          \
        { int grid = (int)((bytes + (size_t)NW * PER * 1024 - 1) / ((size_t)NW * PER * 1024)); grid = (grid + 7) & ~7;         \
          printf(" | %s %.4f", NAME, timeit([&] { hipLaunchKernelGGL((k_fill<NW, PER, INTER>), dim3(grid), dim3(NW * 64), 0, 0, buf, bytes); })); }
        RUN(4, 8, false, "4w x 8 plain")
        RUN(4, 8, true, "4w x 8 inter")
        RUN(4, 4, true, "4w x 4 inter")
        RUN(8, 8, true, "8w x 8 inter")
        RUN(16, 8, true, "16w x 8 inter")
        RUN(4, 1, false, "1 KiB")
        RUN(4, 2, true, "4w x 2 inter")
        printf("\n");
    }
    return 0;
}
