// alloc_probe7.hip — along a long sequence of 1.65 GB allocations (all kept), which ones are of the fast kind for the
// XCD-contiguous 8 KiB tile fill?  (Is a box "all slow", or only the first ten?)  Then: free everything, allocate one 64 GB
// spacer, and look at the next ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_fill(uint8_t* out, size_t bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const size_t t = b * 4 + wave;
    if (t * 8192 >= bytes) return;
    uint8_t* o = out + t * 8192;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const uint32_t h = (uint32_t)(u * 64 + lane) * 7u;
        const u32x4 v = {h & 0x01010101u, (h >> 1) & 0x01010101u, (h >> 2) & 0x01010101u, (h >> 3) & 0x01010101u};
        *(u32x4*)(o + (size_t)u * 1024 + (size_t)lane * 16) = v;
    }
}
// writes min(bytes, alloc_bytes) rounded down to whole groups of 8 blocks of four 8 KiB tiles: never past the allocation
static float probe(uint8_t* buf, size_t alloc_bytes, size_t bytes) {
    if (bytes > alloc_bytes) bytes = alloc_bytes;
    bytes = bytes / (8 * 4 * 8192) * (8 * 4 * 8192);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = (int)(bytes / 8192 / 4);
    float sum = 0;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, bytes); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 4) sum += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return sum / 6;
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)65536 * 25200;  // a multiple of 32 KiB * 8
    const int n = argc > 1 ? atoi(argv[1]) : 60;
    std::vector<uint8_t*> keep;
    printf("sequence of %d allocations, all kept (ms per fill):\n", n);
    for (int i = 0; i < n; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("\nhipMalloc failed at %d\n", i); break; }
        keep.push_back(buf);
        printf(" %.3f", probe(buf, bytes, bytes));
        if (i % 20 == 19) printf("\n");
        fflush(stdout);
    }
    for (uint8_t* p : keep) (void)hipFree(p);
    keep.clear();
    printf("\nafter freeing all and a 64 GB spacer:\n");
    uint8_t* spacer = nullptr;
    if (hipMalloc(&spacer, (size_t)64 << 30) != hipSuccess) printf(" (spacer failed)\n");
    for (int i = 0; i < 12; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) break;
        keep.push_back(buf);
        printf(" %.3f", probe(buf, bytes, bytes));
    }
    printf("\n");
    return 0;
}
