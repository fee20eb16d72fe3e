// alloc_probe8.hip — does the SIZE of the allocation decide the kind of its first 1.65 GB?  For each size class, 12 allocations
// (all kept until the class is done), each probed with the XCD-contiguous 8 KiB tile fill over its first 65 536 * 25 200 bytes.
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
int main() {
    const size_t bytes = (size_t)65536 * 25200;
    const size_t M = (size_t)1 << 20;
    const size_t sizes[] = {bytes, (size_t)2 << 30, (size_t)4 << 30, bytes + 64 * M, bytes + 2 * M, bytes + 10 * M, bytes + 100 * M, bytes + 200 * M,
                            bytes + 300 * M, ((size_t)2 << 30) - 2 * M, ((size_t)2 << 30) + 2 * M, (size_t)3 << 30};  // every size >= bytes: the probe writes `bytes`
    for (size_t sz : sizes) {
        std::vector<uint8_t*> keep;
        printf("size %6.3f GiB:", (double)sz / (1 << 30));
        for (int i = 0; i < 12; i++) {
            uint8_t* buf;
            if (hipMalloc(&buf, sz) != hipSuccess) { printf(" (alloc failed)"); break; }
            keep.push_back(buf);
            printf(" %.3f", probe(buf, sz, bytes));
            fflush(stdout);
        }
        printf("\n");
        for (uint8_t* p : keep) (void)hipFree(p);
    }
    return 0;
}
