// alloc_probe9.hip — power-of-two sized (physically contiguous, highly aligned) allocations are ALWAYS of the slow kind, odd
// sizes sometimes fast (alloc_probe8).  Is "fast" simply "physically scattered"?  Buffers built with the virtual-memory API from
// chunks of a given size, mapped into one contiguous virtual range IN ORDER or in a SHUFFLED order, probed with the
// XCD-contiguous 8 KiB tile fill.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_fill(uint8_t* out, size_t bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const size_t t = b * 4 + wave;
    if ((t + 1) * 8192 > bytes) return;  // whole tiles only
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
    const int grid = (int)(bytes / 8192 / 4);  // bytes is a multiple of 8 * 4 * 8192
    float sum = 0;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, buf, bytes); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 4) sum += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return sum / 6;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t bytes = (size_t)65536 * 25200;  // = 8192 * 201600; 201600 / 32 = 6300 blocks of 4 tiles: a multiple of 8? 6300 = 8 * 787.5 -> no
    const size_t probe_bytes = bytes / (8 * 4 * 8192) * (8 * 4 * 8192);  // rounded DOWN to whole groups of 8 blocks
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    std::mt19937 rng(12345);
    for (size_t chunk : {(size_t)(2u << 20), (size_t)(16u << 20), (size_t)(256u << 10)}) {
        for (int shuffled = 0; shuffled < 2; shuffled++) {
            printf("chunks of %5zu KiB, %s:", chunk >> 10, shuffled ? "shuffled" : "in order");
            for (int rep = 0; rep < 4; rep++) {
                const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;  // total >= bytes >= probe_bytes
                void* va = nullptr;
                CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
                std::vector<hipMemGenericAllocationHandle_t> hs(n);
                for (size_t i = 0; i < n; i++) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
                std::vector<size_t> order(n);
                for (size_t i = 0; i < n; i++) order[i] = i;
                if (shuffled) std::shuffle(order.begin(), order.end(), rng);
                for (size_t i = 0; i < n; i++) CK(hipMemMap((uint8_t*)va + i * chunk, chunk, 0, hs[order[i]], 0));
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                CK(hipMemSetAccess(va, total, &acc, 1));
                printf(" %.3f", probe((uint8_t*)va, total, probe_bytes));
                fflush(stdout);
                CK(hipMemUnmap(va, total));
                for (size_t i = 0; i < n; i++) CK(hipMemRelease(hs[i]));
                CK(hipMemAddressFree(va, total));
            }
            printf("\n");
        }
    }
    return 0;
}
