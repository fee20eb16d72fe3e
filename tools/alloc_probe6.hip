// alloc_probe6.hip — does the way a 1.65 GB buffer is allocated decide whether it is of the "fast" or the "slow" kind for the
// tile render's store pattern?  For each of N buffers from (a) hipMalloc, (b) the virtual-memory API (hipMemCreate in chunks of
// the recommended granularity, mapped into one reserved range) it times the one-shot 8 KiB tile fill in launch order and
// XCD-contiguous (tools/store_bw9.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MAP>
__global__ void __launch_bounds__(256) k_fill(uint8_t* out, size_t bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t b = blockIdx.x;
    if (MAP == 1) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
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
static void probe(const char* name, uint8_t* buf, size_t bytes) {
    int grid = (int)((bytes / 8192 + 3) / 4) + 1; grid = (grid + 7) & ~7;
    const float a = timeit([&] { hipLaunchKernelGGL(k_fill<0>, dim3(grid), dim3(256), 0, 0, buf, bytes); });
    const float b = timeit([&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, 0, buf, bytes); });
    printf("  %-28s %p  launch order %.4f ms (%.2f TB/s)   XCD-contiguous %.4f ms (%.2f TB/s)\n", name, (void*)buf, a, bytes / a / 1e9, b, bytes / b / 1e9);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t bytes = (size_t)65536 * 25200;
    const int n = argc > 1 ? atoi(argv[1]) : 6;
    std::vector<uint8_t*> keep;
    printf("hipMalloc\n");
    for (int i = 0; i < n; i++) {
        uint8_t* buf; CK(hipMalloc(&buf, bytes));
        keep.push_back(buf);
        probe("hipMalloc", buf, bytes);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("virtual-memory API: granularity min %zu recommended %zu\n", gmin, grec);
    for (size_t chunk : {(size_t)0, (size_t)(2u << 20), (size_t)(64u << 20), (size_t)(1u << 30)}) {
        for (int i = 0; i < (n + 1) / 2; i++) {
            const size_t g = grec ? grec : (2u << 20);
            size_t c = chunk ? (chunk + g - 1) / g * g : (bytes + g - 1) / g * g;  // 0: ONE physical allocation
            const size_t total = (bytes + c - 1) / c * c;
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
            for (size_t off = 0; off < total; off += c) {
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, c, &prop, 0));
                CK(hipMemMap((uint8_t*)va + off, c, 0, h, 0));
                CK(hipMemRelease(h));
            }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, total, &acc, 1));
            char name[64];
            snprintf(name, sizeof name, "hipMemCreate chunks of %zu MiB", c >> 20);
            probe(name, (uint8_t*)va, bytes);
        }
    }
    return 0;
}
