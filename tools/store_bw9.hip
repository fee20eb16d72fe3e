// store_bw9.hip — what else moves the rate of a ONE-SHOT tile fill (wave t writes TILE consecutive bytes and exits):
//   POLICY  the store's cache policy bits: plain / nt / sc0 / sc1 / sc0 sc1 / sc0 sc1 nt
//   MAP     which tile a block's waves take: 0 launch order (consecutive blocks go to consecutive XCDs, so neighbouring tiles
//           are written by different XCDs), 1 XCD-contiguous (XCD x writes the x-th eighth of the buffer front to back),
//           2 XCD-striped in 64 KiB runs (a run of consecutive tiles stays on one XCD)
//   LDS     padding per block, to cap the resident waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(uint8_t* p, u32x4 v) {
    if (POLICY == 0) *(u32x4*)p = v;
    else if (POLICY == 1) __builtin_nontemporal_store(v, (u32x4*)p);
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int TILE_KB, int POLICY, int MAP>
__global__ void __launch_bounds__(256) k_fill(uint8_t* out, size_t bytes, int pad_lds) {
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const size_t tile_bytes = (size_t)TILE_KB * 1024;
    const size_t n_tiles = (bytes + tile_bytes - 1) / tile_bytes;
    size_t b = blockIdx.x;
    const size_t nb = gridDim.x;
    if (MAP == 1) {  // XCD x = b % 8 takes blocks [x * nb / 8, (x + 1) * nb / 8)
        b = (b & 7) * (nb >> 3) + (b >> 3);
    } else if (MAP == 2) {  // runs of R consecutive blocks per XCD: block q of XCD x -> run (q / R) * 8 + x, position q % R
        const size_t R = (size_t)pad_lds / (4 * tile_bytes) ? (size_t)pad_lds / (4 * tile_bytes) : 1;  // pad_lds carries the run length in bytes
        const size_t x = b & 7, q = b >> 3;
        b = ((q / R) * 8 + x) * R + q % R;
    }
    const size_t t = b * 4 + wave;
    if (t >= n_tiles) return;
    if (MAP != 2 && pad_lds && lane == 0) pad[wave] = 1;
    uint8_t* o = out + t * tile_bytes;
    const size_t left = bytes - t * tile_bytes;
#pragma unroll
    for (int u = 0; u < TILE_KB; u++) {
        const size_t off = (size_t)u * 1024 + (size_t)lane * 16;
        const uint32_t h = (uint32_t)(u * 64 + lane) * 7u;
        const u32x4 v = {h & 0x01010101u, (h >> 1) & 0x01010101u, (h >> 2) & 0x01010101u, (h >> 3) & 0x01010101u};
        if (off + 16 <= left) store16<POLICY>(o + off, v);
    }
}

template <class F>
static float timeit(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float sum = 0;
    for (int rep = 0; rep < 30; rep++) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 10) sum += ms;
    }
    return sum / 20;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)65536 * 25200;
    const int nbuf = argc > 1 ? atoi(argv[1]) : 2;
    for (int i = 0; i < nbuf; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
        printf("alloc %d\n", i);
#define RUNS(KB, RUN, NAME)                                                                                                  \
        { int grid = (int)((bytes / (KB * 1024) + 3) / 4) + 1; const int per = 8 * (RUN / (4 * KB * 1024)); grid = (grid + per - 1) / per * per; \
          const float ms = timeit([&] { hipLaunchKernelGGL((k_fill<KB, 0, 2>), dim3(grid), dim3(256), 0, 0, buf, bytes, RUN); }); \
          printf("  %-44s %.4f ms  %.2f TB/s\n", NAME, ms, (double)bytes / ms / 1e9); }
#define FILL(KB, POL, MAP, LDS, NAME)                                                                                        \
        { int grid = (int)((bytes / (KB * 1024) + 3) / 4) + 1; grid = (grid + 7) & ~7;                                           \
          const float ms = timeit([&] { hipLaunchKernelGGL((k_fill<KB, POL, MAP>), dim3(grid), dim3(256), LDS, 0, buf, bytes, LDS ? 1 : 0); }); \
          printf("  %-44s %.4f ms  %.2f TB/s\n", NAME, ms, (double)bytes / ms / 1e9); }
        FILL(1, 0, 0, 0, "1 KiB plain")
        FILL(8, 0, 0, 0, "8 KiB plain")
        FILL(8, 1, 0, 0, "8 KiB nt")
        FILL(8, 2, 0, 0, "8 KiB sc0")
        FILL(8, 3, 0, 0, "8 KiB sc1")
        FILL(8, 4, 0, 0, "8 KiB sc0 sc1")
        FILL(8, 5, 0, 0, "8 KiB sc0 sc1 nt")
        FILL(4, 0, 0, 0, "4 KiB plain")
        FILL(4, 1, 0, 0, "4 KiB nt")
        FILL(4, 4, 0, 0, "4 KiB sc0 sc1")
        FILL(1, 0, 1, 0, "1 KiB plain, XCD-contiguous")
        FILL(4, 0, 1, 0, "4 KiB plain, XCD-contiguous")
        FILL(8, 0, 1, 0, "8 KiB plain, XCD-contiguous")
        RUNS(8, (64 << 10), "8 KiB, XCD runs of 64 KiB")
        FILL(8, 0, 0, 40960, "8 KiB plain, 4 blocks (16 waves) per CU")
        FILL(8, 0, 0, 81920, "8 KiB plain, 2 blocks (8 waves) per CU")
        FILL(8, 0, 1, 40960, "8 KiB XCD-contiguous, 16 waves per CU")
        FILL(4, 0, 0, 40960, "4 KiB plain, 16 waves per CU")
        FILL(8, 0, 0, 0, "8 KiB plain (again)")
RUNS(8, (256 << 10), "8 KiB, XCD runs of 256 KiB")
        RUNS(8, (1 << 20), "8 KiB, XCD runs of 1 MiB")
        RUNS(8, (2 << 20), "8 KiB, XCD runs of 2 MiB")
        RUNS(8, (4 << 20), "8 KiB, XCD runs of 4 MiB")
        RUNS(8, (16 << 20), "8 KiB, XCD runs of 16 MiB")
        RUNS(8, (64 << 20), "8 KiB, XCD runs of 64 MiB")
        RUNS(1, (2 << 20), "1 KiB, XCD runs of 2 MiB")
        RUNS(4, (2 << 20), "4 KiB, XCD runs of 2 MiB")
        (void)hipFree(buf);
    }
    return 0;
}
