// store_bw7.hip — what feeds the observation store stream best, and how many waves per CU it needs.
// All variants write E = 65 536 blocks of 25 200 bytes (the arena observation) wave-per-env, 1 KiB per store instruction:
//   bare     constant data (the ceiling)
//   bitmap   k_observe's shape: halfword k of a 3 152-byte LDS bitmap -> 16 bytes by 3 VALU ops per 4 bytes
//   table    the same bitmap, expanded through a 256-entry byte -> 8-byte LDS table (two ds_read_b64 per chunk)
//   image    a persistent 25 200-byte BYTE image per wave in LDS (only ~250 bytes change per env): ds_read_b128 -> store
// each at several occupancies (waves per CU).  Per env every variant also does a token "build" (a few LDS writes +
// a wave barrier) so that the stream is interrupted the way the real render is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LGKM_ONLY 0xC07F
constexpr int ENV_BYTES = 25200, NCHUNK = ENV_BYTES / 16, BITMAP_BYTES = 3168;

__device__ __forceinline__ uint32_t expand4(uint32_t h, int j) { return (((h >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u; }

template <int MODE>  // 0 bare, 1 bitmap, 2 table, 3 image
__global__ void k_stream(uint8_t* out, int n_envs, int lds_per_wave) {
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    uint8_t* wl = lds + wave * lds_per_wave;
    uint8_t* table = lds + wpb * lds_per_wave;  // MODE 2: shared by the block
    if (MODE == 1 || MODE == 2) for (int q = lane; q < BITMAP_BYTES / 4; q += 64) ((uint32_t*)wl)[q] = 0;
    if (MODE == 3) for (int q = lane; q < ENV_BYTES / 4; q += 64) ((uint32_t*)wl)[q] = 0;
    if (MODE == 2) {
        for (int v = threadIdx.x; v < 256; v += blockDim.x) {
            u32x2 t = {expand4(v, 0), expand4(v, 1)};
            ((u32x2*)table)[v] = t;
        }
        __syncthreads();
    }
    for (int e = blockIdx.x * wpb + wave; e < n_envs; e += gridDim.x * wpb) {
        // token build: ~4 LDS byte/bit updates per lane, then a wave barrier
        if (MODE == 1 || MODE == 2) {
            for (int r = 0; r < 4; r++) {
                const uint32_t bit = ((uint32_t)(e * 97 + lane * 389 + r * 6311)) % (uint32_t)ENV_BYTES;
                atomicXor((uint32_t*)wl + (bit >> 5), 1u << (bit & 31));
            }
        } else if (MODE == 3) {
            for (int r = 0; r < 4; r++) {
                const uint32_t byte = ((uint32_t)(e * 97 + lane * 389 + r * 6311)) % (uint32_t)ENV_BYTES;
                wl[byte] ^= 1;
            }
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        const size_t base = (size_t)e * ENV_BYTES;
        uint8_t* o = out + base;
        const int k0 = -(int)((base >> 4) & 63);
        const int niter = (NCHUNK - k0 + 63) / 64;
        const uint16_t* hb = (const uint16_t*)wl;
        for (int it0 = 0; it0 < niter; it0 += 8) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    const u32x4 v = {(uint32_t)e, (uint32_t)lane, 1u, 0x01000100u};
                    if (k >= 0 && k < NCHUNK) *(u32x4*)(o + ((size_t)k << 4)) = v;
                }
            } else if (MODE == 1) {
                uint32_t h[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    h[u] = (k >= 0 && k < NCHUNK) ? hb[k] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    const u32x4 v = {expand4(h[u], 0), expand4(h[u], 1), expand4(h[u], 2), expand4(h[u], 3)};
                    if (k >= 0 && k < NCHUNK) *(u32x4*)(o + ((size_t)k << 4)) = v;
                }
            } else if (MODE == 2) {
                uint32_t h[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    h[u] = (k >= 0 && k < NCHUNK) ? hb[k] : 0u;
                }
                u32x2 lo[8], hi[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    lo[u] = ((const u32x2*)table)[h[u] & 255u];
                    hi[u] = ((const u32x2*)table)[h[u] >> 8];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    const u32x4 v = {lo[u][0], lo[u][1], hi[u][0], hi[u][1]};
                    if (k >= 0 && k < NCHUNK) *(u32x4*)(o + ((size_t)k << 4)) = v;
                }
            } else {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    const int kc = k < 0 ? 0 : (k >= NCHUNK ? NCHUNK - 1 : k);
                    v[u] = ((const u32x4*)wl)[kc];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + lane + (it0 + u) * 64;
                    if (k >= 0 && k < NCHUNK) *(u32x4*)(o + ((size_t)k << 4)) = v[u];
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
    }
}

int main(int argc, char** argv) {
    const int E = 65536;
    const int nbuf = argc > 1 ? atoi(argv[1]) : 3;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    uint8_t* bufs[8];
    for (int i = 0; i < nbuf; i++) if (hipMalloc(&bufs[i], (size_t)E * ENV_BYTES) != hipSuccess) return 1;
    struct Cfg { int mode, wpb, blocks_per_cu; const char* name; };
    const Cfg cfgs[] = {
        {0, 4, 8, "bare   32w/CU"}, {0, 4, 4, "bare   16w/CU"}, {0, 4, 2, "bare    8w/CU"}, {0, 2, 3, "bare    6w/CU"}, {0, 2, 2, "bare    4w/CU"},
        {1, 4, 8, "bitmap 32w/CU"}, {1, 4, 4, "bitmap 16w/CU"}, {1, 4, 2, "bitmap  8w/CU"},
        {2, 4, 8, "table  32w/CU"}, {2, 4, 4, "table  16w/CU"}, {2, 4, 2, "table   8w/CU"},
        {3, 2, 3, "image   6w/CU"}, {3, 2, 2, "image   4w/CU"}, {3, 1, 6, "image 6x1w/CU"}, {3, 1, 3, "image 3x1w/CU"},
    };
    for (int i = 0; i < nbuf; i++) {
        printf("alloc %d\n", i);
        for (const Cfg& c : cfgs) {
            const int per_wave = c.mode == 3 ? ENV_BYTES : (c.mode == 0 ? 16 : BITMAP_BYTES);
            const size_t sh = (size_t)c.wpb * per_wave + (c.mode == 2 ? 2048 : 0);
            const dim3 grid(256 * c.blocks_per_cu), block(64 * c.wpb);
            if (sh > 64 * 1024) {
                hipError_t er = hipSuccess;
                if (c.mode == 3) er = hipFuncSetAttribute((const void*)k_stream<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
                if (er != hipSuccess) { printf("  %s: cannot raise LDS limit\n", c.name); continue; }
            }
            float sum = 0;
            for (int rep = 0; rep < 30; rep++) {
                (void)hipEventRecord(a);
                if (c.mode == 0) hipLaunchKernelGGL(k_stream<0>, grid, block, sh, 0, bufs[i], E, per_wave);
                if (c.mode == 1) hipLaunchKernelGGL(k_stream<1>, grid, block, sh, 0, bufs[i], E, per_wave);
                if (c.mode == 2) hipLaunchKernelGGL(k_stream<2>, grid, block, sh, 0, bufs[i], E, per_wave);
                if (c.mode == 3) hipLaunchKernelGGL(k_stream<3>, grid, block, sh, 0, bufs[i], E, per_wave);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep >= 10) sum += ms;
            }
            if (hipGetLastError() != hipSuccess) { printf("  %s: launch error\n", c.name); continue; }
            const float ms = sum / 20;
            printf("  %s  %.4f ms  %.2f TB/s\n", c.name, ms, (double)E * ENV_BYTES / ms / 1e9);
        }
    }
    return 0;
}
