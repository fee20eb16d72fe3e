// store_bw8.hip — a render organised as ONE-SHOT TILES PER WAVE: wave t of the grid writes bytes [t * TILE, (t + 1) * TILE)
// of the flat observation buffer and exits.  Before it can store, it loads the state of the (at most two) envs its tile
// touches (368 B each), zeroes a TILE / 8 byte bitmap in LDS and sets a build's worth of hot bits (LDS atomics whose
// addresses depend on the loaded state), then streams TILE / 1 KiB store instructions: halfword -> 16 bytes -> store.
// Nothing is shared between the waves of a block (no __syncthreads).  Compared with:
//   fill      the same loop without any load / build (constant data)
//   persist   the same tiles in a persistent grid-stride loop (tile = it * n_waves + w: a compact moving window too)
//   stream    round 1's shape: wave per env, 25 200 contiguous bytes per wave, grid-stride over envs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LGKM_ONLY 0xC07F
constexpr int ENV_BYTES = 25200, STATE_WORDS = 92;
__device__ __forceinline__ uint32_t expand4(uint32_t h, int j) { return (((h >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u; }

// MODE 0 fill (no build), 1 one-shot tiles with build, 2 persistent tiles with build
template <int TILE_KB, int MODE, int WORK>
__global__ void __launch_bounds__(256) k_tiles(uint8_t* out, const uint32_t* state, size_t bytes, int n_envs) {
    __shared__ uint32_t lds_all[4 * (TILE_KB * 32 + 4)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* bits = lds_all + wave * (TILE_KB * 32 + 4);
    const size_t tile_bytes = (size_t)TILE_KB * 1024;
    const size_t n_tiles = (bytes + tile_bytes - 1) / tile_bytes;
    const size_t t_first = (size_t)blockIdx.x * 4 + wave, t_stride = (MODE == 2) ? (size_t)gridDim.x * 4 : n_tiles;
    for (size_t t = t_first; t < n_tiles; t += t_stride) {
        const size_t lo = t * tile_bytes;
        uint32_t salt = 0;
        if (MODE != 0) {
            const int e0 = (int)(lo / ENV_BYTES);
            const int e1 = min(e0 + 1, n_envs - 1);
            // the two envs' state: lanes 0..45 take two dwords of env e0, lanes 46..63 + wrap take env e1 (coalesced)
            const uint32_t s0 = state[(size_t)e0 * STATE_WORDS + lane], s1 = state[(size_t)e0 * STATE_WORDS + 64 + (lane < 28 ? lane : 0)];
            const uint32_t s2 = state[(size_t)e1 * STATE_WORDS + lane];
            for (int q = lane; q < TILE_KB * 8; q += 64) ((u32x4*)bits)[q] = u32x4{0u, 0u, 0u, 0u};
            __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
            __builtin_amdgcn_wave_barrier();
            uint32_t s = s0 ^ (s1 * 3u) ^ (s2 * 5u);
            for (int r = 0; r < WORK; r++) {
                const uint32_t bit = (s * 2654435761u + r * 40503u + lane) % (uint32_t)(TILE_KB * 1024);
                atomicOr(bits + (bit >> 5), 1u << (bit & 31));
                s = s * 1664525u + 1013904223u;
            }
            salt = s & 0x100u;
            __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
            __builtin_amdgcn_wave_barrier();
        }
        const uint16_t* hb = (const uint16_t*)bits;
        uint8_t* o = out + lo;
        const int nchunk = (int)(((bytes - lo < tile_bytes ? bytes - lo : tile_bytes)) >> 4);
#pragma unroll
        for (int u = 0; u < TILE_KB; u++) {
            const int k = u * 64 + lane;
            uint32_t h = (MODE == 0) ? (uint32_t)(k * 7) : hb[k];
            const u32x4 v = {expand4(h, 0), expand4(h, 1), expand4(h, 2), expand4(h, 3) | salt};
            if (k < nchunk) *(u32x4*)(o + ((size_t)k << 4)) = v;
        }
        if (MODE == 2) {
            __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// INTERLEAVED tiles: a block of NW waves owns a super-tile of NW * PER KiB; store j of wave w goes to KiB j * NW + w, so the
// block's waves together write a dense NW KiB per round.  MODE 0 fill, 1 with a per-wave build (bitmap of the wave's own PER
// pieces).  One-shot.
template <int NW, int PER, int MODE, int WORK>
__global__ void __launch_bounds__(NW * 64) k_inter(uint8_t* out, const uint32_t* state, size_t bytes, int n_envs) {
    __shared__ uint32_t lds_all[NW * (PER * 32 + 4)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* bits = lds_all + wave * (PER * 32 + 4);
    const size_t super = (size_t)NW * PER * 1024;
    const size_t lo = (size_t)blockIdx.x * super;
    if (lo >= bytes) return;
    uint32_t salt = 0;
    if (MODE != 0) {
        const int e0 = (int)(lo / ENV_BYTES);
        const int e1 = min(e0 + 1, n_envs - 1), e2 = min(e0 + 2, n_envs - 1);
        const uint32_t s0 = state[(size_t)e0 * STATE_WORDS + lane], s1 = state[(size_t)e1 * STATE_WORDS + lane];
        const uint32_t s2 = state[(size_t)e2 * STATE_WORDS + lane];
        for (int q = lane; q < PER * 8; q += 64) ((u32x4*)bits)[q] = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        uint32_t s = s0 ^ (s1 * 3u) ^ (s2 * 5u);
        for (int r = 0; r < WORK; r++) {
            const uint32_t bit = (s * 2654435761u + r * 40503u + lane) % (uint32_t)(PER * 1024);
            atomicOr(bits + (bit >> 5), 1u << (bit & 31));
            s = s * 1664525u + 1013904223u;
        }
        salt = s & 0x100u;
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
    }
    const uint16_t* hb = (const uint16_t*)bits;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const size_t off = lo + ((size_t)(j * NW + wave) << 10) + ((size_t)lane << 4);
        uint32_t h = (MODE == 0) ? (uint32_t)(lane * 7 + j) : hb[j * 64 + lane];
        const u32x4 v = {expand4(h, 0), expand4(h, 1), expand4(h, 2), expand4(h, 3) | salt};
        if (off + 16 <= bytes) *(u32x4*)(out + off) = v;
    }
}

// COOPERATIVE one-shot: a block of NW waves writes NW dense KiB, ONE store per lane; the block builds the bitmap of its NW KiB
// together (WORK atomics per lane), one __syncthreads, then stores.
template <int NW, int WORK>
__global__ void __launch_bounds__(NW * 64) k_coop(uint8_t* out, const uint32_t* state, size_t bytes, int n_envs) {
    __shared__ uint32_t bits[NW * 32 + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const size_t lo = (size_t)blockIdx.x * NW * 1024;
    if (lo >= bytes) return;
    const int e0 = (int)(lo / ENV_BYTES);
    const int e1 = min(e0 + 1, n_envs - 1);
    const uint32_t s0 = state[(size_t)e0 * STATE_WORDS + lane], s1 = state[(size_t)e1 * STATE_WORDS + lane];
    if (tid < NW * 8) ((u32x4*)bits)[tid] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    uint32_t s = s0 ^ (s1 * 3u) ^ (uint32_t)tid;
    for (int r = 0; r < WORK; r++) {
        const uint32_t bit = (s * 2654435761u + r * 40503u + tid) % (uint32_t)(NW * 1024);
        atomicOr(bits + (bit >> 5), 1u << (bit & 31));
        s = s * 1664525u + 1013904223u;
    }
    __syncthreads();
    const uint32_t h = ((const uint16_t*)bits)[tid];
    const u32x4 v = {expand4(h, 0), expand4(h, 1), expand4(h, 2), expand4(h, 3) | (s & 0x100u)};
    const size_t off = lo + ((size_t)tid << 4);
    if (off + 16 <= bytes) *(u32x4*)(out + off) = v;
}

// round 1's shape with a comparable build per env
template <int WORK>
__global__ void __launch_bounds__(256) k_stream(uint8_t* out, const uint32_t* state, int n_envs) {
    __shared__ uint32_t lds_all[4 * 792];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* bits = lds_all + wave * 792;
    for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
        const uint32_t s0 = state[(size_t)e * STATE_WORDS + lane], s1 = state[(size_t)e * STATE_WORDS + 64 + (lane < 28 ? lane : 0)];
        for (int q = lane; q < 198; q += 64) ((u32x4*)bits)[q] = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        uint32_t s = s0 ^ (s1 * 3u);
        for (int r = 0; r < WORK * 3; r++) {  // a whole env's bits: ~3 tiles' worth
            const uint32_t bit = (s * 2654435761u + r * 40503u + lane) % (uint32_t)ENV_BYTES;
            atomicOr(bits + (bit >> 5), 1u << (bit & 31));
            s = s * 1664525u + 1013904223u;
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        const size_t base = (size_t)e * ENV_BYTES;
        uint8_t* o = out + base;
        const int k0 = -(int)((base >> 4) & 63);
        const uint16_t* hb = (const uint16_t*)bits;
        for (int it = 0; k0 + it * 64 < ENV_BYTES / 16; it++) {
            const int k = k0 + it * 64 + lane;
            const uint32_t h = (k >= 0 && k < ENV_BYTES / 16) ? hb[k] : 0u;
            const u32x4 v = {expand4(h, 0), expand4(h, 1), expand4(h, 2), expand4(h, 3)};
            if (k >= 0 && k < ENV_BYTES / 16) *(u32x4*)(o + ((size_t)k << 4)) = v;
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
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
    const int E = 65536;
    const size_t bytes = (size_t)E * ENV_BYTES;
    const int nbuf = argc > 1 ? atoi(argv[1]) : 3;
    uint32_t* st;
    if (hipMalloc(&st, (size_t)(E + 2) * STATE_WORDS * 4) != hipSuccess) return 1;
    (void)hipMemset(st, 7, (size_t)(E + 2) * STATE_WORDS * 4);
    for (int i = 0; i < nbuf; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        printf("alloc %d\n", i);
#define TILES(KB, MODE, WORK, GRID, NAME)                                                                                   \
        { const float ms = timeit([&] { hipLaunchKernelGGL((k_tiles<KB, MODE, WORK>), dim3(GRID), dim3(256), 0, 0, buf, st, bytes, E); }); \
          printf("  %-34s %.4f ms  %.2f TB/s\n", NAME, ms, (double)bytes / ms / 1e9); }
        const int t4 = (int)((bytes / 4096 + 3) / 4) + 1, t8 = (int)((bytes / 8192 + 3) / 4) + 1, t16 = (int)((bytes / 16384 + 3) / 4) + 1;
        { const float ms = timeit([&] { (void)hipMemsetAsync(buf, 1, bytes, 0); });
          printf("  %-34s %.4f ms  %.2f TB/s\n", "hipMemsetAsync", ms, (double)bytes / ms / 1e9); }
        const int t1 = (int)((bytes / 1024 + 3) / 4) + 1, t2 = (int)((bytes / 2048 + 3) / 4) + 1;
        TILES(1, 0, 0, t1, "fill      1 KiB tiles, one-shot")
        TILES(2, 0, 0, t2, "fill      2 KiB tiles, one-shot")
        TILES(4, 0, 0, t4, "fill      4 KiB tiles, one-shot")
        TILES(8, 0, 0, t8, "fill      8 KiB tiles, one-shot")
#define INTER(NW, PER, MODE, WORK, NAME)                                                                                     \
        { const int grid = (int)((bytes + (size_t)NW * PER * 1024 - 1) / ((size_t)NW * PER * 1024));                             \
          const float ms = timeit([&] { hipLaunchKernelGGL((k_inter<NW, PER, MODE, WORK>), dim3(grid), dim3(NW * 64), 0, 0, buf, st, bytes, E); }); \
          printf("  %-34s %.4f ms  %.2f TB/s\n", NAME, ms, (double)bytes / ms / 1e9); }
        INTER(4, 4, 0, 0, "inter fill 4 waves x 4")
        INTER(4, 8, 0, 0, "inter fill 4 waves x 8")
        INTER(8, 4, 0, 0, "inter fill 8 waves x 4")
        INTER(16, 4, 0, 0, "inter fill 16 waves x 4")
        INTER(16, 8, 0, 0, "inter fill 16 waves x 8")
        INTER(4, 4, 1, 4, "inter build 4 waves x 4")
        INTER(4, 8, 1, 8, "inter build 4 waves x 8")
        INTER(16, 4, 1, 4, "inter build 16 waves x 4")
        INTER(16, 8, 1, 8, "inter build 16 waves x 8")
#define COOP(NW, WORK, NAME)                                                                                                 \
        { const int grid = (int)((bytes + (size_t)NW * 1024 - 1) / ((size_t)NW * 1024));                                         \
          const float ms = timeit([&] { hipLaunchKernelGGL((k_coop<NW, WORK>), dim3(grid), dim3(NW * 64), 0, 0, buf, st, bytes, E); }); \
          printf("  %-34s %.4f ms  %.2f TB/s\n", NAME, ms, (double)bytes / ms / 1e9); }
        COOP(4, 1, "coop 4 waves, 1 atomic/lane")
        COOP(4, 2, "coop 4 waves, 2 atomics/lane")
        COOP(8, 1, "coop 8 waves, 1 atomic/lane")
        COOP(16, 1, "coop 16 waves, 1 atomic/lane")
        TILES(1, 1, 1, t1, "one-shot  1 KiB tiles, build 1")
        TILES(2, 1, 2, t2, "one-shot  2 KiB tiles, build 2")
        TILES(4, 1, 4, t4, "one-shot  4 KiB tiles, build 4")
        TILES(8, 1, 4, t8, "one-shot  8 KiB tiles, build 4")
        TILES(8, 1, 8, t8, "one-shot  8 KiB tiles, build 8")
        TILES(16, 1, 8, t16, "one-shot 16 KiB tiles, build 8")
        TILES(8, 2, 4, 2048, "persist   8 KiB tiles, build 4, 2048 blocks")
        TILES(8, 2, 4, 1024, "persist   8 KiB tiles, build 4, 1024 blocks")
        { const float ms = timeit([&] { hipLaunchKernelGGL((k_stream<4>), dim3(2048), dim3(256), 0, 0, buf, st, E); });
          printf("  %-34s %.4f ms  %.2f TB/s\n", "stream    wave per env, build 12", ms, (double)bytes / ms / 1e9); }
        { const float ms = timeit([&] { hipLaunchKernelGGL((k_stream<4>), dim3(1536), dim3(256), 0, 0, buf, st, E); });
          printf("  %-34s %.4f ms  %.2f TB/s\n", "stream    wave per env, 24 w/CU", ms, (double)bytes / ms / 1e9); }
    }
    return 0;
}
