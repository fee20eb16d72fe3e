// store_bw11.hip — would a PERSISTENT, software-pipelined tile render beat the one-shot one?  Both walk the buffer XCD-contiguously
// (tools/store_bw9.hip) in 8 KiB tiles and do the same token build per tile (3 state dwords per lane loaded, TILE/8 bytes of
// bitmap zeroed, WORK LDS atomics per lane whose addresses depend on the loaded state, then 8 store instructions):
//   oneshot   wave per tile: loads -> wait -> build -> stores -> end (s_endpgm waits for the stores: the slot is held meanwhile)
//   persist   8 192 resident waves; the loads of tile k+1 are issued BEFORE tile k is built and stored, so a wave never waits
//             for its stores and its load latency hides behind the previous tile's build
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LGKM_ONLY 0xC07F
constexpr int ENV_BYTES = 25200, STATE_WORDS = 92, TILE = 8192;
__device__ __forceinline__ uint32_t expand4(uint32_t h, int j) { return (((h >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u; }

struct St { uint32_t s0, s1, s2; };
__device__ __forceinline__ St load_state(const uint32_t* state, size_t tile, int n_envs, int lane) {
    const int e0 = (int)(tile * TILE / ENV_BYTES);
    const int e1 = min(e0 + 1, n_envs - 1);
    St r;
    r.s0 = state[(size_t)e0 * STATE_WORDS + lane];
    r.s1 = state[(size_t)e0 * STATE_WORDS + 64 + (lane < 28 ? lane : 0)];
    r.s2 = state[(size_t)e1 * STATE_WORDS + lane];
    return r;
}
template <int WORK, bool FULL = false>
__device__ __forceinline__ void build_and_store(uint32_t* bits, St st, uint8_t* out, size_t tile, size_t bytes, int lane) {
    for (int q = lane; q < TILE / 8 / 16; q += 64) ((u32x4*)bits)[q] = u32x4{0u, 0u, 0u, 0u};
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
    __builtin_amdgcn_wave_barrier();
    uint32_t s = st.s0 ^ (st.s1 * 3u) ^ (st.s2 * 5u);
    for (int r = 0; r < WORK; r++) {
        const uint32_t bit = (s * 2654435761u + r * 40503u + lane) % (uint32_t)TILE;
        atomicOr(bits + (bit >> 5), 1u << (bit & 31));
        s = s * 1664525u + 1013904223u;
    }
    const uint32_t salt = s & 0x100u;
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
    __builtin_amdgcn_wave_barrier();
    const uint16_t* hb = (const uint16_t*)bits;
    const size_t lo = tile * TILE;
    const int nchunk = (int)((bytes - lo < (size_t)TILE ? bytes - lo : (size_t)TILE) >> 4);
    uint32_t h[8];
#pragma unroll
    for (int u = 0; u < 8; u++) h[u] = hb[u * 64 + lane];
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
    __builtin_amdgcn_wave_barrier();  // the bitmap is in registers: the next tile may zero it
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int k = u * 64 + lane;
        const u32x4 v = {expand4(h[u], 0), expand4(h[u], 1), expand4(h[u], 2), expand4(h[u], 3) | salt};
        if (FULL || k < nchunk) *(u32x4*)(out + lo + ((size_t)k << 4)) = v;  // FULL: no branch, so that a counted wait can skip the stores
    }
}

template <int WORK>
__global__ void __launch_bounds__(256) k_oneshot(uint8_t* out, const uint32_t* state, size_t bytes, int n_envs) {
    __shared__ uint32_t lds_all[4 * (TILE / 32 + 4)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t n_tiles = (bytes + TILE - 1) / TILE;
    const size_t b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const size_t t = b * 4 + wave;
    if (t >= n_tiles) return;
    const St st = load_state(state, t, n_envs, lane);
    build_and_store<WORK>(lds_all + wave * (TILE / 32 + 4), st, out, t, bytes, lane);
}

template <int WORK>
__global__ void __launch_bounds__(256) k_persist(uint8_t* out, const uint32_t* state, size_t bytes, int n_envs) {
    __shared__ uint32_t lds_all[4 * (TILE / 32 + 4)];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t n_tiles = (bytes + TILE - 1) / TILE;
    const size_t per_xcd = (n_tiles + 7) / 8;                       // tiles of one XCD's eighth
    const size_t x = blockIdx.x & 7, q = (size_t)(blockIdx.x >> 3) * 4 + wave, stride = (size_t)(gridDim.x >> 3) * 4;
    const size_t lo = x * per_xcd, hi = lo + per_xcd < n_tiles ? lo + per_xcd : n_tiles;
    size_t t = lo + q;
    if (t >= hi) return;
    St cur = load_state(state, t, n_envs, lane);
    // the first tile's loads are waited for HERE, with a wait the compiler sees: a pending load on the loop's entry edge would make
    // it put a vmcnt(0) into the loop header, i.e. drain the previous tile's stores in every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (;;) {
        const size_t tn = t + stride;
        const bool more = tn < hi;
        const St nxt = load_state(state, more ? tn : t, n_envs, lane);  // issued before this tile's build and stores
        build_and_store<WORK, true>(lds_all + wave * (TILE / 32 + 4), cur, out, t, bytes, lane);  // (bytes is a multiple of the tile here)
        if (!more) break;
        cur = nxt;
        t = tn;
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
    const int E = 65536;
    const size_t bytes = (size_t)E * ENV_BYTES;
    const int n = argc > 1 ? atoi(argv[1]) : 6;
    uint32_t* st;
    if (hipMalloc(&st, (size_t)(E + 2) * STATE_WORDS * 4) != hipSuccess) return 1;
    (void)hipMemset(st, 7, (size_t)(E + 2) * STATE_WORDS * 4);
    int grid1 = (int)((bytes / TILE + 3) / 4) + 1; grid1 = (grid1 + 7) & ~7;
    for (int i = 0; i < n; i++) {
        uint8_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        const float f = timeit([&] { (void)hipMemsetAsync(buf, 1, bytes, 0); });
        const float a4 = timeit([&] { hipLaunchKernelGGL(k_oneshot<4>, dim3(grid1), dim3(256), 0, 0, buf, st, bytes, E); });
        const float a12 = timeit([&] { hipLaunchKernelGGL(k_oneshot<12>, dim3(grid1), dim3(256), 0, 0, buf, st, bytes, E); });
        const float p4 = timeit([&] { hipLaunchKernelGGL(k_persist<4>, dim3(2048), dim3(256), 0, 0, buf, st, bytes, E); });
        const float p12 = timeit([&] { hipLaunchKernelGGL(k_persist<12>, dim3(2048), dim3(256), 0, 0, buf, st, bytes, E); });
        const float p12h = timeit([&] { hipLaunchKernelGGL(k_persist<12>, dim3(1024), dim3(256), 0, 0, buf, st, bytes, E); });
        printf("alloc %d: memset %.4f | one-shot build 4 %.4f, build 12 %.4f | persistent pipelined build 4 %.4f, build 12 %.4f, build 12 at 16 waves/CU %.4f ms\n",
               i, f, a4, a12, p4, p12, p12h);
    }
    return 0;
}
