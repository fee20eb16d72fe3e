// alloc_probe.hip — are some hipMalloc allocations slower to stream into than others, and for which access patterns?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0 wave per env (window = all resident waves x 25 KB), MODE 1 block per env, MODE 2 flat slabs per block (contiguous 806 KB per block)
template <int MODE>
__global__ void __launch_bounds__(256) k(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    if (MODE == 0) {
        for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
            const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + lane; kk < nchunks; kk += 64) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else if (MODE == 1) {
        for (int e = blockIdx.x; e < n_envs; e += gridDim.x) {
            const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + wave * 64 + lane; kk < nchunks; kk += 256) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else {
        const size_t n = (size_t)n_envs * env_bytes / 16, slab = (n + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * slab, hi = lo + slab < n ? lo + slab : n;
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) ((u32x4*)out)[i] = v;
    }
}
template <int MODE> float run(uint8_t* buf, int E, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 12; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, buf, E, B); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep >= 4 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200; const size_t bytes = (size_t)E * B;
    std::vector<uint8_t*> bufs;
    for (int round = 0; round < 2; round++) {
        for (int i = 0; i < 8; i++) { uint8_t* p; if (hipMalloc(&p, bytes) != hipSuccess) return 1; bufs.push_back(p); }
        for (auto p : bufs) printf("round %d buf %p : wave/env %.3f  block/env %.3f  slabs %.3f ms\n", round, (void*)p, run<0>(p, E, B), run<1>(p, E, B), run<2>(p, E, B));
        for (auto p : bufs) (void)hipFree(p);
        bufs.clear();
    }
    return 0;
}
