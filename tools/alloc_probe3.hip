// alloc_probe3.hip — static stride vs dynamic (atomic counter) assignment of envs to persistent waves, bare store stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// DYN 2: the next env index is fetched ONE ENV AHEAD by an untracked returning atomic issued before the current env's
// stores and waited for with a counted vmcnt 16 stores later (vmcnt retires in order: a tracked atomic behind the
// stores would first drain them all — that is what DYN 1 measures)
#ifndef NSH
#define NSH 256
#endif
__global__ void __launch_bounds__(256) k2(uint8_t* out, int n_envs, int env_bytes, unsigned* counters) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int shard = wg % NSH;
    unsigned* counter = counters + shard * 16;  // one 64-byte line per shard
#define ENV_OF(t) ((int)(t) * NSH + shard)      /* shard s owns envs s, s + NSH, s + 2 NSH, ... (the static stride pattern) */
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    unsigned t = 0, one = 1;
    if (lane == 0) t = atomicAdd(counter, 1u);
    int e = ENV_OF(__builtin_amdgcn_readfirstlane(t));
    if (lane == 0) t = atomicAdd(counter, 1u);
    int e1 = ENV_OF(__builtin_amdgcn_readfirstlane(t));
    while (e < n_envs) {
        const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
        const int niter = (nchunks - k0 + 63) / 64;
        unsigned nxt = 0;
        for (int it = 0; it < niter; it++) {
            if (it == 0 && lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(nxt) : "v"(counter), "v"(one) : "memory");
            if (it == 16) asm volatile("s_waitcnt vmcnt(8)" : "+v"(nxt)::"memory");
            const int kk = k0 + lane + it * 64;
            if (kk >= 0 && kk < nchunks) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
        if (niter <= 16) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt)::"memory");
        e = e1;
        e1 = ENV_OF(__builtin_amdgcn_readfirstlane(nxt));
    }
}
template <int DYN>
__global__ void __launch_bounds__(256) k(uint8_t* out, int n_envs, int env_bytes, unsigned* counter) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    int e = DYN ? 0 : blockIdx.x * 4 + wave;
    if (DYN) { unsigned t = 0; if (lane == 0) t = atomicAdd(counter, 1u); e = __builtin_amdgcn_readfirstlane(t); }
    while (e < n_envs) {
        const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
        for (int kk = k0 + lane; kk < nchunks; kk += 64) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        if (DYN) { unsigned t = 0; if (lane == 0) t = atomicAdd(counter, 1u); e = __builtin_amdgcn_readfirstlane(t); }
        else e += gridDim.x * 4;
    }
}
template <int DYN> float run(uint8_t* buf, int E, int B, unsigned* ctr, int grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipMemsetAsync(ctr, 0, 65536, 0);
        (void)hipEventRecord(a);
        if (DYN == 2) hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, 0, buf, E, B, ctr);
        else hipLaunchKernelGGL(k<DYN>, dim3(grid), dim3(256), 0, 0, buf, E, B, ctr);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep >= 3 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200; const size_t bytes = (size_t)E * B;
    unsigned* ctr; (void)hipMalloc(&ctr, 65536);
    std::vector<uint8_t*> bufs;
    for (int i = 0; i < 6; i++) { uint8_t* p; if (hipMalloc(&p, bytes) != hipSuccess) return 1; bufs.push_back(p); }
    printf("%-16s %10s %10s %10s %10s %10s\n", "buffer", "static2048", "dyn2048", "static2032", "dyn2032", "naive-dyn");
    for (auto p : bufs)
        printf("%p %10.3f %10.3f %10.3f %10.3f %10.3f\n", (void*)p, run<0>(p, E, B, ctr, 2048), run<2>(p, E, B, ctr, 2048), run<0>(p, E, B, ctr, 2032), run<2>(p, E, B, ctr, 2032), run<1>(p, E, B, ctr, 2048));
    return 0;
}
