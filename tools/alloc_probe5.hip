// alloc_probe5.hip — is the fast / slow kind a property of a whole allocation or of regions inside a big one?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
        const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((((size_t)out + base) >> 4) & 63);
        for (int kk = k0 + lane; kk < nchunks; kk += 64) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
    }
}
float run(uint8_t* buf, int E, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 8; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, buf, E, B); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep >= 2 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200; const size_t bytes = (size_t)E * B;
    for (int trial = 0; trial < 2; trial++) {
        uint8_t* big; const size_t total = (size_t)24 << 30;
        if (hipMalloc(&big, total) != hipSuccess) return 1;
        printf("trial %d: one 24 GiB allocation at %p, 1.65 GB windows:\n", trial, (void*)big);
        for (size_t off = 0; off + bytes <= total; off += (size_t)2 << 30) printf("  +%2zu GiB %.3f ms\n", off >> 30, run(big + off, E, B));
        uint8_t* small[4];
        for (int i = 0; i < 4; i++) { (void)hipMalloc(&small[i], bytes); printf("  separate 1.65 GB allocation %p %.3f ms\n", (void*)small[i], run(small[i], E, B)); }
        for (int i = 0; i < 4; i++) (void)hipFree(small[i]);
        (void)hipFree(big);
    }
    return 0;
}
