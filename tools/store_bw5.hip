// store_bw5.hip — the bare wave-per-env aligned store stream, 300 launches (warm clocks), for PMC comparison with k_observe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_store_stream(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
        const size_t base = (size_t)e * env_bytes;
        const int nchunks = env_bytes >> 4;
        const int k0 = -(int)((base >> 4) & 63);
        const u32x4 v = {(uint32_t)e, (uint32_t)lane, 1u, 0x01000100u};
        for (int k = k0 + lane; k < nchunks; k += 64)
            if (k >= 0) *(u32x4*)(out + base + ((size_t)k << 4)) = v;
    }
}
int main() {
    const int E = 65536, B = 25200;
    uint8_t* buf;
    if (hipMalloc(&buf, (size_t)E * B) != hipSuccess) return 1;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 300; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k_store_stream, dim3(2048), dim3(256), 0, 0, buf, E, B);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        if (rep >= 200) sum += ms;
    }
    printf("bare store stream: best %.3f ms, mean of last 100 %.3f ms (%.2f TB/s)\n", best, sum / 100, (double)E * B / (sum / 100) / 1e9);
    return 0;
}
