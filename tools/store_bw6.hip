// store_bw6.hip — bare store streams over the observation buffer in four shapes, in one process on several allocations:
//   A  wave per env (k_observe's shape): each of 8 192 waves streams 25 200 contiguous bytes, 1 KiB per instruction
//   B  block per env: the 4 waves of a block share one env, 4 KiB per round (the 4 waves' instructions are adjacent)
//   C  block per env, 8 blocks per CU but 2 waves per block... (skipped) ; C = B with 512-thread blocks (8 KiB per round)
//   D  one-shot: grid of E blocks, block e writes env e and exits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_wave_per_env(uint8_t* out, int n_envs, int env_bytes) {
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
template <int T>
__global__ void __launch_bounds__(T) k_block_per_env(uint8_t* out, int n_envs, int env_bytes) {
    for (int e = blockIdx.x; e < n_envs; e += gridDim.x) {
        const size_t base = (size_t)e * env_bytes;
        const int nchunks = env_bytes >> 4;
        const int k0 = -(int)((base >> 4) & (T - 1));  // rounds aligned to T * 16 bytes
        const u32x4 v = {(uint32_t)e, threadIdx.x, 1u, 0x01000100u};
        for (int k = k0 + (int)threadIdx.x; k < nchunks; k += T)
            if (k >= 0) *(u32x4*)(out + base + ((size_t)k << 4)) = v;
    }
}
int main() {
    const int E = 65536, B = 25200;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    uint8_t* bufs[6];
    for (int i = 0; i < 6; i++) if (hipMalloc(&bufs[i], (size_t)E * B) != hipSuccess) return 1;
    auto run = [&](int shape, uint8_t* buf) {
        float sum = 0;
        for (int rep = 0; rep < 60; rep++) {
            (void)hipEventRecord(a);
            if (shape == 0) hipLaunchKernelGGL(k_wave_per_env, dim3(2048), dim3(256), 0, 0, buf, E, B);
            if (shape == 1) hipLaunchKernelGGL(k_block_per_env<256>, dim3(2048), dim3(256), 0, 0, buf, E, B);
            if (shape == 2) hipLaunchKernelGGL(k_block_per_env<512>, dim3(1024), dim3(512), 0, 0, buf, E, B);
            if (shape == 3) hipLaunchKernelGGL(k_block_per_env<256>, dim3(E), dim3(256), 0, 0, buf, E, B);
            if (shape == 4) hipLaunchKernelGGL(k_block_per_env<256>, dim3(1024), dim3(256), 0, 0, buf, E, B);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (rep >= 20) sum += ms;
        }
        return sum / 40;
    };
    const char* names[5] = {"A wave/env 2048x256", "B block/env 2048x256", "C block/env 1024x512", "D one-shot Ex256", "E block/env 1024x256"};
    for (int i = 0; i < 6; i++) {
        printf("alloc %d:", i);
        for (int s = 0; s < 5; s++) printf("  %s %.3f", names[s], run(s, bufs[i]));
        printf("\n");
    }
    return 0;
}
