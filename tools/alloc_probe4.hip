// alloc_probe4.hip — does 4 KiB alignment of what a BLOCK writes per pass matter (block cooperating on one env)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0 wave per env persistent (reference)   MODE 1 block per env, rows interleaved, 1-KiB-aligned wave instrs (unaligned 4 KiB groups)
// MODE 2 block per env, the block's 4 waves write one 4-KiB-ALIGNED group per pass
template <int MODE>
__global__ void __launch_bounds__(256) k(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    if (MODE == 0) {
        for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
            const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + lane; kk < nchunks; kk += 64) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else {
        for (int e = blockIdx.x; e < n_envs; e += gridDim.x) {
            const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4;
            const int k0 = (MODE == 1) ? -(int)((base >> 4) & 63) : -(int)((base >> 4) & 255);
            for (int kk = k0 + (int)threadIdx.x; kk < nchunks; kk += 256) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    }
}
template <int MODE> float run(uint8_t* buf, int E, int B, int grid) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, buf, E, B); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep >= 3 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200; const size_t bytes = (size_t)E * B;
    std::vector<uint8_t*> bufs;
    for (int i = 0; i < 6; i++) { uint8_t* p; if (hipMalloc(&p, bytes) != hipSuccess) return 1; bufs.push_back(p); }
    printf("%-16s %10s %12s %12s %12s %12s\n", "buffer", "wave/env", "blk/env 1K", "blk/env 4K", "1shot 1K", "1shot 4K");
    for (auto p : bufs)
        printf("%p %10.3f %12.3f %12.3f %12.3f %12.3f\n", (void*)p, run<0>(p, E, B, 2048), run<1>(p, E, B, 2048), run<2>(p, E, B, 2048),
               run<1>(p, E, B, 65536), run<2>(p, E, B, 65536));
    return 0;
}
