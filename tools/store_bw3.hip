// store_bw3.hip — can the bare store stream go faster than one 1-KiB store per loop iteration?
//   U = stores issued back to back per iteration (unrolled), T = threads per block, envs handled wave-per-env
//   mode 1: each wave covers 2 envs interleaved (two independent streams)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int U, int T>
__global__ void __launch_bounds__(T) k_store(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = T / 64;
    for (int e = blockIdx.x * WPB + wave; e < n_envs; e += gridDim.x * WPB) {
        const size_t base = (size_t)e * env_bytes;
        const int nchunks = env_bytes >> 4;
        const int k0 = -(int)((base >> 4) & 63);
        const u32x4 v = {(uint32_t)e, (uint32_t)lane, 1u, 0x01000100u};
        for (int k = k0 + lane; k < nchunks; k += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int kk = k + 64 * u;
                if (kk >= 0 && kk < nchunks) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
            }
        }
    }
}
// whole buffer as one flat stream, grid-stride (what a memset-like kernel does)
__global__ void __launch_bounds__(256) k_flat(u32x4* out, size_t n) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

template <int U, int T>
float run(uint8_t* buf, int blocks, int E, int B, hipEvent_t a, hipEvent_t b) {
    float best = 1e9;
    for (int rep = 0; rep < 8; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_store<U, T>), dim3(blocks), dim3(T), 0, 0, buf, E, B);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int E = 65536, B = 25200;
    uint8_t* buf;
    CHK(hipMalloc(&buf, (size_t)E * B));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const double gb = (double)E * B / 1e9;
    float t;
    t = run<1, 256>(buf, 2048, E, B, a, b); printf("U=1 T=256 blocks=2048 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<2, 256>(buf, 2048, E, B, a, b); printf("U=2 T=256 blocks=2048 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<4, 256>(buf, 2048, E, B, a, b); printf("U=4 T=256 blocks=2048 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<8, 256>(buf, 2048, E, B, a, b); printf("U=8 T=256 blocks=2048 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<4, 256>(buf, 1024, E, B, a, b); printf("U=4 T=256 blocks=1024 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<4, 256>(buf, 512, E, B, a, b);  printf("U=4 T=256 blocks=512  : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<1, 64>(buf, 8192, E, B, a, b);  printf("U=1 T=64  blocks=8192 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<1, 64>(buf, 4096, E, B, a, b);  printf("U=1 T=64  blocks=4096 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<1, 512>(buf, 1024, E, B, a, b); printf("U=1 T=512 blocks=1024 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<1, 1024>(buf, 512, E, B, a, b); printf("U=1 T=1024 blocks=512 : %.3f ms %.2f TB/s\n", t, gb / t);
    t = run<1, 256>(buf, 65536 / 4, E, B, a, b); printf("U=1 T=256 blocks=16384 (1 env/wave) : %.3f ms %.2f TB/s\n", t, gb / t);
    {
        float best = 1e9; size_t n = (size_t)E * B / 16;
        for (int blocks : {2048, 4096, 16384}) {
            best = 1e9;
            for (int rep = 0; rep < 8; rep++) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k_flat, dim3(blocks), dim3(256), 0, 0, (u32x4*)buf, n);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("flat grid-stride fill blocks=%d : %.3f ms %.2f TB/s\n", blocks, best, gb / best);
        }
        CHK(hipMemsetAsync(buf, 0, (size_t)E * B, 0));
        (void)hipEventRecord(a); CHK(hipMemsetAsync(buf, 1, (size_t)E * B, 0)); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("hipMemsetAsync : %.3f ms %.2f TB/s\n", ms, gb / ms);
    }
    return 0;
}
