// store_bw4.hip — which assignment of output bytes to waves reaches hipMemset's write rate (6.3 TB/s)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: wave per env (baseline)            MODE 1: block per env, waves interleave 1-KiB rows
// MODE 2: wave per env, env order permuted so that resident waves are far apart
// MODE 3: flat stream: block b, iteration it writes chunk (it*gridDim + b)*T  (= flat grid-stride fill)
// MODE 4: flat stream in per-block contiguous slabs: block b owns [b*slab, (b+1)*slab)
template <int MODE, int T>
__global__ void __launch_bounds__(T) k(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = T / 64;
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    if (MODE == 0 || MODE == 2) {
        for (int e0 = blockIdx.x * WPB + wave; e0 < n_envs; e0 += gridDim.x * WPB) {
            int e = e0;
            if (MODE == 2) e = (e0 & 7) * (n_envs >> 3) + (e0 >> 3);
            const size_t base = (size_t)e * env_bytes;
            const int nchunks = env_bytes >> 4;
            const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + lane; kk < nchunks; kk += 64)
                if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else if (MODE == 1) {
        for (int e = blockIdx.x; e < n_envs; e += gridDim.x) {
            const size_t base = (size_t)e * env_bytes;
            const int nchunks = env_bytes >> 4;
            const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + wave * 64 + lane; kk < nchunks; kk += 64 * WPB)
                if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else if (MODE == 3) {
        const size_t n = (size_t)n_envs * env_bytes / 16;
        for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) ((u32x4*)out)[i] = v;
    } else {
        const size_t n = (size_t)n_envs * env_bytes / 16;
        const size_t slab = (n + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * slab, hi = lo + slab < n ? lo + slab : n;
        for (size_t i = lo + threadIdx.x; i < hi; i += T) ((u32x4*)out)[i] = v;
    }
}
template <int MODE, int T>
float run(uint8_t* buf, int blocks, int E, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 8; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, T>), dim3(blocks), dim3(T), 0, 0, buf, E, B);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200;
    uint8_t* buf; CHK(hipMalloc(&buf, (size_t)E * B));
    const double gb = (double)E * B / 1e9; float t;
#define R(desc, MODE, T, BL) t = run<MODE, T>(buf, BL, E, B); printf("%-52s : %.3f ms %.2f TB/s\n", desc, t, gb / t);
    R("wave per env, 2048x256", 0, 256, 2048)
    R("block(256) per env, 2048 blocks", 1, 256, 2048)
    R("block(256) per env, 65536 blocks", 1, 256, 65536)
    R("block(1024) per env, 512 blocks", 1, 1024, 512)
    R("block(1024) per env, 65536 blocks", 1, 1024, 65536)
    R("wave per env, permuted env order", 2, 256, 2048)
    R("flat grid-stride, 2048x256", 3, 256, 2048)
    R("flat grid-stride, 16384x256", 3, 256, 16384)
    R("flat grid-stride, 65536x256", 3, 256, 65536)
    R("flat grid-stride, 4096x1024", 3, 1024, 4096)
    R("flat slabs, 2048x256", 4, 256, 2048)
    R("flat slabs, 16384x256", 4, 256, 16384)
    R("flat slabs, 65536x256 (25 KB per block = 1 env)", 4, 256, 65536)
    R("flat slabs, 65536x64", 4, 64, 65536)
    return 0;
}
