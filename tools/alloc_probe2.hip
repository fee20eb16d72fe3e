// alloc_probe2.hip — store patterns vs allocation kind: which pattern is fast on EVERY allocation?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: persistent, wave per env (the render's pattern)
// MODE 1: one-shot blocks, each writes CH contiguous KiB (torch fill-like), grid = bytes / (CH KiB)
// MODE 2: persistent 2048 blocks, block-contiguous CH KiB pieces taken in order (piece p -> block p % grid ... grid-stride over pieces)
template <int MODE, int CH>
__global__ void __launch_bounds__(256) k(uint8_t* out, size_t bytes, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {(uint32_t)blockIdx.x, (uint32_t)lane, 1u, 0x01000100u};
    if (MODE == 0) {
        for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
            const size_t base = (size_t)e * env_bytes; const int nchunks = env_bytes >> 4; const int k0 = -(int)((base >> 4) & 63);
            for (int kk = k0 + lane; kk < nchunks; kk += 64) if (kk >= 0) *(u32x4*)(out + base + ((size_t)kk << 4)) = v;
        }
    } else {
        const size_t piece = (size_t)CH * 1024, npieces = bytes / piece;
        for (size_t p = blockIdx.x; p < npieces; p += (MODE == 1 ? npieces : gridDim.x)) {
            u32x4* dst = (u32x4*)(out + p * piece);
#pragma unroll
            for (int j = 0; j < CH * 1024 / (256 * 16); j++) dst[j * 256 + threadIdx.x] = v;
        }
    }
}
template <int MODE, int CH> float run(uint8_t* buf, size_t bytes, int grid, int E, int B) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
        (void)hipEventRecord(a); hipLaunchKernelGGL((k<MODE, CH>), dim3(grid), dim3(256), 0, 0, buf, bytes, E, B); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (rep >= 3 && ms < best) best = ms;
    }
    return best;
}
int main() {
    const int E = 65536, B = 25200; const size_t bytes = (size_t)E * B;
    std::vector<uint8_t*> bufs;
    for (int i = 0; i < 6; i++) { uint8_t* p; if (hipMalloc(&p, bytes) != hipSuccess) return 1; bufs.push_back(p); }
    printf("%-16s %9s %9s %9s %9s %9s %9s %9s\n", "buffer", "wave/env", "1shot16K", "1shot64K", "1shot4K", "pers16K", "pers64K", "pers256K");
    for (auto p : bufs)
        printf("%p %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f %9.3f\n", (void*)p, run<0, 16>(p, bytes, 2048, E, B),
               run<1, 16>(p, bytes, (int)(bytes / (16 * 1024)), E, B), run<1, 64>(p, bytes, (int)(bytes / (64 * 1024)), E, B),
               run<1, 4>(p, bytes, (int)(bytes / (4 * 1024)), E, B), run<2, 16>(p, bytes, 2048, E, B), run<2, 64>(p, bytes, 2048, E, B),
               run<2, 256>(p, bytes, 2048, E, B));
    return 0;
}
