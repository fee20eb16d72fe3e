// store_bw2.hip — microbenchmark 2: how much do (a) a per-env dependent load + vmcnt drain, (b) small 2-byte
// side stores, (c) raw-barrier producer/consumer structure cost the 16 B/lane store stream?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// VAR 0: stores only; 1: vector load of the env's record at the top of every env (value feeds the stores);
//     2: same load but scalar (uniform address -> s_load, lgkmcnt only); 3: VAR1 + 176 two-byte side stores per env;
//     4: VAR1 + side stores as 22 sixteen-byte stores; 5: load issued one env ahead (software prefetch)
template <int VAR>
__global__ void __launch_bounds__(256) k_store(uint8_t* out, const uint32_t* rec, uint16_t* meta, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t pre = 0;
    if (VAR == 5) { int e0 = blockIdx.x * 4 + wave; if (e0 < n_envs) pre = rec[(size_t)e0 * 32 + (lane & 31)]; }
    for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
        uint32_t r = 0;
        if (VAR == 1 || VAR == 3 || VAR == 4) r = rec[(size_t)e * 32 + (lane & 31)];
        if (VAR == 2) r = rec[(size_t)e * 32 + 3];
        if (VAR == 5) {
            r = pre;
            int en = e + gridDim.x * 4;
            if (en < n_envs) pre = rec[(size_t)en * 32 + (lane & 31)];
        }
        const size_t base = (size_t)e * env_bytes;
        const int nchunks = env_bytes >> 4;
        const int k0 = -(int)((base >> 4) & 63);
        if (VAR == 3) for (int i = lane; i < 176; i += 64) meta[(size_t)e * 176 + i] = (uint16_t)(r + i);
        if (VAR == 4) if (lane < 22) { u32x4 v = {r, r, r, r}; *(u32x4*)((uint8_t*)meta + (size_t)e * 352 + lane * 16) = v; }
        u32x4 v = {(uint32_t)e, (uint32_t)lane, r, 0x01000100u};
        for (int k = k0 + lane; k < nchunks; k += 64)
            if (k >= 0) *(u32x4*)(out + base + ((size_t)k << 4)) = v;
    }
}

// producer/consumer: wave 0 loads the records of the block's next 3 envs into LDS while waves 1..3 stream the
// current 3; raw s_barrier between phases (no vmcnt drain for the consumers).
__global__ void __launch_bounds__(256) k_pc(uint8_t* out, const uint32_t* rec, int n_envs, int env_bytes) {
    __shared__ uint32_t slot[2][3][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = gridDim.x * 3;
    int phase = 0;
    if (wave == 0) {
        for (int j = 0; j < 3; j++) { int e = blockIdx.x * 3 + j; if (e < n_envs && lane < 32) slot[0][j][lane] = rec[(size_t)e * 32 + lane]; }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __builtin_amdgcn_s_barrier();
    for (int eb = blockIdx.x * 3; eb < n_envs; eb += stride, phase ^= 1) {
        if (wave == 0) {
            for (int j = 0; j < 3; j++) { int e = eb + stride + j; if (e < n_envs && lane < 32) slot[phase ^ 1][j][lane] = rec[(size_t)e * 32 + lane]; }
            __builtin_amdgcn_s_waitcnt(0);
        } else {
            const int e = eb + wave - 1;
            if (e < n_envs) {
                const uint32_t r = slot[phase][wave - 1][lane & 31];
                const size_t base = (size_t)e * env_bytes;
                const int nchunks = env_bytes >> 4;
                const int k0 = -(int)((base >> 4) & 63);
                u32x4 v = {(uint32_t)e, (uint32_t)lane, r, 0x01000100u};
                for (int k = k0 + lane; k < nchunks; k += 64)
                    if (k >= 0) *(u32x4*)(out + base + ((size_t)k << 4)) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        __builtin_amdgcn_s_barrier();
    }
}

int main() {
    const int E = 65536, B = 25200;
    uint8_t* buf; uint32_t* rec; uint16_t* meta;
    CHK(hipMalloc(&buf, (size_t)E * B)); CHK(hipMalloc(&rec, (size_t)E * 128)); CHK(hipMalloc(&meta, (size_t)E * 352));
    CHK(hipMemset(rec, 1, (size_t)E * 128));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int var = 0; var < 7; var++) {
        for (int blocks : {2048, 2731}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                CHK(hipEventRecord(a));
                switch (var) {
                    case 0: hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 1: hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 2: hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 3: hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 4: hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 5: hipLaunchKernelGGL(k_store<5>, dim3(blocks), dim3(256), 0, 0, buf, rec, meta, E, B); break;
                    case 6: hipLaunchKernelGGL(k_pc, dim3(blocks), dim3(256), 0, 0, buf, rec, E, B); break;
                }
                CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
                float ms; CHK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("var=%d blocks=%5d : %.3f ms  %.2f TB/s\n", var, blocks, best, (double)E * B / best / 1e9);
        }
    }
    return 0;
}
