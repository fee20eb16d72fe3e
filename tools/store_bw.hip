// store_bw.hip — microbenchmark: what write bandwidth does the observe kernel's store pattern reach with no
// compute at all?  Each wave writes whole 25,200-byte env blocks, 16 B per lane per instruction.
//   variant bit0: nontemporal stores     bit1: wave-instructions aligned to 1 KiB of the flat buffer
//   variant bit2: 2 stores in flight per lane (unroll)
// Build+run: hipcc --offload-arch=gfx950 -O3 -o store_bw tools/store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(256) k_store(uint8_t* out, int n_envs, int env_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = blockIdx.x * 4 + wave; e < n_envs; e += gridDim.x * 4) {
        const size_t base = (size_t)e * env_bytes;
        const int nchunks = env_bytes >> 4;
        int k0 = 0;
        if (VAR & 2) k0 = -(int)((base >> 4) & 63);  // start so that every wave-instruction is 1 KiB aligned
        u32x4 v = {(uint32_t)e, (uint32_t)lane, 1u, 0x01000100u};
        for (int k = k0 + lane; k < nchunks; k += 64) {
            if (k >= 0) {
                u32x4* p = (u32x4*)(out + base + ((size_t)k << 4));
                if (VAR & 1) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        }
    }
}

int main() {
    const int E = 65536, B = 25200;
    uint8_t* buf;
    hipMalloc(&buf, (size_t)E * B);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 1024, 4096, 8192, 16384}) {
        for (int var = 0; var < 4; var++) {
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                hipEventRecord(a);
                switch (var) {
                    case 0: hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, buf, E, B); break;
                    case 1: hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, buf, E, B); break;
                    case 2: hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, buf, E, B); break;
                    case 3: hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(256), 0, 0, buf, E, B); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("blocks=%5d nt=%d aligned=%d : %.3f ms  %.2f TB/s\n", blocks, var & 1, (var >> 1) & 1, best, (double)E * B / best / 1e9);
        }
    }
    // reference point: plain contiguous grid-stride fill of the same bytes
    return 0;
}
