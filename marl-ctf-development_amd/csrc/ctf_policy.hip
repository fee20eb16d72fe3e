// ctf_policy.hip — the convolutional front of the reference's policy / value network (agent_network.py:13-14,30-36:
// conv3x3(C->16) tanh, conv3x3(16->32) tanh, flatten ++ metadata) for hundreds of thousands of agents per launch, fed by
// the env's compact observation (ctf_observe_codes) instead of the 14x larger one-hot planes.
//
// One wave per sample, everything between the code bytes and the activation row stays on the CU:
//   h0  LDS bf16 [G*G cells][16 ch]      the one-hot input, one 32-byte row per cell, written straight from the codes
//   conv1 = 16x16x32 MFMAs: D[out ch][position] over K = (2 taps) x (16 in ch); A = weights, register-resident for the
//           whole launch; B = ds_read_b128 of h0 rows (a lane's 8 consecutive channels of one cell)
//   h1  LDS bf16 [G1*G1 positions][16 ch]  tanh(conv1), written 8 bytes per lane from the accumulator layout
//   conv2 = 32x32x16 MFMAs: D[out ch][position], one MFMA per tap (K = 16 in ch), B = ds_read_b128 of h1 rows
//   out HBM bf16 [sample][Kp]            tanh(conv2) as 8-byte stores in the order the accumulators hold it:
//           column ((c/4) * P2 + p) * 4 + c%4 for out channel c, position p — the fc1 weight's columns are permuted to this
//           order once on the host (policy_native.py), so no transpose happens anywhere; then the M metadata values
//           (f16 -> bf16) and zero padding up to Kp (a multiple of 32).
// tanh(x) = 1 - 2 / (2^(x * 2 log2 e) + 1): the factor 2 log2 e is folded into the conv weights and biases on the host,
// so an activation costs v_exp_f32 + v_add + v_rcp_f32 + v_fma.
// fc1 / fc2 / heads are plain GEMMs and stay with hipBLASLt (through torch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ctf_policy.h"

#define WAVE 64
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

struct PolicyArgs {
    const uint8_t* codes;     // u8 [E][N][GG]
    const uint16_t* meta;     // f16 [E][N][M]
    uint16_t* act;            // bf16 [n_sel * E][Kp]
    const u32x4_t* w1frag;    // [5][64] lanes x 8 bf16
    const float* b1;          // [16]
    const u32x4_t* w2frag;    // [9][64]
    const float* b2;          // [32]
    int32_t n_envs, N, G, M, Kp, n_sel;
    uint64_t sel_pack;        // nibble k = agent index of selection slot k
    uint32_t inv_g1, inv_g2;  // ceil(65536 / G1), ceil(65536 / G2): exact for the position ranges used (checked on the host)
};

__device__ __forceinline__ float tanh_from_scaled(float z) {  // z = x * 2 log2(e)
    const float e = __builtin_amdgcn_exp2f(z);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {  // round to nearest even
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}
__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }

__host__ __device__ inline int pol_h0_bytes(int G) { return G * G * 32; }
__host__ __device__ inline int pol_h1_bytes(int G) { return (((G - 2) * (G - 2) + 15) / 16) * 16 * 32; }

template <int TG>
__global__ void __launch_bounds__(256) k_policy_features(PolicyArgs a) {
    extern __shared__ uint32_t lds[];
    const int G = TG ? TG : a.G;
    const int G1 = G - 2, G2 = G - 4, GG = G * G, P1 = G1 * G1, P2 = G2 * G2;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    uint8_t* h0 = (uint8_t*)lds + wave * (pol_h0_bytes(G) + pol_h1_bytes(G));
    uint8_t* h1 = h0 + pol_h0_bytes(G);

    // ---- launch-lifetime registers: both convolutions' weights in MFMA A-operand order, and the biases
    u32x4_t w1[5], w2[9];
#pragma unroll
    for (int s = 0; s < 5; s++) w1[s] = a.w1frag[s * WAVE + lane];
#pragma unroll
    for (int t = 0; t < 9; t++) w2[t] = a.w2frag[t * WAVE + lane];
    f32x4_t bias1;
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = a.b1[(lane >> 4) * 4 + r];
    f32x16_t bias2;
#pragma unroll
    for (int r = 0; r < 16; r++) bias2[r] = a.b2[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];

    // conv1 B-operand geometry: lane = (position n = lane & 15, k-group g = lane >> 4): taps 2s + (g >> 1), channels 8 (g & 1) ..
    const int n1 = lane & 15, g1 = lane >> 4;
    int off1[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int tap = min(2 * s + (g1 >> 1), 8);  // "tap 9" has zero weights: any valid address
        off1[s] = ((tap / 3) * G + (tap % 3)) * 32 + (g1 & 1) * 16;
    }
    // conv2: lane = (position n = lane & 31, channel half h = lane >> 5)
    const int n2 = lane & 31, hh = lane >> 5;

    const int S = a.n_sel * a.n_envs;
    for (int s = blockIdx.x * wpb + wave; s < S; s += gridDim.x * wpb) {
        const int k = s / a.n_envs, e = s - k * a.n_envs;
        const int agent = (int)((a.sel_pack >> (4 * k)) & 15u);
        const size_t row = (size_t)e * a.N + agent;
        // ---- h0: one 32-byte one-hot row per cell
        const uint8_t* cp = a.codes + row * GG;
        for (int c = lane; c < GG; c += WAVE) {
            const uint32_t code = cp[c];
            const uint32_t ch = code & 0x7Fu;
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = ((ch >> 1) == (uint32_t)j && ch != 0) ? (0x3F80u << (16 * (ch & 1u))) : 0u;
            w[0] |= (code >> 7) ? 0x3F80u : 0u;
            u32x4_t* dst = (u32x4_t*)(h0 + c * 32);
            dst[0] = (u32x4_t){w[0], w[1], w[2], w[3]};
            dst[1] = (u32x4_t){w[4], w[5], w[6], w[7]};
        }
        // metadata: f16 -> bf16 pairs behind the conv features, zero padding to Kp
        uint16_t* arow = a.act + (size_t)s * a.Kp;
        {
            const int npair = (a.Kp - 32 * P2) >> 1;
            if (lane < npair) {
                uint32_t out = 0;
                if (lane < (a.M >> 1)) {
                    const uint32_t two = ((const uint32_t*)(a.meta + row * a.M))[lane];
                    const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(two & 0xFFFFu));
                    const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(two >> 16));
                    out = pack_bf16(lo, hi);
                }
                ((uint32_t*)(arow + 32 * P2))[lane] = out;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();

        // ---- conv1 + tanh -> h1
        const int T1 = (P1 + 15) >> 4;
#pragma unroll 1
        for (int t = 0; t < T1; t++) {
            const int p = 16 * t + n1, pc = min(p, P1 - 1);
            const int y = (int)(((uint32_t)pc * a.inv_g1) >> 16), x = pc - y * G1;
            const uint8_t* base = h0 + (y * G + x) * 32;
            f32x4_t acc = bias1;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const u32x4_t b = *(const u32x4_t*)(base + off1[q]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(b), acc, 0, 0, 0);
            }
            u32x2_t o;
            o[0] = pack_bf16(tanh_from_scaled(acc[0]), tanh_from_scaled(acc[1]));
            o[1] = pack_bf16(tanh_from_scaled(acc[2]), tanh_from_scaled(acc[3]));
            *(u32x2_t*)(h1 + p * 32 + g1 * 8) = o;  // rows up to 16 * T1 exist
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();

        // ---- conv2 + tanh -> activation row
        const int T2 = (P2 + 31) >> 5;
#pragma unroll 1
        for (int t = 0; t < T2; t++) {
            const int p = 32 * t + n2, pc = min(p, P2 - 1);
            const int y = (int)(((uint32_t)pc * a.inv_g2) >> 16), x = pc - y * G2;
            const uint8_t* base = h1 + (y * G1 + x) * 32 + hh * 16;
            f32x16_t acc = bias2;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const u32x4_t b = *(const u32x4_t*)(base + ((tap / 3) * G1 + (tap % 3)) * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(b), acc, 0, 0, 0);
            }
            if (p < P2) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x2_t o;
                    o[0] = pack_bf16(tanh_from_scaled(acc[4 * q]), tanh_from_scaled(acc[4 * q + 1]));
                    o[1] = pack_bf16(tanh_from_scaled(acc[4 * q + 2]), tanh_from_scaled(acc[4 * q + 3]));
                    *(u32x2_t*)(arow + ((2 * q + hh) * P2 + p) * 4) = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

static thread_local char g_perr[256];
extern "C" const char* ctf_policy_last_error(void) { return g_perr; }
static int pfail(const char* msg) {
    snprintf(g_perr, sizeof(g_perr), "%s", msg);
    return -1;
}

extern "C" int32_t ctf_policy_act_stride(int32_t grid_size, int32_t meta_len) {
    const int p2 = (grid_size - 4) * (grid_size - 4);
    return (32 * p2 + meta_len + 31) & ~31;
}

extern "C" int ctf_policy_features(const uint8_t* codes_dev, const uint16_t* meta_dev, int32_t n_envs, int32_t n_agents,
                                   int32_t grid_size, int32_t meta_len, const int32_t* agent_sel, int32_t n_sel,
                                   const void* conv1_frag_dev, const float* conv1_bias_dev, const void* conv2_frag_dev,
                                   const float* conv2_bias_dev, uint16_t* act_dev, int32_t device_id, void* stream) {
    if (!codes_dev || !meta_dev || !agent_sel || !conv1_frag_dev || !conv1_bias_dev || !conv2_frag_dev || !conv2_bias_dev || !act_dev)
        return pfail("null argument");
    if (grid_size < 5 || grid_size > 32) return pfail("grid_size outside 5..32");
    if (n_envs < 1 || n_agents < 1 || n_agents > 16 || n_sel < 1 || n_sel > 16) return pfail("n_envs / n_agents / n_sel out of range");
    if (meta_len < 2 || (meta_len & 1)) return pfail("meta_len must be even (2N + 6)");
    if (((uintptr_t)act_dev & 15) || ((uintptr_t)meta_dev & 3)) return pfail("act_dev must be 16-byte, meta_dev 4-byte aligned");
    if ((int64_t)n_envs * n_sel > 0x7FFFFFFF) return pfail("too many samples for one launch");
    PolicyArgs a;
    a.codes = codes_dev; a.meta = meta_dev; a.act = act_dev;
    a.w1frag = (const u32x4_t*)conv1_frag_dev; a.b1 = conv1_bias_dev;
    a.w2frag = (const u32x4_t*)conv2_frag_dev; a.b2 = conv2_bias_dev;
    a.n_envs = n_envs; a.N = n_agents; a.G = grid_size; a.M = meta_len; a.n_sel = n_sel;
    a.Kp = ctf_policy_act_stride(grid_size, meta_len);
    a.sel_pack = 0;
    for (int k = 0; k < n_sel; k++) {
        if (agent_sel[k] < 0 || agent_sel[k] >= n_agents) return pfail("agent_sel entry out of range");
        a.sel_pack |= (uint64_t)agent_sel[k] << (4 * k);
    }
    const int G1 = grid_size - 2, G2 = grid_size - 4;
    a.inv_g1 = (65536 + G1 - 1) / G1;
    a.inv_g2 = (65536 + G2 - 1) / G2;
    for (int p = 0; p < G1 * G1; p++)
        if ((int)(((uint32_t)p * a.inv_g1) >> 16) != p / G1) return pfail("internal: reciprocal of G-2 not exact");
    for (int p = 0; p < G2 * G2; p++)
        if ((int)(((uint32_t)p * a.inv_g2) >> 16) != p / G2) return pfail("internal: reciprocal of G-4 not exact");

    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    hipDeviceProp_t prop;
    static thread_local int cus_of[64];
    int n_cus = (device_id >= 0 && device_id < 64) ? cus_of[device_id] : 0;
    if (!n_cus) {
        if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return pfail("hipGetDeviceProperties failed");
        n_cus = prop.multiProcessorCount;
        if (device_id >= 0 && device_id < 64) cus_of[device_id] = n_cus;
    }
    const int per_wave = pol_h0_bytes(grid_size) + pol_h1_bytes(grid_size);
    int wpb = 4;
    while (wpb > 1 && wpb * per_wave > 64 * 1024) wpb >>= 1;
    const size_t sh = (size_t)wpb * per_wave;
    int per_cu = (int)((160 * 1024) / sh);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * wpb > 12) per_cu = 12 / wpb;  // 3 waves per SIMD: what the register budget allows
    const int S = n_envs * n_sel;
    int blocks = (S + wpb - 1) / wpb;
    if (blocks > n_cus * per_cu) blocks = n_cus * per_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    if (grid_size == 15) {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features<15>, dim3(blocks), dim3(wpb * WAVE), sh, st, a);
    } else if (grid_size == 11) {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features<11>, dim3(blocks), dim3(wpb * WAVE), sh, st, a);
    } else {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features<0>, dim3(blocks), dim3(wpb * WAVE), sh, st, a);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}
