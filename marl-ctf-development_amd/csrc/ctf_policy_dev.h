// ctf_policy_dev.h — device-side helpers shared by the policy kernels (ctf_policy.hip, ctf_policy_fact.hip): vector types, the
// packed tanh, the hand-placed (compiler-untracked) prefetch loads and their counted waits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ctf_policy.h"

#define WAVE 64
// Profiling-only ablations (never defined in the shipped build; tools/ablate_policy.sh):
//   bit0 no activation stores, bit1 no exp/rcp in tanh, bit2 no h0 update, bit3 every operand read from one LDS address,
//   bit4 every sample of a wave stored to the same row (store instructions without the HBM traffic),
//   bit5 team kernel: half of the shared activation stores skipped, bit6 team kernel: rows env-major
#ifndef POL_ABLATE
#define POL_ABLATE 0
// host side: make device_id current for the duration of an entry point, then put the caller's device back
struct DeviceScope {
    int prev = -1, want;
    bool ok = true;
    explicit DeviceScope(int device_id) : want(device_id) {
        if (hipGetDevice(&prev) != hipSuccess) ok = false;
        else if (prev != want && hipSetDevice(want) != hipSuccess) ok = false;
    }
    ~DeviceScope() {
        if (ok && prev != want) (void)hipSetDevice(prev);
    }
};

#endif
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
    // TRAIN instantiation only (ctf_policy_features_train): what a backward pass needs beside the activation row, channels-last
    uint16_t* h0_out;         // bf16 [S][G*G][16]: the one-hot input image (planes C..15 are zero)
    uint16_t* h1_out;         // bf16 [S][(G-2)^2][16]: tanh(conv1)
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two activations -> one packed bf16 pair: 2 v_exp_f32, v_pk_add_f32, 2 v_rcp_f32, v_pk_fma_f32, v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t tanh2_pack(float z0, float z1) {  // z = x * 2 log2(e)
    if (POL_ABLATE & 2) return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){z0, z1}, bf16x2_t));
    f32x2_t e = {__builtin_amdgcn_exp2f(z0), __builtin_amdgcn_exp2f(z1)};
    e = e + 1.0f;
    const f32x2_t r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    const f32x2_t m2 = {-2.0f, -2.0f}, one = {1.0f, 1.0f};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(__builtin_elementwise_fma(m2, r, one), bf16x2_t));
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {  // round to nearest even
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }

__host__ __device__ inline int pol_h0_bytes(int G) { return G * G * 32; }
__host__ __device__ inline int pol_h1_bytes(int G) { return (((G - 2) * (G - 2) + 15) / 16) * 16 * 32; }

// One lane's code bytes of a sample (cells lane, lane + 64, ...: at most 4 when G*G <= 256).  Kept as separate
// registers until they are used, so that the loads can stay in flight for a whole sample.
template <int NP>
struct PolCodes {
    uint32_t b[NP];
};
template <int NP>
__device__ __forceinline__ PolCodes<NP> pol_load_codes(const uint8_t* cp, int lane, int GG) {
    PolCodes<NP> v;
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const int c = lane + WAVE * q;
        v.b[q] = (c < GG) ? (uint32_t)cp[c] : 0u;
    }
    return v;
}

// Loads the compiler does not track (the idiom of k_observe): left to itself it sinks a "prefetch" down to its first use, i.e.
// BEHIND the sample's stores, and then waits with vmcnt(0) — which also drains every one of those stores before the next
// sample may start.  Issued by hand at the top of a sample and waited for, at the END OF THE SAME ITERATION, with a COUNTED
// vmcnt (the counter retires in issue order; the stores issued since are younger), the next sample's inputs arrive while
// this one computes and the stores keep draining in the background.  (Waiting at the top of the NEXT iteration is a bug:
// the compiler may copy the destination registers at the back edge, before the data has landed.)
__device__ __forceinline__ uint32_t pol_async_ubyte(const uint8_t* ptr) {
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t pol_async_ushort(const uint16_t* ptr) {
    uint32_t v;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t pol_async_dword(const uint32_t* ptr) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int NP>
__device__ __forceinline__ PolCodes<NP> pol_async_codes(const uint8_t* cp, int lane, int GG) {
    PolCodes<NP> v;
#pragma unroll
    for (int q = 0; q < NP; q++) v.b[q] = pol_async_ubyte(cp + min(lane + WAVE * q, GG - 1));  // clamped, not predicated: every lane loads
    return v;
}
// Wait until at most CNT vector-memory operations are outstanding, tying the prefetched registers to the wait.  Every
// register is an operand EXACTLY ONCE: listing one lvalue twice makes the compiler satisfy the duplicates with v_mov copies
// of the load's destination placed BEFORE the s_waitcnt, i.e. copies of a register whose data has not landed (this was the
// G = 11 bug of round 1: NP = 2 listed b[0] three times).  tools/isa_lint.py checks the compiled ISA for that pattern.
template <int CNT, int NP>
__device__ __forceinline__ void pol_wait_codes(PolCodes<NP>& c) {
    static_assert(NP >= 1 && NP <= 4, "pol_wait_codes ties at most four code registers");
    if constexpr (NP == 1) asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(c.b[0]) : "n"(CNT) : "memory");
    else if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(c.b[0]), "+v"(c.b[1]) : "n"(CNT) : "memory");
    else if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(%c3)" : "+v"(c.b[0]), "+v"(c.b[1]), "+v"(c.b[2]) : "n"(CNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(c.b[0]), "+v"(c.b[1]), "+v"(c.b[2]), "+v"(c.b[3]) : "n"(CNT) : "memory");
}
// the same wait for further registers (an s_waitcnt that follows one with the same count costs nothing)
#define POL_WAIT_VM1(N, r0) asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(r0) : "n"(N) : "memory")

// ---- the network's tail (k_policy_head, and fused behind k_policy_fc1_patch)
#define HEAD_TILE 128
#define HEAD_XS_ROW (256 * 2 + 16)
#define HEAD_HS_ROW (128 * 2 + 16)
struct HeadArgs {
    const uint16_t* y1;       // bf16 [B][256]
    const u32x4_t* fc2_frag;  // [4 M-tiles][16 K-steps][64]
    const float* fc2_bias;    // [128] (scaled)
    const u32x4_t* head_frag; // [4 K-steps][64]
    const float* head_bias;   // [16]
    const float* mask;        // [B] decision: 1 => only actions 0..4, or NULL
    const int32_t* given;     // [B] actions to evaluate instead of sampling, or NULL
    int32_t* action;          // [B]
    float* logprob;           // [B]
    float* entropy;           // [B]
    float* value;             // [B]
    float* logits;            // [B][A] raw logits, or NULL
    int64_t B;
    int32_t A;
    uint64_t seed, offset;
};

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}
__device__ __forceinline__ float xlane(float v, int mask) { return __shfl_xor(v, mask, WAVE); }

// Stages b-e of the head on a tile of 128 samples whose tanh(fc1) rows lie in `xs` (LDS, rows of HEAD_XS_ROW bytes; `hs` aliases it):
// fc2, tanh, heads, mask, distribution, sampling.  `smp_of(r)` maps the tile's sample r to its global sample index (the Philox counter
// and the row of every output), or a negative number for a slot that holds no sample.  Shared by k_policy_head (smp = tile * 128 + r)
// and by k_policy_fc1_patch's fused epilogue (smp = the row of the tile's slot r).  Ends with every wave past its last LDS read of hs
// except for the caller's own barrier.
template <typename SmpOf>
__device__ __forceinline__ void head_stages_bcde(const HeadArgs& a, uint8_t* xs, uint8_t* hs, const u32x4_t (&w2)[16], const u32x4_t (&wh)[4],
                                                 const f32x16_t& bias2, const f32x4_t& biash, int wave, int lane, SmpOf smp_of) {
    const int n32 = lane & 31, hh = lane >> 5, n16 = lane & 15, g4 = lane >> 4;
    // ---- b. fc2: this wave's 32 channels x the tile's 128 samples
    f32x16_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = bias2;
#pragma unroll
    for (int s = 0; s < 16; s++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32x4_t b = *(const u32x4_t*)(xs + (32 * t + n32) * HEAD_XS_ROW + (2 * s + hh) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[s]), as_bf16x8(b), acc[t], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done reading xs: hs may overwrite it
    // ---- c. tanh -> hs[sample][channel]
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32x2_t o;
            o[0] = tanh2_pack(acc[t][4 * q], acc[t][4 * q + 1]);
            o[1] = tanh2_pack(acc[t][4 * q + 2], acc[t][4 * q + 3]);
            *(u32x2_t*)(hs + (32 * t + n32) * HEAD_HS_ROW + (32 * wave + 8 * q + 4 * hh) * 2) = o;
        }
    }
    __syncthreads();
    // ---- d. heads for this wave's 32 samples, e. distribution
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int r = 32 * wave + 16 * u + n16;
        f32x4_t out = biash;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const u32x4_t b = *(const u32x4_t*)(hs + r * HEAD_HS_ROW + (4 * s + g4) * 16);
            out = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wh[s]), as_bf16x8(b), out, 0, 0, 0);
        }
        const int64_t smp = smp_of(r);
        const bool live = smp >= 0;
        const float dec = (a.mask && live) ? a.mask[smp] : 0.0f;
        float l[4], mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int o = 4 * g4 + q;
            // agent_network.py:71-75: decision 1 -> mask_5 (actions 0..4), decision 0 -> all ones, anything else -> the
            // all-zero mask (every logit + -1e9: a uniform distribution in float32)
            const bool off = (dec == 1.0f) ? (o >= 5) : (dec != 0.0f);
            l[q] = (o < a.A) ? out[q] + (off ? -1e9f : 0.0f) : -INFINITY;
            mx = fmaxf(mx, l[q]);
        }
        mx = fmaxf(mx, xlane(mx, 16));
        mx = fmaxf(mx, xlane(mx, 32));
        float e[4], part = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) { e[q] = __expf(l[q] - mx); part += e[q]; }
        float tot = part + xlane(part, 16);
        tot += xlane(tot, 32);
        const float logz = mx + __logf(tot);
        float ent = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) if (4 * g4 + q < a.A) ent -= (e[q] / tot) * (l[q] - logz);
        ent += xlane(ent, 16);
        ent += xlane(ent, 32);
        // inverse CDF over the outputs in index order: lanes g4 = 0..3 hold consecutive blocks of 4
        const float p1 = xlane(part, 16), p2 = xlane(part, 32), p3 = xlane(p1, 32);  // partner sums: g4^1, g4^2, g4^3
        float below = 0.0f;  // sum of the blocks with a smaller g4
        if (g4 == 1) below = p1;            // block 0
        else if (g4 == 2) below = p2 + p3;  // blocks 0 and 1
        else if (g4 == 3) below = p1 + p2 + p3;
        int act;
        if (a.given) act = live ? a.given[smp] : 0;
        else {
            const float target = philox_uniform(a.seed, a.offset, (uint64_t)smp) * tot;
            float c = below;
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { c += e[q]; cnt += (4 * g4 + q < a.A && c <= target) ? 1 : 0; }
            cnt += __shfl_xor(cnt, 16, WAVE);
            cnt += __shfl_xor(cnt, 32, WAVE);
            const int last = (dec == 1.0f) ? min(4, a.A - 1) : a.A - 1;
            act = min(cnt, last);
        }
        float lp = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; q++) if (4 * g4 + q == act) lp = l[q] - logz;
        lp += xlane(lp, 16);
        lp += xlane(lp, 32);
        if (live) {
            if (g4 == 0) {
                a.action[smp] = act;
                a.logprob[smp] = lp;
                a.entropy[smp] = ent;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int o = 4 * g4 + q;
                if (o == a.A) a.value[smp] = out[q];
                if (a.logits && o < a.A) a.logits[smp * a.A + o] = out[q];
            }
        }
    }
}


// Deterministic mode (ctf_policy_set_deterministic, include/ctf_policy.h): with a workspace registered for the device, the weight /
// bias gradient kernels do not end in float atomics on the gradient (whose order of arrival differs from run to run) — every block
// stores its partial sums in its own slice of the workspace and a second launch adds the slices IN BLOCK ORDER.
struct DetWorkspace {
    float* ptr;      // NULL: off (atomics)
    int64_t floats;
};
__attribute__((visibility("hidden"))) DetWorkspace ctf_policy_det(int device_id);
// dst[i] += sum over b = 0 .. n_blocks - 1 (in that order, four interleaved chains) of part[b * stride + i], i < elems
__attribute__((visibility("hidden"))) hipError_t ctf_policy_det_reduce(const float* part, int n_blocks, int64_t stride, int elems, float* dst,
                                                                      hipStream_t st);

// host-side helpers defined in ctf_policy.hip
__attribute__((visibility("hidden"))) int ctf_policy_fail(const char* msg);      // sets ctf_policy_last_error(), returns -1
__attribute__((visibility("hidden"))) int ctf_policy_cus(int device_id);          // compute units of a device (cached), 0 on error
