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

// host-side helpers defined in ctf_policy.hip
__attribute__((visibility("hidden"))) int ctf_policy_fail(const char* msg);      // sets ctf_policy_last_error(), returns -1
__attribute__((visibility("hidden"))) int ctf_policy_cus(int device_id);          // compute units of a device (cached), 0 on error
