// ctf_policy_fact.hip — fc1 of the reference's network (agent_network.py:15,36-37) WITHOUT the per-agent activation matrix.
//
// Teammates that look through the same reversal see identical tile planes (gridworld_ctf.py:981-988); one agent's own-position bit
// reaches 3 x 3 conv1 outputs and 5 x 5 conv2 outputs.  fc1 is linear in tanh(conv2) = h2, so for an agent a of view v
//
//     fc1(h2_a ++ meta_a) = W_flat . h2_v  +  W[:, patch(a)] . (h2_a - h2_v)[patch(a)]  +  W_meta . meta_a  +  b
//
// with patch(a) = the 25 positions x 32 channels around the agent's own cell.  The first term is ONE GEMM row per (env, view) instead
// of one per agent (4 x fewer rows); the second is a K = 800 product whose weight columns depend only on the own cell.  Three kernels:
//
//   k_fact_hist / _scan / _assign   bucket the agents of a launch by own cell: every bucket is padded to whole tiles of 128 slots, so
//                                   that a tile's 128 agents share one 800 x 256 slice of fc1's weight
//   k_policy_features_fact          the shared-view front (k_policy_features_team's arithmetic, bit for bit): conv1 / conv2 once per
//                                   env -> ONE row of the view matrix (bf16 [E][32 * PP]); per agent the two patch tiles -> the agent's
//                                   patch row (bf16 [slot][25 * 32 deltas ++ metadata ++ pad]) = bf16(h2_a) - bf16(h2_v), in fp32, rounded
//                                   to bf16, staged in LDS and written as whole 16-byte pieces into the agent's bucket slot
//   k_policy_fc1_patch              per tile of 128 slots: D[256 out][128 slots] = W_patch(cell) . rows  (32x32x16 bf16 MFMA; the weight
//                                   fragments stream from L2 straight into registers, the rows through LDS), then
//                                   y1[agent] = bf16(D + (W_flat . h2_v)[env] + b) — the pre-activation ctf_policy_head consumes
//
// The view GEMM itself ([E][32 * PP] x [32 * PP][256], fp32 out) is the BLAS library's (policy_native.py).
#include "ctf_policy_dev.h"

#define FACT_MT 128      // slots per tile
#define FACT_BINS 256    // own cells (G * G <= 256)
#define FACT_ROW_PAD 16  // bytes of padding per staged row in LDS (the B-operand reads of 32 consecutive slots then spread over the banks)

// ------------------------------------------------------------------------------------------------
// bucketing by own cell
// ------------------------------------------------------------------------------------------------
struct BucketArgs {
    const uint16_t* selfcells;  // u16 [E][N]
    int32_t* hist;              // [FACT_BINS] zero on entry (the caller zeroes the work buffer ONCE; k_fact_scan leaves it zeroed)
    int32_t* cursor;            // [FACT_BINS]
    int32_t* n_tiles;           // [1]
    int32_t* tile_cell;         // [t_max]
    int32_t* slot_of;           // [A * E]: row k * E + e -> slot
    int32_t* row_of_slot;       // [t_max * FACT_MT]: slot -> row, -1 where a bucket is padded
    int32_t E, N, A, GG, t_max;
    uint64_t sel_pack;
};

__device__ __forceinline__ int fact_cell_of(const BucketArgs& a, int i) {
    const int k = i / a.E, e = i - k * a.E;
    const int c = a.selfcells[(size_t)e * a.N + (int)((a.sel_pack >> (4 * k)) & 15u)];
    return min(c, a.GG - 1);
}

#define FACT_ITEMS_PER_THREAD 8  // rows per thread of k_fact_hist / k_fact_assign: a block covers 2 048 consecutive rows
__global__ void __launch_bounds__(256) k_fact_hist(BucketArgs a) {
    __shared__ int32_t h[FACT_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int total = a.A * a.E;
    const int i0 = blockIdx.x * (256 * FACT_ITEMS_PER_THREAD);
#pragma unroll
    for (int q = 0; q < FACT_ITEMS_PER_THREAD; q++) {
        const int i = i0 + q * 256 + threadIdx.x;
        if (i < total) atomicAdd(&h[fact_cell_of(a, i)], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&a.hist[threadIdx.x], h[threadIdx.x]);
}

// one block of FACT_BINS threads: padded exclusive scan of the histogram -> bucket bases (in slots), the tile -> cell table, the
// "-1" marks of every bucket's padding slots; leaves the histogram zeroed for the next call
__global__ void __launch_bounds__(FACT_BINS) k_fact_scan(BucketArgs a) {
    __shared__ int32_t base[FACT_BINS + 1];
    const int t = threadIdx.x;
    const int count = a.hist[t];
    a.hist[t] = 0;
    const int padded = (count + FACT_MT - 1) / FACT_MT;  // tiles of this bucket
    base[t + 1] = padded;
    if (t == 0) base[0] = 0;
    __syncthreads();
    for (int d = 1; d < FACT_BINS; d <<= 1) {  // inclusive scan over base[1..]
        const int v = (t + 1 > d) ? base[t + 1 - d] : 0;
        __syncthreads();
        base[t + 1] += (t + 1 > d) ? v : 0;
        __syncthreads();
    }
    a.cursor[t] = base[t] * FACT_MT;
    const int n = min(base[FACT_BINS], a.t_max);
    if (t == 0) *a.n_tiles = n;
    for (int s = base[t] * FACT_MT + count; s < min(base[t + 1], a.t_max) * FACT_MT; s++) a.row_of_slot[s] = -1;  // < FACT_MT slots
    for (int tile = t; tile < n; tile += FACT_BINS) {  // the bucket whose tile range holds `tile`: last c with base[c] <= tile
        int lo = 0, hi = FACT_BINS - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (base[mid] <= tile) lo = mid; else hi = mid - 1;
        }
        a.tile_cell[tile] = lo;
    }
}

// slot of every row.  A block ranks its rows per cell in LDS, reserves one range per (block, cell) with ONE global atomic — all cells'
// reservations of a block are in flight together — and hands out slot = range start + rank.  (A first version took the ranges wave by
// wave, one returning atomic per distinct cell in sequence: 0.3 ms for 262 144 rows.)
#define FACT_ASSIGN_PER_THREAD FACT_ITEMS_PER_THREAD
__global__ void __launch_bounds__(256) k_fact_assign(BucketArgs a) {
    __shared__ int32_t cnt[FACT_BINS], start[FACT_BINS];
    const int total = a.A * a.E;
    const int i0 = blockIdx.x * (256 * FACT_ASSIGN_PER_THREAD);
    cnt[threadIdx.x] = 0;
    __syncthreads();
    int cell[FACT_ASSIGN_PER_THREAD], rank[FACT_ASSIGN_PER_THREAD];
#pragma unroll
    for (int q = 0; q < FACT_ASSIGN_PER_THREAD; q++) {
        const int i = i0 + q * 256 + threadIdx.x;
        cell[q] = i < total ? fact_cell_of(a, i) : -1;
        rank[q] = cell[q] >= 0 ? atomicAdd(&cnt[cell[q]], 1) : 0;
    }
    __syncthreads();
    if (cnt[threadIdx.x]) start[threadIdx.x] = atomicAdd(&a.cursor[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FACT_ASSIGN_PER_THREAD; q++) {
        const int i = i0 + q * 256 + threadIdx.x;
        if (cell[q] >= 0) {
            const int slot = start[cell[q]] + rank[q];
            a.slot_of[i] = slot;
            if (slot < a.t_max * FACT_MT) a.row_of_slot[slot] = i;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the shared-view front: one view row per env, one patch row per agent
// ------------------------------------------------------------------------------------------------
// Profiling-only phase trace (tools/trace_fact.py builds with -DPOL_TRACE=1): wave 0 of block 0 records s_memtime at the phase
// boundaries of its first 64 envs.
// Profiling-only ablations (never defined in the shipped build): bit0 no patch-row / view stores, bit1 patch row = the patched values
// (no subtraction), bit2 conv2 patch operands always from the shared h1 (no address select), bit3 no conv2 patches at all,
// bit4 no h0 rebuild and no shared conv1 (h1 keeps whatever LDS held: the upper bound of ANY cheaper conv1, round 5), bit5 no conv1 patches
#ifndef FACT_ABLATE
#define FACT_ABLATE 0
#endif
#ifndef POL_TRACE
#define POL_TRACE 0
#endif
#if POL_TRACE
__device__ uint64_t g_fact_trace[16 * 64];
extern "C" int ctf_policy_fact_trace_read(uint64_t* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fact_trace), sizeof(uint64_t) * 16 * 64) == hipSuccess ? 0 : -1;
}
#define FACT_STAMP(slot)                                                                        \
    do {                                                                                        \
        if (blockIdx.x == 0 && wave == 0 && trace_env < 64) {                                   \
            const uint64_t tstamp = __builtin_amdgcn_s_memtime();                               \
            if (lane == 0) g_fact_trace[trace_env * 16 + (slot)] = tstamp;                      \
        }                                                                                       \
    } while (0)
#else
#define FACT_STAMP(slot) do { } while (0)
#endif
struct FactFrontArgs {
    PolicyArgs p;               // p.act is unused
    const uint16_t* selfcells;  // u16 [E][N]
    const int32_t* slot_of;     // [A * E]
    uint16_t* view;             // bf16 [E][KV], KV = 32 * PP
    uint16_t* prow;             // bf16 [slots][KR]
    int32_t A, KV, KR;
};

// The next env's code bytes travel for a whole env: their four destination registers are PINNED (v156..v159, above what the kernel
// otherwise needs) — left to the allocator they were moved mid-flight once h2s went over h0 and the pressure around the conv1 tail rose
// (v_mov copies of registers whose loads had not landed: tools/isa_lint.py caught it before the kernel ever ran).
__device__ __forceinline__ void fact_async_codes4(const uint8_t* cp, int lane, int GG, uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
    asm volatile("global_load_ubyte %0, %1, off" : "={v156}"(b0) : "v"(cp + min(lane, GG - 1)) : "memory");
    asm volatile("global_load_ubyte %0, %1, off" : "={v157}"(b1) : "v"(cp + min(lane + WAVE, GG - 1)) : "memory");
    asm volatile("global_load_ubyte %0, %1, off" : "={v158}"(b2) : "v"(cp + min(lane + 2 * WAVE, GG - 1)) : "memory");
    asm volatile("global_load_ubyte %0, %1, off" : "={v159}"(b3) : "v"(cp + min(lane + 3 * WAVE, GG - 1)) : "memory");
}
__device__ __forceinline__ void fact_wait_codes4(uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
    asm volatile("s_waitcnt vmcnt(0)" : "+{v156}"(b0), "+{v157}"(b1), "+{v158}"(b2), "+{v159}"(b3) : : "memory");
}

// Per env, one wave:
//   1. h0 <- the shared planes (only the cells whose code changed since the wave's previous env);
//   2. shared conv1 -> tanh -> h1;
//   3. shared conv2 -> tanh -> the env's view row (HBM) and h2s (LDS); beside it, for every agent, the 3 x 3 conv1 PATCH: the same
//      operand reads of h0 with the agent's own-position bit OR-ed into the one operand register that holds it -> hp[agent] (LDS).
//      Nothing is written into h0 or h1, so the (up to four) agents' chains run side by side;
//   4. for every agent the 5 x 5 conv2 PATCH: a lane's operand of a tap comes from hp[agent] where the tap falls on the agent's 3 x 3
//      conv1 patch and from the shared h1 elsewhere (one address select per tap); the patch row = bf16(tanh) - h2s, rounded to bf16,
//      is staged in LDS (over h1, which is dead by then) and leaves as whole 16-byte pieces: 13 full lines into the agent's slot.
// Every output is the sum k_policy_features forms, in the same order: view and patch values equal that kernel's bit for bit.
template <int TG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_policy_features_fact(FactFrontArgs fa) {
    extern __shared__ uint32_t lds[];
    const PolicyArgs& a = fa.p;
    constexpr int G = TG, G1 = G - 2, G2 = G - 4, GG = G * G, P1 = G1 * G1, P2 = G2 * G2, PP = ((P2 + 31) >> 5) << 5;
    constexpr int NP = (GG + WAVE - 1) / WAVE;
    static_assert(GG <= 256, "one dword of code bytes per lane");
    const int A = fa.A;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    const int KRB = fa.KR * 2;  // bytes of a patch row
    constexpr int H2R = 72;  // bytes per position of h2s and of a staged patch row: 64 + 8, so that 16 lanes' 8-byte pieces cover all banks once
    constexpr int PSTB = 25 * H2R + 64;  // a staged row: 25 positions, then the 32 metadata / padding elements
    constexpr int H2S = PP * H2R, HPB = 9 * 32;
    constexpr int H1B = (((G - 2) * (G - 2) + 15) / 16) * 16 * 32;
    constexpr int STAGE_REGION = (H1B + 4 * HPB) > 4 * PSTB ? (H1B + 4 * HPB) : 4 * PSTB;  // h1 ++ hp, later the staged patch rows
    constexpr int H0H2 = G * G * 32 > H2S ? G * G * 32 : H2S;  // h0, then (once the last conv1 operand has been read) h2s over it
    constexpr int PER_WAVE = H0H2 + STAGE_REGION;
    uint8_t* h0 = (uint8_t*)lds + wave * PER_WAVE;
    uint8_t* h2s = h0;                    // bf16 [PP positions][32 channels]: tanh(conv2) of the shared view — written over h0, which is
                                          // rebuilt from the code bytes for every env (two waves per SIMD are worth more than the
                                          // incremental update of a persistent image: 1.25 -> 1.04 ms with the LDS this frees)
    uint8_t* h1 = h0 + H0H2;
    uint8_t* hp = h1 + pol_h1_bytes(G);   // bf16 [4 agents][9 positions][16 channels]: tanh(conv1) on each agent's 3 x 3 patch
    uint8_t* pst = h1;                    // the env's patch rows [4][KR], staged over h1 / hp once the last operand has been read
    static_assert(STAGE_REGION >= 4 * PSTB && STAGE_REGION >= H1B + 4 * HPB, "h1 ++ hp and the staged patch rows share one region");
    constexpr int H0A = GG * 16;
    const int H1A = pol_h1_bytes(G) / 2;

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

    const int n1 = lane & 15, g1 = lane >> 4;
    int off1[5], tap1[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        tap1[s] = min(2 * s + (g1 >> 1), 8);  // "tap 9" has zero weights: any valid address
        off1[s] = ((tap1[s] / 3) * G + (tap1[s] % 3)) * 16 + (g1 & 1) * H0A;
    }
    const int y1_0 = (int)(((uint32_t)n1 * a.inv_g1) >> 16), x1_0 = n1 - y1_0 * G1;
    constexpr int dy1 = 16 / G1, dx1 = 16 - dy1 * G1;
    uint8_t* h1w = h1 + (g1 >> 1) * H1A + n1 * 16 + (g1 & 1) * 8;
    const int n2 = lane & 31, hh = lane >> 5;
    const int j1 = min(n1, 8), pdy1 = j1 / 3, pdx1 = j1 - 3 * pdy1;
    const int j2 = min(n2, 24), pdy2 = j2 / 5, pdx2 = j2 - 5 * pdy2;

    const int e_first = blockIdx.x * wpb + wave, e_stride = gridDim.x * wpb;
    const int ag0 = (int)(a.sel_pack & 15u);
    const int mpairs = (fa.KR - 800) >> 1;  // dwords behind the 800 patch values: metadata pairs, then zeros

    static_assert(NP <= 4, "four pinned code registers");
    uint32_t nc0 = 0, nc1 = 0, nc2 = 0, nc3 = 0;  // the next env's code bytes (cells lane, lane + 64, ...)
    if (e_first < a.n_envs) {
        const PolCodes<NP> first = pol_load_codes<NP>(a.codes + ((size_t)e_first * a.N + ag0) * GG, lane, GG);
        nc0 = first.b[0];
        if (NP > 1) nc1 = first.b[NP > 1 ? 1 : 0];
        if (NP > 2) nc2 = first.b[NP > 2 ? 2 : 0];
        if (NP > 3) nc3 = first.b[NP > 3 ? 3 : 0];
    }

    int trace_env = 0;
    (void)trace_env;
    for (int e = e_first; e < a.n_envs; e += e_stride, trace_env++) {
        FACT_STAMP(0);
        PolCodes<NP> cur;
        cur.b[0] = nc0;
        if (NP > 1) cur.b[NP > 1 ? 1 : 0] = nc1;
        if (NP > 2) cur.b[NP > 2 ? 2 : 0] = nc2;
        if (NP > 3) cur.b[NP > 3 ? 3 : 0] = nc3;
        fact_async_codes4(a.codes + ((size_t)min(e + e_stride, a.n_envs - 1) * a.N + ag0) * GG, lane, GG, nc0, nc1, nc2, nc3);
        uint32_t scw[4], metaw[4], slotw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int kk = min(k, A - 1);  // k >= A: agent A - 1 again
            const size_t row = (size_t)e * a.N + (int)((a.sel_pack >> (4 * kk)) & 15u);
            scw[k] = pol_async_ushort(fa.selfcells + row);
            metaw[k] = pol_async_dword((const uint32_t*)(a.meta + row * a.M) + min(lane, (a.M >> 1) - 1));
            slotw[k] = pol_async_dword((const uint32_t*)fa.slot_of + (size_t)kk * a.n_envs + e);
        }
        // ---- 1. h0 <- the shared planes (own-position bits stripped), from scratch: the previous env's h2s lies over it
        if (!(FACT_ABLATE & 16)) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < (GG * 2 + WAVE - 1) / WAVE; q++)
                if (lane + WAVE * q < GG * 2) ((u32x4_t*)h0)[lane + WAVE * q] = z;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int c = lane + WAVE * q;
            const uint32_t n = cur.b[q] & 0x7Fu;
            if (!(FACT_ABLATE & 16) && c < GG && n) *(uint16_t*)(h0 + c * 16 + (n >> 3) * H0A + (n & 7u) * 2) = 0x3F80;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        FACT_STAMP(1);

        // ---- 2. shared conv1 + tanh -> h1 (two tiles in flight)
        constexpr int T1 = (FACT_ABLATE & 16) ? 0 : ((P1 + 15) >> 4);
        int x1 = x1_0, cell1 = y1_0 * G + x1_0;
        int t = 0;
#pragma unroll 1
        for (; t + 1 < T1; t += 2) {
            const uint8_t* base_a = h0 + cell1 * 16;
            x1 += dx1;
            cell1 += dy1 * G + dx1;
            if (x1 >= G1) { x1 -= G1; cell1 += G - G1; }
            const uint8_t* base_b = h0 + cell1 * 16;
            x1 += dx1;
            cell1 += dy1 * G + dx1;
            if (x1 >= G1) { x1 -= G1; cell1 += G - G1; }
            f32x4_t acc_a = bias1, acc_b = bias1;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const u32x4_t ba = *(const u32x4_t*)(base_a + off1[q]);
                const u32x4_t bb = *(const u32x4_t*)(base_b + off1[q]);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(ba), acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(bb), acc_b, 0, 0, 0);
            }
            u32x2_t o;
            o[0] = tanh2_pack(acc_a[0], acc_a[1]);
            o[1] = tanh2_pack(acc_a[2], acc_a[3]);
            *(u32x2_t*)(h1w + 16 * t * 16) = o;
            o[0] = tanh2_pack(acc_b[0], acc_b[1]);
            o[1] = tanh2_pack(acc_b[2], acc_b[3]);
            *(u32x2_t*)(h1w + 16 * (t + 1) * 16) = o;
        }
        if (t < T1) {
            const uint8_t* base_a = h0 + cell1 * 16;
            f32x4_t acc_a = bias1;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const u32x4_t ba = *(const u32x4_t*)(base_a + off1[q]);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(ba), acc_a, 0, 0, 0);
            }
            u32x2_t o;
            o[0] = tanh2_pack(acc_a[0], acc_a[1]);
            o[1] = tanh2_pack(acc_a[2], acc_a[3]);
            *(u32x2_t*)(h1w + 16 * t * 16) = o;
        }
        FACT_STAMP(2);
        // This env's own cells / metadata / slots and the next env's codes, issued at the top (this also drains the previous env's
        // stores, which have had the whole of conv1 to land).
        {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(scw[0]), "+v"(scw[1]), "+v"(scw[2]), "+v"(scw[3]), "+v"(metaw[0]), "+v"(metaw[1]), "+v"(metaw[2]),
                           "+v"(metaw[3]), "+v"(slotw[0]), "+v"(slotw[1]), "+v"(slotw[2]), "+v"(slotw[3])
                         :
                         : "memory");
            fact_wait_codes4(nc0, nc1, nc2, nc3);  // every register exactly once (see pol_wait_codes)
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        FACT_STAMP(3);

        // ---- 3a. every agent's 3 x 3 conv1 patch -> hp (reads h0 only; the own-position bit is OR-ed into the operand register)
        int syv[4], sxv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sc = __builtin_amdgcn_readfirstlane((int)scw[k]);
            syv[k] = sc / G;
            sxv[k] = sc - syv[k] * G;
        }
        if (!(FACT_ABLATE & 32)) {
            f32x4_t acc[4];
            const uint8_t* base[4];
            int own_tap[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int oy1 = syv[k] - 2 + pdy1, ox1 = sxv[k] - 2 + pdx1;
                const int cy1 = min(max(oy1, 0), G1 - 1), cx1 = min(max(ox1, 0), G1 - 1);
                base[k] = h0 + (cy1 * G + cx1) * 16;
                const int ty = syv[k] - cy1, tx = sxv[k] - cx1;  // the tap of this lane's output that reads the own cell
                own_tap[k] = ((unsigned)ty < 3u && (unsigned)tx < 3u && !(g1 & 1)) ? 3 * ty + tx : -1;  // (channel 0 sits in half 0)
                acc[k] = bias1;
            }
#pragma unroll
            for (int q = 0; q < 5; q++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32x4_t b = *(const u32x4_t*)(base[k] + off1[q]);
                    b[0] |= (own_tap[k] == tap1[q]) ? 0x3F80u : 0u;  // channel 0 of the own cell = 1.0
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(b), acc[k], 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32x2_t o;
                o[0] = tanh2_pack(acc[k][0], acc[k][1]);
                o[1] = tanh2_pack(acc[k][2], acc[k][3]);
                if (n1 < 9) *(u32x2_t*)(hp + k * HPB + n1 * 32 + g1 * 8) = o;  // channels 4 g1 .. 4 g1 + 3 of patch position n1
            }
        }

        __builtin_amdgcn_s_waitcnt(0xC07F);  // every lane's conv1 operands have been read: h2s may overwrite h0
        __builtin_amdgcn_wave_barrier();
        FACT_STAMP(4);
        // ---- 3b. shared conv2 + tanh -> the env's ONE view row (all positions) and h2s
        uint8_t* const vrow = (uint8_t*)fa.view + (size_t)e * fa.KV * 2;
        constexpr int T2 = (P2 + 31) >> 5;
        static_assert((T2 & 1) == 0, "tile pairs");
#pragma unroll 1
        for (int t2 = 0; t2 < T2; t2 += 2) {
            const int pa = 32 * t2 + n2, pb = pa + 32;
            const int pca = min(pa, P2 - 1), pcb = min(pb, P2 - 1);
            const int ya = (int)(((uint32_t)pca * a.inv_g2) >> 16), yb = (int)(((uint32_t)pcb * a.inv_g2) >> 16);
            const uint8_t* base_a = h1 + (ya * G1 + (pca - ya * G2)) * 16 + hh * H1A;
            const uint8_t* base_b = h1 + (yb * G1 + (pcb - yb * G2)) * 16 + hh * H1A;
            f32x16_t acc_a = bias2, acc_b = bias2;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int off = ((tap / 3) * G1 + (tap % 3)) * 16;
                const u32x4_t ba = *(const u32x4_t*)(base_a + off);
                const u32x4_t bb = *(const u32x4_t*)(base_b + off);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(ba), acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(bb), acc_b, 0, 0, 0);
            }
            const uint32_t lane_off = (uint32_t)(((3 + hh) * PP + pa) * 8);  // the four channel groups sit (2 q - 3) * PP * 8 around it
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32x2_t oa, ob;
                oa[0] = tanh2_pack(acc_a[4 * q], acc_a[4 * q + 1]);
                oa[1] = tanh2_pack(acc_a[4 * q + 2], acc_a[4 * q + 3]);
                ob[0] = tanh2_pack(acc_b[4 * q], acc_b[4 * q + 1]);
                ob[1] = tanh2_pack(acc_b[4 * q + 2], acc_b[4 * q + 3]);
                if (!(FACT_ABLATE & 1)) {
                    *(u32x2_t*)(vrow + lane_off + (2 * q - 3) * PP * 8) = oa;
                    *(u32x2_t*)(vrow + lane_off + (2 * q - 3) * PP * 8 + 256) = ob;
                }
                *(u32x2_t*)(h2s + pa * H2R + (8 * q + 4 * hh) * 2) = oa;  // channels 8 q + 4 hh .. + 3 of position pa
                *(u32x2_t*)(h2s + pb * H2R + (8 * q + 4 * hh) * 2) = ob;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        FACT_STAMP(5);

        // ---- 4. every agent's 5 x 5 conv2 patch, side by side; patch row = bf16(tanh) - h2s
        if (!(FACT_ABLATE & 8)) {
            f32x16_t acc[4];
            const uint8_t *shared_base[4], *priv_base[4];
            int cpos[4];
            uint32_t pmask[4];  // bit tap: this lane's operand of that tap lies on the agent's 3 x 3 conv1 patch
            bool ok2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int oy2 = syv[k] - 4 + pdy2, ox2 = sxv[k] - 4 + pdx2;
                ok2[k] = (unsigned)oy2 < (unsigned)G2 && (unsigned)ox2 < (unsigned)G2;
                const int cy2 = min(max(oy2, 0), G2 - 1), cx2 = min(max(ox2, 0), G2 - 1);
                cpos[k] = cy2 * G2 + cx2;
                shared_base[k] = h1 + (cy2 * G1 + cx2) * 16 + hh * H1A;
                const int r0y = cy2 - (syv[k] - 2), r0x = cx2 - (sxv[k] - 2);  // this lane's top-left input, relative to the 3 x 3 patch
                // rows ty (columns tx) in 0..2 with 0 <= r0 + t <= 2, as 3-bit sets
                const uint32_t rb = (r0y <= 0 ? (7u << min(-r0y, 3)) : (7u >> min(r0y, 3))) & 7u;
                const uint32_t cb = (r0x <= 0 ? (7u << min(-r0x, 3)) : (7u >> min(r0x, 3))) & 7u;
                pmask[k] = cb * ((rb & 1u) + 8u * ((rb >> 1) & 1u) + 64u * (rb >> 2));
                priv_base[k] = hp + k * HPB + (r0y * 3 + r0x) * 32 + hh * 16;
                acc[k] = bias2;
            }
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int ty = tap / 3, tx = tap % 3;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool patched = !(FACT_ABLATE & 4) && ((pmask[k] >> tap) & 1u);
                    const uint8_t* src = patched ? priv_base[k] + (ty * 3 + tx) * 32 : shared_base[k] + (ty * G1 + tx) * 16;
                    const u32x4_t b = *(const u32x4_t*)src;
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(b), acc[k], 0, 0, 0);
                }
            }
            // The rows leave through LDS: written straight from the accumulator layout they would be 8-byte pieces 64 bytes apart — 200
            // partial-line writes per agent, which cost the launch 0.45 ms of its 1.2 (FACT_ABLATE & 1); staged, a row is 13 whole lines.
            __builtin_amdgcn_s_waitcnt(0xC07F);  // every lane's operand reads of h1 / hp have returned: the stage may overwrite them
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < A) {
                    uint8_t* const prow_l = pst + k * PSTB;
                    const uint8_t* sh = h2s + cpos[k] * H2R + 8 * hh;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const u32x2_t s2 = *(const u32x2_t*)(sh + 16 * q);
                        const uint32_t p0 = tanh2_pack(acc[k][4 * q], acc[k][4 * q + 1]), p1 = tanh2_pack(acc[k][4 * q + 2], acc[k][4 * q + 3]);
                        u32x2_t d;
                        d[0] = pack_bf16(__uint_as_float(p0 << 16) - __uint_as_float(s2[0] << 16),
                                         __uint_as_float(p0 & 0xFFFF0000u) - __uint_as_float(s2[0] & 0xFFFF0000u));
                        d[1] = pack_bf16(__uint_as_float(p1 << 16) - __uint_as_float(s2[1] << 16),
                                         __uint_as_float(p1 & 0xFFFF0000u) - __uint_as_float(s2[1] & 0xFFFF0000u));
                        if (FACT_ABLATE & 2) d = (u32x2_t){p0 ^ s2[0], p1};
                        if (!ok2[k]) d = (u32x2_t){0u, 0u};  // a patch position outside the image: its weights are zero too
                        if (n2 < 25) *(u32x2_t*)(prow_l + n2 * H2R + (8 * q + 4 * hh) * 2) = d;
                    }
                    if (lane < mpairs) {  // metadata (f16 -> bf16) behind the patch values, zeros up to the row's end
                        const uint32_t mw = metaw[k];
                        uint32_t out = 0;
                        if (lane < (a.M >> 1)) {
                            const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(mw & 0xFFFFu));
                            const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(mw >> 16));
                            out = pack_bf16(lo, hi);
                        }
                        ((uint32_t*)(prow_l + 25 * H2R))[lane] = out;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // 104 sixteen-byte pieces per row: lanes 0..63 take pieces 0..63, lanes 0..39 pieces 64..103; piece c < 100 = quarter c & 3
            // of position c >> 2 (8-byte aligned in the padded stage: two 8-byte reads)
            const int c1 = lane + WAVE;
            const int src0 = (lane >> 2) * H2R + (lane & 3) * 16;
            const int src1 = c1 < 100 ? (c1 >> 2) * H2R + (c1 & 3) * 16 : 25 * H2R + (c1 - 100) * 16;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < A && !(FACT_ABLATE & 1)) {
                    const int slot = __builtin_amdgcn_readfirstlane((int)slotw[k]);
                    uint8_t* dst = (uint8_t*)fa.prow + (size_t)slot * KRB + lane * 16;
                    const uint8_t* src = pst + k * PSTB;
                    const u32x2_t a0 = *(const u32x2_t*)(src + src0), a1 = *(const u32x2_t*)(src + src0 + 8);
                    *(u32x4_t*)dst = (u32x4_t){a0[0], a0[1], a1[0], a1[1]};
                    if (c1 < (KRB >> 4)) {
                        const u32x2_t b0 = *(const u32x2_t*)(src + src1), b1 = *(const u32x2_t*)(src + src1 + 8);
                        *(u32x4_t*)(dst + WAVE * 16) = (u32x4_t){b0[0], b0[1], b1[0], b1[1]};
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // this env's LDS reads are done before the next env rewrites h0 / h1 / h2s / hp
        __builtin_amdgcn_wave_barrier();
        FACT_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------------
// the per-agent correction of fc1: grouped GEMM over the bucket tiles
// ------------------------------------------------------------------------------------------------
struct PatchArgs {
    const uint16_t* prow;        // bf16 [slots][KR]
    const int32_t* row_of_slot;  // [slots]
    const int32_t* tile_cell;    // [t_max]
    const int32_t* n_tiles;      // [1]
    const float* yview;          // f32 [E][256]: W_flat . h2_v (no bias), scaled by 2 log2 e like the weights
    const u32x4_t* wfrag;        // bf16 fragments [P2 + 2 blocks][2 k-steps][8 n-tiles][64 lanes]: blocks 0..P2-1 = conv2 positions,
                                 // block P2 = zeros (a patch position outside the image), block P2 + 1 = the metadata columns
    const float* bias;           // f32 [256] (scaled)
    uint16_t* y1;                // bf16 [A * E][256] (HEAD instantiation: unused)
    int32_t E, KR;
    HeadArgs head;               // HEAD instantiation: the network's tail runs on the tile while it is still in LDS (head.y1 / head.B unused)
};

// One block of four waves per tile of 128 slots; wave w owns the outputs 64 w .. 64 w + 63 (two 32-row A tiles) of all four 32-slot
// B tiles: 8 accumulator tiles = 128 registers.  K = 832 runs in 13 stages of 64 (two patch positions): a stage's rows [128][64] are
// fetched FACT_RDEPTH stages ahead into a register ring (they come from HBM: with two stages of lead every stage waited ~1 us for its
// rows, 27 us per tile against 5.5 us of MFMAs) and parked in an LDS double buffer one stage ahead; its weight fragments — 16 KB per
// patch position for the whole block, straight from L2 — are fetched FACT_WDEPTH stages ahead into a second ring.  One wave per SIMD
// (the accumulators alone are a quarter of the register file), so everything the MFMAs wait for is in flight long before it is used.
// Profiling-only ablations of the patch kernel: bit0 no row loads, bit1 no weight loads, bit2 no epilogue, bit3 no barriers
#ifndef FACT_PATCH_ABLATE
#define FACT_PATCH_ABLATE 0
#endif
#define FACT_NST 13     // stages: KR / 64 with KR = 832 (800 patch values + up to 32 metadata columns)
#ifndef FACT_WDEPTH
#define FACT_WDEPTH 3   // weight fragments (L2 hits: every tile of a cell reads the same 26 blocks) are fetched this many stages ahead
#endif
#ifndef FACT_RDEPTH
#define FACT_RDEPTH 5   // the rows come from HBM (the front wrote them): fetched this many stages (0.43 us of MFMAs each) ahead
#endif
// HEAD = true: the tile of 128 slots IS a tile of 128 samples of the network's tail (k_policy_head), so fc1's pre-activation never goes
// to HBM: the epilogue rounds it to bf16 exactly as the unfused path stores it, applies tanh and parks it in the tail's `xs` image, and
// stages b-e of the tail (ctf_policy_dev.h) run on it in place, every output written to the row of its slot — bit-identical results.
template <int TG, bool HEAD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_policy_fc1_patch(PatchArgs a) {
    constexpr int G = TG, G2 = G - 4, P2 = G2 * G2;
    constexpr int ROWB = 128 + FACT_ROW_PAD;      // bytes of a staged row: 64 k of one stage + padding
    constexpr int STAGEB = FACT_MT * ROWB;
    constexpr int KRB = FACT_NST * 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t patch_lds[];
    uint8_t* const stage = patch_lds;             // [2 * STAGEB]; HEAD: then xs [128][HEAD_XS_ROW]
    uint8_t* const xs = patch_lds + 2 * STAGEB;
    const int tile = blockIdx.x;
    if (tile >= *a.n_tiles) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int n32 = lane & 31, hh = lane >> 5;
    const int cell = a.tile_cell[tile];
    const int sy = cell / G, sx = cell - sy * G;
    const uint8_t* rows = (const uint8_t*)a.prow + (size_t)tile * FACT_MT * KRB;

    // this thread's four 16-byte pieces of a stage: piece c = tid + 256 i -> row c >> 3, 16-byte column c & 7
    const int tid = threadIdx.x;
    const uint8_t* my_rows = rows + (size_t)(tid >> 3) * KRB + (tid & 7) * 16;  // piece i: + 32 i rows
    uint8_t* my_park = stage + (tid >> 3) * ROWB + (tid & 7) * 16;
    u32x4_t pre[FACT_RDEPTH][4];           // ring: the rows of stage s live in pre[s % FACT_RDEPTH] until they are parked
    u32x4_t wq[FACT_WDEPTH + 1][2][2][2];  // [ring slot][half of the stage][k-step][n-tile of this wave]
    const u32x4_t* wlane = a.wfrag + (size_t)(2 * wave) * WAVE + lane;
    // weight block of patch index j (0..24; 25 = the metadata block) for this tile's own cell
    auto block_of = [&](int j) {
        if (j >= 25) return P2 + 1;
        const int dy = j / 5, dx = j - 5 * dy;
        const int oy = sy - 4 + dy, ox = sx - 4 + dx;
        return ((unsigned)oy < (unsigned)G2 && (unsigned)ox < (unsigned)G2) ? oy * G2 + ox : P2;
    };

    f32x16_t acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][m][r] = 0.0f;

#define FACT_FETCH(S)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                 \
        pre[(S) % FACT_RDEPTH][i] = (FACT_PATCH_ABLATE & 1) ? (u32x4_t){(uint32_t)tid, 1u, 2u, (uint32_t)(S)}    \
                                                            : *(const u32x4_t*)(my_rows + (size_t)(32 * i) * KRB + (S) * 128)
#define FACT_PARK(S)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; i++) *(u32x4_t*)(my_park + ((S) & 1) * STAGEB + 32 * i * ROWB) = pre[(S) % FACT_RDEPTH][i]
#define FACT_WFETCH(S)                                                                                            \
    _Pragma("unroll") for (int h = 0; h < 2; h++) {                                                               \
        const int blk = block_of(2 * (S) + h);                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 2; nt++)                                                          \
            wq[(S) % (FACT_WDEPTH + 1)][h][ks][nt] = (FACT_PATCH_ABLATE & 2) ? (u32x4_t){(uint32_t)blk, 3u, (uint32_t)lane, 0x3F803F80u} \
                                                         : wlane[(size_t)((blk * 2 + ks) * 8 + nt) * WAVE];       \
    }

#pragma unroll
    for (int s = 0; s < FACT_RDEPTH; s++) { FACT_FETCH(s); }
#pragma unroll
    for (int s = 0; s < FACT_WDEPTH; s++) { FACT_WFETCH(s); }
    FACT_PARK(0);
    if (!(FACT_PATCH_ABLATE & 8)) __syncthreads();
#pragma unroll
    for (int s = 0; s < FACT_NST; s++) {
        if (s + FACT_WDEPTH < FACT_NST) { FACT_WFETCH(s + FACT_WDEPTH); }
        if (s + FACT_RDEPTH < FACT_NST) { FACT_FETCH(s + FACT_RDEPTH); }  // into the ring slot stage s left when it was parked
        const uint8_t* buf = stage + (s & 1) * STAGEB;
#pragma unroll
        for (int u = 0; u < 4; u++) {  // k-step u of the stage: k = 16 u .. 16 u + 15 (half u >> 1, k-step u & 1 of its weight block)
            u32x4_t b[4];
#pragma unroll
            for (int m = 0; m < 4; m++) b[m] = *(const u32x4_t*)(buf + (32 * m + n32) * ROWB + (16 * u + 8 * hh) * 2);
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int m = 0; m < 4; m++)
                    acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wq[s % (FACT_WDEPTH + 1)][u >> 1][u & 1][nt]), as_bf16x8(b[m]),
                                                                         acc[nt][m], 0, 0, 0);
        }
        if (s + 1 < FACT_NST) {
            FACT_PARK(s + 1);
            if (!(FACT_PATCH_ABLATE & 8)) __syncthreads();
        }
    }
#undef FACT_FETCH
#undef FACT_PARK
#undef FACT_WFETCH

    // ---- y1[row][n] = bf16(D[n][slot] + yview[env][n] + bias[n]).  A lane holds 4 consecutive n of ONE slot per register group:
    // straight from there the yview reads and y1 writes would be 16- / 8-byte pieces of 32 different rows per instruction (0.19 of the
    // launch's 0.47 ms).  The accumulators go through LDS instead, one 32-slot tile at a time ([32 slots][256 n] float32 over the
    // stage buffers), and leave row by row: one wave instruction reads one whole yview row and writes one whole y1 row.
    constexpr int EPB = 1024 + 16;  // bytes of a slot's row in the transposition buffer (+16: eight lanes' pieces cover all banks once)
    static_assert(32 * EPB <= 2 * STAGEB, "the transposition buffer fits over the stage buffers");
    // Wave w writes the rows of slots 32 m + w + 4 i (m = 0..3, i = 0..7).  Their row numbers: ONE load per lane, read back by
    // v_readlane; all 32 yview rows of the wave are requested here, in one burst — the rings of the K loop are dead, their registers
    // hold the rows until the transposition passes reach them (a first version fetched row number, then yview row, pass by pass: 32
    // dependent HBM round trips per tile with nothing to overlap them, 0.59 ms for the launch pair instead of 0.47).
    const int my_row = a.row_of_slot[tile * FACT_MT + 32 * ((lane >> 3) & 3) + wave + 4 * (lane & 7)];
    const f32x4_t bs = *(const f32x4_t*)(a.bias + 4 * lane);  // this lane always handles n = 4 lane .. 4 lane + 3
    int rows_w[4][8];
    f32x4_t yv[4][8];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            rows_w[m][i] = __builtin_amdgcn_readlane(my_row, 8 * m + i);
            const int env = max(rows_w[m][i], 0) % a.E;  // (a padding slot reads env 0's row and stores nothing)
            yv[m][i] = *(const f32x4_t*)(a.yview + (size_t)env * 256 + 4 * lane);
        }
    __syncthreads();  // every wave is done with the stage buffers
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n0 = 64 * wave + 32 * nt + 8 * q + 4 * hh;
                *(f32x4_t*)(stage + n32 * EPB + n0 * 4) = (f32x4_t){acc[nt][m][4 * q], acc[nt][m][4 * q + 1], acc[nt][m][4 * q + 2], acc[nt][m][4 * q + 3]};
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int row = rows_w[m][i];  // uniform over the wave
            const f32x4_t d = *(const f32x4_t*)(stage + (wave + 4 * i) * EPB + lane * 16);
            u32x2_t o;
            o[0] = pack_bf16(d[0] + yv[m][i][0] + bs[0], d[1] + yv[m][i][1] + bs[1]);
            o[1] = pack_bf16(d[2] + yv[m][i][2] + bs[2], d[3] + yv[m][i][3] + bs[3]);
            if (HEAD) {  // tanh of the bf16-rounded pre-activation (what k_policy_head's stage a does to the stored y1), into the tail's image
                u32x2_t x;
                x[0] = tanh2_pack(__uint_as_float(o[0] << 16), __uint_as_float(o[0] & 0xFFFF0000u));
                x[1] = tanh2_pack(__uint_as_float(o[1] << 16), __uint_as_float(o[1] & 0xFFFF0000u));
                if (row < 0) x = (u32x2_t){0u, 0u};
                *(u32x2_t*)(xs + (32 * m + wave + 4 * i) * HEAD_XS_ROW + lane * 8) = x;
            } else if (row >= 0 && !((FACT_PATCH_ABLATE & 4) && acc[0][m][0] != 12345.0f)) {
                *(u32x2_t*)(a.y1 + (size_t)row * 256 + 4 * lane) = o;
            }
        }
        if (m < 3) __syncthreads();
    }
    if (HEAD) {
        const HeadArgs& ha = a.head;
        u32x4_t w2[16], wh[4];
#pragma unroll
        for (int s = 0; s < 16; s++) w2[s] = ha.fc2_frag[(wave * 16 + s) * WAVE + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) wh[s] = ha.head_frag[s * WAVE + lane];
        f32x16_t bias2;
#pragma unroll
        for (int r = 0; r < 16; r++) bias2[r] = ha.fc2_bias[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh];
        f32x4_t biash;
#pragma unroll
        for (int r = 0; r < 4; r++) biash[r] = ha.head_bias[4 * (lane >> 4) + r];
        __syncthreads();  // xs is complete
        const int32_t* slot_rows = a.row_of_slot + tile * FACT_MT;
        head_stages_bcde(ha, xs, xs, w2, wh, bias2, biash, wave, lane, [&](int r) { return (int64_t)slot_rows[r]; });
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int fact_geometry(int grid_size, int meta_len, int* kv, int* kr) {
    if (grid_size != 15 && grid_size != 11) return ctf_policy_fail("the factored fc1 path is built for grid_size 11 and 15");
    if (meta_len < 2 || (meta_len & 1) || meta_len > 32) return ctf_policy_fail("meta_len must be even and <= 32");
    const int pp = (((grid_size - 4) * (grid_size - 4) + 31) >> 5) << 5;
    *kv = 32 * pp;
    *kr = (800 + meta_len + 63) & ~63;
    if (*kr != FACT_NST * 64) return ctf_policy_fail("internal: patch row length");
    return 0;
}

extern "C" int32_t ctf_policy_fact_view_stride(int32_t grid_size) {
    const int pp = (((grid_size - 4) * (grid_size - 4) + 31) >> 5) << 5;
    return 32 * pp;
}
extern "C" int32_t ctf_policy_fact_row_stride(int32_t meta_len) { return (800 + meta_len + 63) & ~63; }
extern "C" int32_t ctf_policy_fact_max_tiles(int32_t n_envs, int32_t n_sel, int32_t grid_size) {
    return (int32_t)(((int64_t)n_envs * n_sel + FACT_MT - 1) / FACT_MT + grid_size * grid_size);
}

static int pack_sel(const int32_t* agent_sel, int n_sel, int n_agents, uint64_t* out) {
    if (!agent_sel || n_sel < 1 || n_sel > 4) return ctf_policy_fail("the factored path takes 1..4 selected agents");
    uint64_t p = 0;
    for (int k = 0; k < n_sel; k++) {
        if (agent_sel[k] < 0 || agent_sel[k] >= n_agents) return ctf_policy_fail("agent_sel entry out of range");
        p |= (uint64_t)agent_sel[k] << (4 * k);
    }
    *out = p;
    return 0;
}

extern "C" int ctf_policy_fact_bucket(const uint16_t* selfcell_dev, int32_t n_envs, int32_t n_agents, int32_t grid_size,
                                      const int32_t* agent_sel, int32_t n_sel, int32_t* work_dev, int32_t* slot_of_dev,
                                      int32_t* row_of_slot_dev, int32_t device_id, void* stream) {
    if (!selfcell_dev || !work_dev || !slot_of_dev || !row_of_slot_dev) return ctf_policy_fail("null argument");
    if (grid_size * grid_size > FACT_BINS) return ctf_policy_fail("grid_size * grid_size must be <= 256");
    if (n_envs < 1 || n_agents < 1 || n_agents > 16 || (int64_t)n_envs * n_sel > 0x7FFFFFF0) return ctf_policy_fail("n_envs / n_agents out of range");
    BucketArgs a;
    if (pack_sel(agent_sel, n_sel, n_agents, &a.sel_pack)) return -1;
    a.selfcells = selfcell_dev;
    a.hist = work_dev;
    a.cursor = work_dev + FACT_BINS;
    a.n_tiles = work_dev + 2 * FACT_BINS;
    a.tile_cell = work_dev + 2 * FACT_BINS + 64;
    a.slot_of = slot_of_dev;
    a.row_of_slot = row_of_slot_dev;
    a.E = n_envs; a.N = n_agents; a.A = n_sel; a.GG = grid_size * grid_size;
    a.t_max = ctf_policy_fact_max_tiles(n_envs, n_sel, grid_size);
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const int total = n_envs * n_sel;
    const int blocks = (total + 256 * FACT_ITEMS_PER_THREAD - 1) / (256 * FACT_ITEMS_PER_THREAD);
    hipLaunchKernelGGL(k_fact_hist, dim3(blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_fact_scan, dim3(1), dim3(FACT_BINS), 0, st, a);
    hipLaunchKernelGGL(k_fact_assign, dim3(blocks), dim3(256), 0, st, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}

extern "C" int ctf_policy_features_fact(const uint8_t* codes_dev, const uint16_t* meta_dev, const uint16_t* selfcell_dev, int32_t n_envs,
                                        int32_t n_agents, int32_t grid_size, int32_t meta_len, const int32_t* agent_sel, int32_t n_sel,
                                        const void* conv1_frag_dev, const float* conv1_bias_dev, const void* conv2_frag_dev,
                                        const float* conv2_bias_dev, const int32_t* slot_of_dev, uint16_t* view_dev, uint16_t* prow_dev,
                                        int32_t device_id, void* stream) {
    if (!codes_dev || !meta_dev || !selfcell_dev || !conv1_frag_dev || !conv1_bias_dev || !conv2_frag_dev || !conv2_bias_dev || !slot_of_dev ||
        !view_dev || !prow_dev)
        return ctf_policy_fail("null argument");
    FactFrontArgs fa;
    if (fact_geometry(grid_size, meta_len, &fa.KV, &fa.KR)) return -1;
    if (n_envs < 1 || n_agents < 1 || n_agents > 16) return ctf_policy_fail("n_envs / n_agents out of range");
    if (((uintptr_t)view_dev & 15) || ((uintptr_t)prow_dev & 15) || ((uintptr_t)meta_dev & 3)) return ctf_policy_fail("view / prow must be 16-byte, meta 4-byte aligned");
    PolicyArgs& a = fa.p;
    if (pack_sel(agent_sel, n_sel, n_agents, &a.sel_pack)) return -1;
    a.codes = codes_dev; a.meta = meta_dev; a.act = nullptr;
    a.w1frag = (const u32x4_t*)conv1_frag_dev; a.b1 = conv1_bias_dev;
    a.w2frag = (const u32x4_t*)conv2_frag_dev; a.b2 = conv2_bias_dev;
    a.n_envs = n_envs; a.N = n_agents; a.G = grid_size; a.M = meta_len; a.n_sel = n_sel; a.Kp = 0;
    a.h0_out = nullptr; a.h1_out = nullptr;
    const int G1 = grid_size - 2, G2 = grid_size - 4;
    a.inv_g1 = (65536 + G1 - 1) / G1;
    a.inv_g2 = (65536 + G2 - 1) / G2;
    for (int p = 0; p < G1 * G1; p++)
        if ((int)(((uint32_t)p * a.inv_g1) >> 16) != p / G1) return ctf_policy_fail("internal: reciprocal of G-2 not exact");
    for (int p = 0; p < G2 * G2; p++)
        if ((int)(((uint32_t)p * a.inv_g2) >> 16) != p / G2) return ctf_policy_fail("internal: reciprocal of G-4 not exact");
    fa.selfcells = selfcell_dev; fa.slot_of = slot_of_dev; fa.view = view_dev; fa.prow = prow_dev; fa.A = n_sel;
    const int n_cus = ctf_policy_cus(device_id);
    if (!n_cus) return ctf_policy_fail("hipGetDeviceProperties failed");
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    const int pp = fa.KV / 32;
    const int h1hp = pol_h1_bytes(grid_size) + 4 * 9 * 32;
    const int pstb = 25 * 72 + 64;  // k_policy_features_fact's PSTB / H2R
    const int h0h2 = pol_h0_bytes(grid_size) > pp * 72 ? pol_h0_bytes(grid_size) : pp * 72;  // k_policy_features_fact's H0H2
    const int per_wave = h0h2 + (h1hp > 4 * pstb ? h1hp : 4 * pstb);
    int wpb = 4;  // two blocks of four waves per CU = two waves per SIMD (16.7 KB of LDS per wave since h2s lies over h0)
    if (const char* ov = getenv("CTF_POLICY_FACT_WPB")) {  // profiling only
        const int v = atoi(ov);
        if (v >= 1 && v <= 4) wpb = v;
    }
    const size_t sh = (size_t)wpb * per_wave;
    int per_cu = (int)((160 * 1024) / sh);
    if (per_cu < 1) per_cu = 1;
    if (const char* ov = getenv("CTF_POLICY_FACT_BLOCKS_PER_CU")) {  // profiling only
        const int v = atoi(ov);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    int blocks = (n_envs + wpb - 1) / wpb;
    if (blocks > n_cus * per_cu) blocks = n_cus * per_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    if (grid_size == 15) {
        err = hipFuncSetAttribute((const void*)k_policy_features_fact<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features_fact<15>, dim3(blocks), dim3(wpb * WAVE), sh, st, fa);
    } else {
        err = hipFuncSetAttribute((const void*)k_policy_features_fact<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features_fact<11>, dim3(blocks), dim3(wpb * WAVE), sh, st, fa);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}

extern "C" int ctf_policy_fc1_patch(const uint16_t* prow_dev, const int32_t* row_of_slot_dev, const int32_t* work_dev, const float* yview_dev,
                                    const void* patch_frag_dev, const float* fc1_bias_dev, int32_t n_envs, int32_t n_sel, int32_t grid_size,
                                    int32_t meta_len, uint16_t* y1_dev, int32_t device_id, void* stream) {
    if (!prow_dev || !row_of_slot_dev || !work_dev || !yview_dev || !patch_frag_dev || !fc1_bias_dev || !y1_dev) return ctf_policy_fail("null argument");
    int kv, kr;
    if (fact_geometry(grid_size, meta_len, &kv, &kr)) return -1;
    if (((uintptr_t)prow_dev & 15) || ((uintptr_t)yview_dev & 15) || ((uintptr_t)y1_dev & 7) || ((uintptr_t)patch_frag_dev & 15) || ((uintptr_t)fc1_bias_dev & 15))
        return ctf_policy_fail("prow / yview / fragments / bias must be 16-byte aligned");
    PatchArgs a;
    a.prow = prow_dev; a.row_of_slot = row_of_slot_dev;
    a.n_tiles = work_dev + 2 * FACT_BINS;
    a.tile_cell = work_dev + 2 * FACT_BINS + 64;
    a.yview = yview_dev; a.wfrag = (const u32x4_t*)patch_frag_dev; a.bias = fc1_bias_dev; a.y1 = y1_dev;
    a.E = n_envs; a.KR = kr;
    const int t_max = ctf_policy_fact_max_tiles(n_envs, n_sel, grid_size);
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const int sh = 2 * FACT_MT * (128 + FACT_ROW_PAD);
    if (grid_size == 15) hipLaunchKernelGGL((k_policy_fc1_patch<15, false>), dim3(t_max), dim3(256), sh, st, a);
    else hipLaunchKernelGGL((k_policy_fc1_patch<11, false>), dim3(t_max), dim3(256), sh, st, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}

extern "C" int ctf_policy_fc1_patch_head(const uint16_t* prow_dev, const int32_t* row_of_slot_dev, const int32_t* work_dev, const float* yview_dev,
                                         const void* patch_frag_dev, const float* fc1_bias_dev, int32_t n_envs, int32_t n_sel, int32_t grid_size,
                                         int32_t meta_len, const void* fc2_frag_dev, const float* fc2_bias_dev, const void* head_frag_dev,
                                         const float* head_bias_dev, const float* mask_decision_dev, const int32_t* given_action_dev,
                                         int32_t n_actions, uint64_t seed, uint64_t offset, int32_t* action_dev, float* logprob_dev,
                                         float* entropy_dev, float* value_dev, float* logits_dev, int32_t device_id, void* stream) {
    if (!prow_dev || !row_of_slot_dev || !work_dev || !yview_dev || !patch_frag_dev || !fc1_bias_dev || !fc2_frag_dev || !fc2_bias_dev || !head_frag_dev ||
        !head_bias_dev || !action_dev || !logprob_dev || !entropy_dev || !value_dev)
        return ctf_policy_fail("null argument");
    int kv, kr;
    if (fact_geometry(grid_size, meta_len, &kv, &kr)) return -1;
    if (n_actions < 1 || n_actions > 15) return ctf_policy_fail("n_actions out of range");
    if (((uintptr_t)prow_dev & 15) || ((uintptr_t)yview_dev & 15) || ((uintptr_t)patch_frag_dev & 15) || ((uintptr_t)fc1_bias_dev & 15) ||
        ((uintptr_t)fc2_frag_dev & 15) || ((uintptr_t)head_frag_dev & 15))
        return ctf_policy_fail("prow / yview / fragments / bias must be 16-byte aligned");
    PatchArgs a;
    a.prow = prow_dev; a.row_of_slot = row_of_slot_dev;
    a.n_tiles = work_dev + 2 * FACT_BINS;
    a.tile_cell = work_dev + 2 * FACT_BINS + 64;
    a.yview = yview_dev; a.wfrag = (const u32x4_t*)patch_frag_dev; a.bias = fc1_bias_dev; a.y1 = nullptr;
    a.E = n_envs; a.KR = kr;
    HeadArgs& h = a.head;
    h.y1 = nullptr; h.fc2_frag = (const u32x4_t*)fc2_frag_dev; h.fc2_bias = fc2_bias_dev; h.head_frag = (const u32x4_t*)head_frag_dev;
    h.head_bias = head_bias_dev; h.mask = mask_decision_dev; h.given = given_action_dev; h.action = action_dev; h.logprob = logprob_dev;
    h.entropy = entropy_dev; h.value = value_dev; h.logits = logits_dev; h.B = (int64_t)n_envs * n_sel; h.A = n_actions; h.seed = seed; h.offset = offset;
    const int t_max = ctf_policy_fact_max_tiles(n_envs, n_sel, grid_size);
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const int sh = 2 * FACT_MT * (128 + FACT_ROW_PAD) + HEAD_TILE * HEAD_XS_ROW;
    hipError_t err;
    if (grid_size == 15) {
        err = hipFuncSetAttribute((const void*)k_policy_fc1_patch<15, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_fc1_patch<15, true>), dim3(t_max), dim3(256), sh, st, a);
    } else {
        err = hipFuncSetAttribute((const void*)k_policy_fc1_patch<11, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_fc1_patch<11, true>), dim3(t_max), dim3(256), sh, st, a);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the view GEMM: yview[E][256] = view[E][KV] x W_flat^T, float32 out
// ------------------------------------------------------------------------------------------------
// OFF by default (policy_native.py: native_view_gemm): measured 0.205 - 0.21 ms a call against the library's 0.185 (both 0.175 under
// rocprofv3; results bit-identical), in every form tried — register staging at 4 and 8 waves, this LDS-DMA ring, a staggered walk over K,
// a tile-blocked view layout (profiles/r04_view_gemm.md).  The shape (M = 65 536, N = 256, K = 4 096: 137 GFLOP against 0.54 GB) is bound
// by what one CU's vector-memory path takes in: 2.1 MB of view rows from HBM (~24 GB/s a CU at the chip's ~6 TB/s) plus, for a 128-row
// tile, 4.2 MB of W from L2 (~70 GB/s a CU) — 0.15 ms, and the MFMAs (37 % busy) hide under it.  Kept as the tested statement of that.
//
// A block of 8 waves takes 128 rows and ALL 256 columns (W, 2 MB, stays in L2), K in chunks of 64; a chunk of both operands (16 + 32 KB)
// goes global -> LDS directly (global_load_lds_dwordx4: no registers, no ds_write) into a ring of THREE stages, two chunks ahead of its
// use, with one raw barrier a chunk and counted waits:
//     chunk c:   s_waitcnt vmcnt(6)   this wave's 6 pieces of chunk c have landed (the 6 of chunk c + 1 may still fly)
//                s_barrier            everybody's have, and everybody has finished reading chunk c - 1
//                6 x glds             chunk c + 2 -> the stage chunk c - 1 was read from
//                16 ds_read_b128 + 16 MFMA on chunk c
// An LDS-DMA instruction writes 1 KiB contiguously (lane L -> bytes 16 L ..), so the stage is unpadded [384 rows][128 B] and the bank
// spread comes from the SOURCE side: slot s of row r holds the row's 16-byte piece s ^ ((r >> 1) & 7); a ds_read_b128 lane group (16
// rows, one piece) then covers the 16 slots of a 256-byte bank row exactly once.  Wave w owns rows 64 (w >> 2) .. + 63 x outputs
// 64 (w & 3) .. + 63 (32x32x16 MFMA, weights as the A operand: a lane ends with 4 consecutive outputs of ONE row).
struct ViewGemmArgs {
    const uint16_t* a;    // bf16 [M][K]
    const uint16_t* w;    // bf16 [256][K]
    float* out;           // f32 [M][256]
    int32_t M, K;
};
#define VG_STAGE 49152   // bytes of a stage: 128 view rows then 256 weight rows of 64 k
__device__ __forceinline__ void vg_glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    // (hand-placed: the compiler neither sees the load nor waits for it — the loop's own s_waitcnt vmcnt(6) does; M0 is put back)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_view_gemm(ViewGemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t vg_lds[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int n32 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * 128;
    const int n_chunks = g.K >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)vg_lds;
    // piece i of this wave = staged rows 8 (6 wave + i) .. + 7: lane -> row + (lane >> 3), slot lane & 7
    const uint8_t* src[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int r = 8 * (6 * wave + i) + (lane >> 3);
        const int pc = (lane & 7) ^ ((r >> 1) & 7);
        src[i] = r < 128 ? (const uint8_t*)g.a + (size_t)min(row0 + r, g.M - 1) * g.K * 2 + pc * 16
                         : (const uint8_t*)g.w + (size_t)(r - 128) * g.K * 2 + pc * 16;
    }
    const uint32_t dst0 = lds0 + 6 * wave * 1024;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][m][r] = 0.0f;
    const int mh = wave >> 2, nq = wave & 3;
    // fragment reads: rows 64 mh + 32 m + n32 (view) / 128 + 64 nq + 32 i + n32 (weights), piece 2 u + hh -> slot (2 u + hh) ^ ((n32 >> 1) & 7)
    const int sw = (n32 >> 1) & 7;
    const uint8_t* const arow = vg_lds + (64 * mh + n32) * 128;
    const uint8_t* const wrow = vg_lds + (128 + 64 * nq + n32) * 128;
    int slot[4];
#pragma unroll
    for (int u = 0; u < 4; u++) slot[u] = ((2 * u + hh) ^ sw) * 16;
#define VG_ISSUE(C, ST)  _Pragma("unroll") for (int i = 0; i < 6; i++) vg_glds16(src[i] + (size_t)(C) * 128, dst0 + (ST) * VG_STAGE + i * 1024)
    VG_ISSUE(0, 0);
    VG_ISSUE(min(1, n_chunks - 1), 1);
    int st = 0, st2 = 2;   // stage of chunk c / of chunk c + 2
#pragma unroll 1
    for (int c = 0; c < n_chunks; c++) {
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        VG_ISSUE(min(c + 2, n_chunks - 1), st2);   // (past the end: the last chunk once more, into a stage nobody reads any longer)
        const uint8_t* A = arow + st * VG_STAGE;
        const uint8_t* B = wrow + st * VG_STAGE;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            u32x4_t wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; i++) wf[i] = *(const u32x4_t*)(B + i * 4096 + slot[u]);
#pragma unroll
            for (int m = 0; m < 2; m++) af[m] = *(const u32x4_t*)(A + m * 4096 + slot[u]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int m = 0; m < 2; m++)
                    acc[i][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[i]), as_bf16x8(af[m]), acc[i][m], 0, 0, 0);
        }
        st = st == 2 ? 0 : st + 1;
        st2 = st2 == 2 ? 0 : st2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two refetches past the end: landed before the wave may go
#undef VG_ISSUE
    // D[n][row]: a lane holds column `row` (64 mh + 32 m + n32) and, per register group q, outputs n0 = 64 nq + 32 i + 8 q + 4 hh .. + 3
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int row = row0 + 64 * mh + 32 * m + n32;
        if (row < g.M) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n0 = 64 * nq + 32 * i + 8 * q + 4 * hh;
                    *(f32x4_t*)(g.out + (size_t)row * 256 + n0) = (f32x4_t){acc[i][m][4 * q], acc[i][m][4 * q + 1], acc[i][m][4 * q + 2], acc[i][m][4 * q + 3]};
                }
        }
    }
}

extern "C" int ctf_policy_view_gemm(const uint16_t* view_dev, const uint16_t* w_rows_dev, int32_t n_rows, int32_t kv, float* yview_dev,
                                    int32_t device_id, void* stream) {
    if (!view_dev || !w_rows_dev || !yview_dev) return ctf_policy_fail("null argument");
    if (n_rows < 1 || kv < 64 || (kv & 63)) return ctf_policy_fail("n_rows >= 1 and kv a multiple of 64");
    if (((uintptr_t)view_dev | (uintptr_t)w_rows_dev | (uintptr_t)yview_dev) & 15) return ctf_policy_fail("16-byte alignment");
    ViewGemmArgs g;
    g.a = view_dev; g.w = w_rows_dev; g.out = yview_dev; g.M = n_rows; g.K = kv;
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    const int sh = 3 * VG_STAGE;
    hipError_t err = hipFuncSetAttribute((const void*)k_view_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
    if (err == hipSuccess) hipLaunchKernelGGL(k_view_gemm, dim3((n_rows + 127) / 128), dim3(512), sh, (hipStream_t)stream, g);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}
