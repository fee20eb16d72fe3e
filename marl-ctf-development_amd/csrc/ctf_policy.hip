// ctf_policy.hip — the reference's policy / value network (agent_network.py:5-81) for hundreds of thousands of agents per
// launch, around the library's fc1 GEMM:
//   k_policy_features        conv3x3(C->16) tanh, conv3x3(16->32) tanh, flatten ++ metadata (agent_network.py:13-14,30-36),
//                            one wave per agent, fed by the env's compact observation (ctf_observe_codes) instead of the
//                            14x larger one-hot planes
//   k_policy_features_team   the same for agents that share a view: the convolutions once per env, a patch per agent
//   k_policy_head            tanh, fc2, tanh, heads, mask, sample, log-prob, entropy (agent_network.py:37-40,63-81)
//
// k_policy_features — one wave per sample, everything between the code bytes and the activation row stays on the CU:
//   h0  LDS bf16 [2 halves][G*G cells][8 ch]  the one-hot input, written straight from the codes (channel halves in separate
//           arrays: a lane's 16-byte operand reads then fall on consecutive addresses across lanes — no bank conflicts)
//   conv1 = 16x16x32 MFMAs: D[out ch][position] over K = (2 taps) x (16 in ch); A = weights, register-resident for the
//           whole launch; B = ds_read_b128 of h0 rows (a lane's 8 consecutive channels of one cell)
//   h1  LDS bf16 [2 halves][G1*G1 positions][8 ch]  tanh(conv1), written 8 bytes per lane from the accumulator layout
//   conv2 = 32x32x16 MFMAs: D[out ch][position], one MFMA per tap (K = 16 in ch), B = ds_read_b128 of h1 rows
//   out HBM bf16 [sample][Kp]            tanh(conv2) as 8-byte stores in the order the accumulators hold it:
//           column ((c/4) * PP + p) * 4 + c%4 for out channel c, position p (PP = positions rounded up to whole
//           32-position tiles, so that every store instruction covers whole 128-byte lines) — the fc1 weight's columns
//           are permuted to this order once on the host (policy_native.py), so no transpose happens anywhere; then the M
//           metadata values (f16 -> bf16) and padding up to Kp (a multiple of 64: rows are whole lines).
// tanh(x) = 1 - 2 / (2^(x * 2 log2 e) + 1): the factor 2 log2 e is folded into the conv weights and biases on the host,
// so a pair of activations costs 2 v_exp_f32, v_pk_add_f32, 2 v_rcp_f32, v_pk_fma_f32, v_cvt_pk_bf16_f32.
// fc1 is a plain GEMM and stays with hipBLASLt (through torch).
#include "ctf_policy_dev.h"

// One LDS image of this wave ([2 channel halves][rows][8 channels] bf16) -> global memory channels-last ([rows][16 channels]): 16-byte
// pieces, consecutive lanes write consecutive pieces.
__device__ __forceinline__ void pol_store_image(const uint8_t* img, int half_bytes, int rows, uint16_t* out, int lane) {
    // piece idx = lane + 64 k: half idx & 1 = lane & 1, row idx >> 1 = (lane >> 1) + 32 k.  Rolled: unrolled, the fourteen address
    // pairs spill, and a scratch reload is a vmcnt(0) — every store of the sample drained before the next one may issue.
    const uint8_t* src = img + (lane & 1) * half_bytes + (lane >> 1) * 16;
    uint8_t* dst = (uint8_t*)out + lane * 16;
#pragma unroll 1
    for (int idx = lane; idx < 2 * rows; idx += WAVE, src += 32 * 16, dst += WAVE * 16) *(u32x4_t*)dst = *(const u32x4_t*)src;
}

template <int TG, bool TRAIN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_policy_features(PolicyArgs a) {
    extern __shared__ uint32_t lds[];
    const int G = TG ? TG : a.G;
    const int G1 = G - 2, G2 = G - 4, GG = G * G, P1 = G1 * G1, P2 = G2 * G2;
    const int PP = ((P2 + 31) >> 5) << 5;  // positions per channel group in the activation row: whole 32-position tiles
    // FAST (G*G <= 256, a compile-time G): h0 persists across the wave's samples and only the cells whose code changed
    // are rewritten; the next sample's codes and metadata are loaded one sample ahead.
    constexpr bool FAST = TG != 0 && TG * TG <= 256;
    constexpr int NP = FAST ? (TG * TG + WAVE - 1) / WAVE : 1;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    uint8_t* h0 = (uint8_t*)lds + wave * (pol_h0_bytes(G) + pol_h1_bytes(G));
    uint8_t* h1 = h0 + pol_h0_bytes(G);
    const int H0A = GG * 16, H1A = pol_h1_bytes(G) / 2;  // bytes of one channel-half array

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
        off1[s] = ((tap / 3) * G + (tap % 3)) * 16 + (g1 & 1) * H0A;
    }
    // position 16 t + n walks the G1 x G1 output row-major: per tile it advances 16 = dy1 rows + dx1 columns
    const int y1_0 = (int)(((uint32_t)n1 * a.inv_g1) >> 16), x1_0 = n1 - y1_0 * G1;
    const int dy1 = 16 / G1, dx1 = 16 - dy1 * G1;
    // conv1 output: this lane's 4 channels 4 g .. 4 g + 3 of position n, as 8 bytes of half g >> 1
    uint8_t* h1w = h1 + (g1 >> 1) * H1A + n1 * 16 + (g1 & 1) * 8;
    // conv2: lane = (position n = lane & 31, channel half h = lane >> 5)
    const int n2 = lane & 31, hh = lane >> 5;

    const int S = a.n_sel * a.n_envs;
    const int s_first = blockIdx.x * wpb + wave, s_stride = gridDim.x * wpb;
    const int npair = (a.Kp - 32 * PP) >> 1;

    PolCodes<NP> oldc, nextc;
#pragma unroll
    for (int q = 0; q < NP; q++) oldc.b[q] = nextc.b[q] = 0;
    uint32_t nextm = 0;
    if (FAST) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int q = lane; q < GG * 2; q += WAVE) ((u32x4_t*)h0)[q] = z;  // both halves
        if (s_first < S) {
            const int k = s_first / a.n_envs, e = s_first - k * a.n_envs;
            const size_t row = (size_t)e * a.N + (int)((a.sel_pack >> (4 * k)) & 15u);
            nextc = pol_load_codes<NP>(a.codes + row * GG, lane, GG);
            if (lane < (a.M >> 1)) nextm = ((const uint32_t*)(a.meta + row * a.M))[lane];
        }
    }

    for (int s = s_first; s < S; s += s_stride) {
        uint16_t* arow = a.act + (size_t)((POL_ABLATE & 16) ? s_first : s) * a.Kp;
        uint32_t two = 0;
        if (FAST) {
            const PolCodes<NP> cur = nextc;
            two = nextm;
            {   // next sample's inputs (the last sample re-reads itself: every lane always issues the same loads)
                const int sn = min(s + s_stride, S - 1);
                const int k = sn / a.n_envs, e = sn - k * a.n_envs;
                const size_t row = (size_t)e * a.N + (int)((a.sel_pack >> (4 * k)) & 15u);
                if (POL_ABLATE == 0) {
                    nextc = pol_async_codes<NP>(a.codes + row * GG, lane, GG);
                    nextm = pol_async_dword((const uint32_t*)(a.meta + row * a.M) + min(lane, (a.M >> 1) - 1));
                } else {
                    nextc = pol_load_codes<NP>(a.codes + row * GG, lane, GG);
                    nextm = ((const uint32_t*)(a.meta + row * a.M))[min(lane, (a.M >> 1) - 1)];
                }
            }
            // h0 rows are one-hot: clear the previous sample's halfword, set this one's; slot 0 (own position) last
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const int c = lane + WAVE * q;
                const uint32_t o = oldc.b[q], n = cur.b[q];
                if (c < GG && o != n && !(POL_ABLATE & 4)) {  // static walls: most cells keep their code from sample to sample
                    uint8_t* cellp = h0 + c * 16;
                    const uint32_t oc = o & 0x7Fu, nc = n & 0x7Fu;
                    *(uint16_t*)(cellp + (oc >> 3) * H0A + (oc & 7u) * 2) = 0;
                    *(uint16_t*)(cellp + (nc >> 3) * H0A + (nc & 7u) * 2) = nc ? 0x3F80 : 0;
                    *(uint16_t*)cellp = (n >> 7) ? 0x3F80 : 0;
                }
            }
            oldc = cur;
        } else {
            const int k = s / a.n_envs, e = s - k * a.n_envs;
            const size_t row = (size_t)e * a.N + (int)((a.sel_pack >> (4 * k)) & 15u);
            const uint8_t* cp = a.codes + row * GG;
            if (lane < (a.M >> 1)) two = ((const uint32_t*)(a.meta + row * a.M))[lane];
            for (int c = lane; c < GG; c += WAVE) {  // one 32-byte one-hot row per cell
                const uint32_t code = cp[c];
                const uint32_t ch = code & 0x7Fu;
                uint32_t w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = ((ch >> 1) == (uint32_t)j && ch != 0) ? (0x3F80u << (16 * (ch & 1u))) : 0u;
                w[0] |= (code >> 7) ? 0x3F80u : 0u;
                *(u32x4_t*)(h0 + c * 16) = (u32x4_t){w[0], w[1], w[2], w[3]};
                *(u32x4_t*)(h0 + H0A + c * 16) = (u32x4_t){w[4], w[5], w[6], w[7]};
            }
        }
        // metadata: f16 -> bf16 pairs behind the conv features; then ONE column of 1.0 (column 32 PP + M: a caller may keep fc1's bias
        // in that column of its weight — the training path does, so that the bias gradient falls out of the weight-gradient GEMM; the
        // inference weights hold zero there); zero padding to Kp
        if (lane < npair) {
            uint32_t out = lane == (a.M >> 1) ? 0x3F80u : 0u;
            if (lane < (a.M >> 1)) {
                const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(two & 0xFFFFu));
                const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(two >> 16));
                out = pack_bf16(lo, hi);
            }
            ((uint32_t*)(arow + 32 * PP))[lane] = out;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (TRAIN && a.h0_out) pol_store_image(h0, H0A, GG, a.h0_out + (size_t)s * GG * 16, lane);

        // ---- conv1 + tanh -> h1.  (Positions >= P1 of the last tile read past h0 into h1 — inside this wave's LDS — and
        // land in h1 rows >= P1, which nothing reads.)
        const int T1 = (P1 + 15) >> 4;
        int x1 = x1_0, cell1 = y1_0 * G + x1_0;
        // two tiles per pass: two independent accumulation chains keep the MFMA pipe and the LDS busy within one wave
        int t = 0;
#pragma unroll 1
        for (; t + 1 < T1; t += 2) {
            const uint8_t* base_a = h0 + ((POL_ABLATE & 8) ? 0 : cell1 * 16);
            x1 += dx1;
            cell1 += dy1 * G + dx1;
            if (x1 >= G1) { x1 -= G1; cell1 += G - G1; }
            const uint8_t* base_b = h0 + ((POL_ABLATE & 8) ? 64 : cell1 * 16);
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
            *(u32x2_t*)(h1w + 16 * t * 16) = o;  // rows up to 16 * T1 exist
            o[0] = tanh2_pack(acc_b[0], acc_b[1]);
            o[1] = tanh2_pack(acc_b[2], acc_b[3]);
            *(u32x2_t*)(h1w + 16 * (t + 1) * 16) = o;
        }
        if (t < T1) {  // odd tile count: the last one alone
            const uint8_t* base_a = h0 + ((POL_ABLATE & 8) ? 0 : cell1 * 16);
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
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (TRAIN) pol_store_image(h1, H1A, P1, a.h1_out + (size_t)s * P1 * 16, lane);

        // ---- conv2 + tanh -> activation row
        const int T2 = (P2 + 31) >> 5;
#pragma unroll 1
        for (int t = 0; t < T2; t += 2) {
            const int pa = 32 * t + n2, pb = pa + 32;
            const int pca = min(pa, P2 - 1), pcb = min(pb, P2 - 1);
            const int ya = (int)(((uint32_t)pca * a.inv_g2) >> 16), yb = (int)(((uint32_t)pcb * a.inv_g2) >> 16);
            const uint8_t* base_a = h1 + ((POL_ABLATE & 8) ? 0 : (ya * G1 + (pca - ya * G2)) * 16 + hh * H1A);
            const uint8_t* base_b = h1 + ((POL_ABLATE & 8) ? 64 : (yb * G1 + (pcb - yb * G2)) * 16 + hh * H1A);
            f32x16_t acc_a = bias2, acc_b = bias2;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int off = ((tap / 3) * G1 + (tap % 3)) * 16;
                const u32x4_t ba = *(const u32x4_t*)(base_a + off);
                const u32x4_t bb = *(const u32x4_t*)(base_b + off);
                acc_a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(ba), acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(bb), acc_b, 0, 0, 0);
            }
            // one address per pass: the four channel groups of a lane sit (2 q - 3) * P2 * 4 elements around `mid`
            uint16_t* mid = arow + ((3 + hh) * PP + pa) * 4;
            // every lane stores, also the positions past P2 of the last tile (finite values under zero fc1 weights): whole lines
            if ((POL_ABLATE & 1) ? (acc_a[0] == 12345.0f && acc_b[5] == 1.0f) : true) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x2_t o;
                    o[0] = tanh2_pack(acc_a[4 * q], acc_a[4 * q + 1]);
                    o[1] = tanh2_pack(acc_a[4 * q + 2], acc_a[4 * q + 3]);
                    *(u32x2_t*)(mid + (2 * q - 3) * PP * 4) = o;
                }
            }
            if ((POL_ABLATE & 1) ? (acc_b[0] == 12345.0f && acc_a[7] == 1.0f) : (pb < PP)) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x2_t o;
                    o[0] = tanh2_pack(acc_b[4 * q], acc_b[4 * q + 1]);
                    o[1] = tanh2_pack(acc_b[4 * q + 2], acc_b[4 * q + 3]);
                    *(u32x2_t*)(mid + (2 * q - 3) * PP * 4 + 32 * 4) = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (FAST && POL_ABLATE == 0) {
            // The next sample's inputs, issued at the top of this sample, are older than its 1 + 4 T2 stores: a counted wait
            // leaves those stores in flight.  The wait sits HERE, in the iteration that issued the loads — the compiler knows
            // nothing about their latency and is free to copy the destination registers at the loop's back edge.
            // (TRAIN issues 14 more stores — the two images — between the loads and the last of these; waiting with the same count
            // is the conservative side: at most the youngest 1 + 4 T2 operations stay in flight, all of them stores.)
            constexpr int STORES = 1 + 4 * ((((TG - 4) * (TG - 4)) + 31) >> 5);
            pol_wait_codes<STORES, NP>(nextc);
            POL_WAIT_VM1(STORES, nextm);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the same front for agents that SHARE A VIEW
// ------------------------------------------------------------------------------------------------
// Teammates looking through the same reversal see identical tile planes (the relabelling is by team,
// gridworld_ctf.py:981-988); only plane 0, the own-position bit, differs.  One such bit reaches 3 x 3 conv1 outputs and
// 5 x 5 conv2 outputs, so a wave takes one env and ALL `A` selected agents of the group:
//   shared   conv1 -> tanh -> h1, conv2 -> tanh once, without any own-position bit; every agent's activation row gets the
//            shared values outside its 5 x 5 patch;
//   per agent  the own-position bit is switched on in h0, ONE 16-position tile recomputes the 3 x 3 conv1 patch into h1
//            (the shared values it displaces are kept in registers), ONE 32-position tile recomputes the 5 x 5 conv2 patch and
//            stores it; h1 and h0 are restored.  A lane's operand address is arbitrary, so a patch is just another tile.
// Per agent that is 14 MFMAs and 40 transcendental instructions instead of 96 and 216.  The arithmetic of every output
// is the same sum in the same order as in k_policy_features, so the two kernels agree bit for bit.
// Profiling-only phase trace of the team kernel (tools/trace_team.py builds with -DPOL_TRACE=1): wave 0 of block 0 records
// s_memtime at the phase boundaries of its first envs.
#ifndef POL_TRACE
#define POL_TRACE 0
#endif
#if POL_TRACE
__device__ uint64_t g_pol_trace[16 * 64];
extern "C" int ctf_policy_trace_read(uint64_t* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pol_trace), sizeof(uint64_t) * 16 * 64) == hipSuccess ? 0 : -1;
}
#define POL_STAMP(slot)                                                                         \
    do {                                                                                        \
        if (blockIdx.x == 0 && wave == 0 && trace_env < 64) {                                   \
            const uint64_t tstamp = __builtin_amdgcn_s_memtime();                               \
            if (lane == 0) g_pol_trace[trace_env * 16 + (slot)] = tstamp;                       \
        }                                                                                       \
    } while (0)
#else
#define POL_STAMP(slot) do { } while (0)
#endif

struct TeamArgs {
    PolicyArgs p;
    const uint16_t* selfcells;  // u16 [E][N]: the cell of every agent's own-position bit (ctf_observe_codes)
    int32_t A;                  // agents in the group (<= 4): p.sel_pack nibbles 0..A-1, all sharing planes 1..C-1
};

template <int TG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_policy_features_team(TeamArgs ta) {
    extern __shared__ uint32_t lds[];
    const PolicyArgs& a = ta.p;
    constexpr int G = TG, G1 = G - 2, G2 = G - 4, GG = G * G, P1 = G1 * G1, P2 = G2 * G2, PP = ((P2 + 31) >> 5) << 5;
    constexpr int NP = (GG + WAVE - 1) / WAVE;
    static_assert(GG <= 256, "one dword of code bytes per lane");
    const int A = ta.A;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    uint8_t* h0 = (uint8_t*)lds + wave * (pol_h0_bytes(G) + pol_h1_bytes(G));
    uint8_t* h1 = h0 + pol_h0_bytes(G);
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
    int off1[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const int tap = min(2 * s + (g1 >> 1), 8);
        off1[s] = ((tap / 3) * G + (tap % 3)) * 16 + (g1 & 1) * H0A;
    }
    const int y1_0 = (int)(((uint32_t)n1 * a.inv_g1) >> 16), x1_0 = n1 - y1_0 * G1;
    constexpr int dy1 = 16 / G1, dx1 = 16 - dy1 * G1;
    uint8_t* h1w = h1 + (g1 >> 1) * H1A + n1 * 16 + (g1 & 1) * 8;
    const int n2 = lane & 31, hh = lane >> 5;
    // patch tiles: conv1 lane n -> offset (n / 3, n % 3) of the 3 x 3 patch (n < 9), conv2 lane n -> (n / 5, n % 5) (n < 25)
    const int j1 = min(n1, 8), pdy1 = j1 / 3, pdx1 = j1 - 3 * pdy1;
    const int j2 = min(n2, 24), pdy2 = j2 / 5, pdx2 = j2 - 5 * pdy2;
    const int h1w_half = (g1 >> 1) * H1A + (g1 & 1) * 8;

    const int npair = (a.Kp - 32 * PP) >> 1;
    const int e_first = blockIdx.x * wpb + wave, e_stride = gridDim.x * wpb;
    const int ag0 = (int)(a.sel_pack & 15u);
    const size_t agent_stride = ((POL_ABLATE & 64) ? (size_t)1 : (size_t)a.n_envs) * a.Kp * 2;  // bytes from agent k's rows to agent k + 1's

    PolCodes<NP> oldc, nextc;
#pragma unroll
    for (int q = 0; q < NP; q++) oldc.b[q] = nextc.b[q] = 0;
    {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int q = lane; q < GG * 2; q += WAVE) ((u32x4_t*)h0)[q] = z;
        if (e_first < a.n_envs) nextc = pol_load_codes<NP>(a.codes + ((size_t)e_first * a.N + ag0) * GG, lane, GG);
    }

    int trace_env = 0;
    (void)trace_env;
    for (int e = e_first; e < a.n_envs; e += e_stride, trace_env++) {
        POL_STAMP(0);
        // nextc was issued at the top of the previous env and is older than that env's drain (below): it has arrived
        const PolCodes<NP> cur = nextc;
        nextc = pol_async_codes<NP>(a.codes + ((size_t)min(e + e_stride, a.n_envs - 1) * a.N + ag0) * GG, lane, GG);
        // every agent's own cell and metadata: needed only after the shared pass (waited for by the drain there)
        uint32_t scw[4], metaw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t row = (size_t)e * a.N + (int)((a.sel_pack >> (4 * min(k, A - 1))) & 15u);  // k >= A: agent A - 1 again
            scw[k] = pol_async_ushort(ta.selfcells + row);
            metaw[k] = pol_async_dword((const uint32_t*)(a.meta + row * a.M) + min(lane, (a.M >> 1) - 1));
        }
        // ---- h0 <- the shared planes (own-position bits stripped)
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int c = lane + WAVE * q;
            const uint32_t o = oldc.b[q] & 0x7Fu, n = cur.b[q] & 0x7Fu;
            if (c < GG && o != n) {
                uint8_t* cellp = h0 + c * 16;
                *(uint16_t*)(cellp + (o >> 3) * H0A + (o & 7u) * 2) = 0;
                if (n) *(uint16_t*)(cellp + (n >> 3) * H0A + (n & 7u) * 2) = 0x3F80;
            }
        }
        oldc = cur;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        POL_STAMP(1);

        // ---- shared conv1 + tanh -> h1, two tiles in flight (the launch runs one wave per SIMD — see the launcher — so
        // registers are plentiful and the wave's own instruction-level parallelism is all there is)
        constexpr int T1 = (P1 + 15) >> 4;
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
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        POL_STAMP(2);

        // ---- shared conv2 + tanh -> EVERY agent's row, all positions; the agent's 5 x 5 patch is overwritten below
        uint8_t* const act_env = (uint8_t*)a.act + ((POL_ABLATE & 64) ? (size_t)e * A : (size_t)e) * a.Kp * 2;  // agent 0's row of this env (uniform)
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
            // every lane stores (also the positions past P2 of the last tile: finite values under zero fc1 weights): whole lines
            const uint32_t lane_off = (uint32_t)(((3 + hh) * PP + pa) * 8);  // the four channel groups sit (2 q - 3) * PP * 8 around it
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32x2_t oa, ob;
                oa[0] = tanh2_pack(acc_a[4 * q], acc_a[4 * q + 1]);
                oa[1] = tanh2_pack(acc_a[4 * q + 2], acc_a[4 * q + 3]);
                ob[0] = tanh2_pack(acc_b[4 * q], acc_b[4 * q + 1]);
                ob[1] = tanh2_pack(acc_b[4 * q + 2], acc_b[4 * q + 3]);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < A && !((POL_ABLATE & 32) && (q & 1))) {
                        *(u32x2_t*)(act_env + k * agent_stride + lane_off + (2 * q - 3) * PP * 8) = oa;
                        *(u32x2_t*)(act_env + k * agent_stride + lane_off + (2 * q - 3) * PP * 8 + 256) = ob;
                    }
                }
            }
        }
        POL_STAMP(3);
        // the patch stores below hit addresses written above: have those writes acknowledged first.  The same drain is the
        // wait for this env's own cells / metadata and for the next env's codes, all issued before the stores.
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(scw[0]), "+v"(scw[1]), "+v"(scw[2]), "+v"(scw[3]), "+v"(metaw[0]), "+v"(metaw[1]), "+v"(metaw[2]),
                       "+v"(metaw[3])
                     :
                     : "memory");
        pol_wait_codes<0, NP>(nextc);  // every register exactly once (see pol_wait_codes)

        POL_STAMP(4);
        // ---- per agent: own-position bit on, the two patches, everything restored
#pragma unroll 1
        for (int k = 0; k < A; k++) {
            // (select chains, not indexing: the arrays live in registers)
            const int sc = __builtin_amdgcn_readfirstlane((int)(k == 0 ? scw[0] : k == 1 ? scw[1] : k == 2 ? scw[2] : scw[3]));
            const uint32_t mw = k == 0 ? metaw[0] : k == 1 ? metaw[1] : k == 2 ? metaw[2] : metaw[3];
            const int syk = sc / G, sxk = sc - syk * G;
            uint8_t* const arow = act_env + k * agent_stride;
            if (lane < npair) {  // metadata behind the conv features, then the column of 1.0 (see k_policy_features)
                uint32_t out = lane == (a.M >> 1) ? 0x3F80u : 0u;
                if (lane < (a.M >> 1)) {
                    const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(mw & 0xFFFFu));
                    const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(mw >> 16));
                    out = pack_bf16(lo, hi);
                }
                ((uint32_t*)(arow + 32 * PP * 2))[lane] = out;
            }
            uint16_t* selfp = (uint16_t*)(h0 + sc * 16);  // channel 0 of the own cell
            if (lane == 0) *selfp = 0x3F80;
            // conv1 patch: outputs (sy - 2 + dy, sx - 2 + dx), dy, dx in 0..2
            const int oy1 = syk - 2 + pdy1, ox1 = sxk - 2 + pdx1;
            const bool ok1 = n1 < 9 && (unsigned)oy1 < (unsigned)G1 && (unsigned)ox1 < (unsigned)G1;
            const int cy1 = min(max(oy1, 0), G1 - 1), cx1 = min(max(ox1, 0), G1 - 1);
            u32x2_t* h1p = (u32x2_t*)(h1 + h1w_half + (cy1 * G1 + cx1) * 16);
            const u32x2_t keep = *h1p;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            {
                const uint8_t* base = h0 + (cy1 * G + cx1) * 16;
                f32x4_t acc = bias1;
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const u32x4_t b = *(const u32x4_t*)(base + off1[q]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w1[q]), as_bf16x8(b), acc, 0, 0, 0);
                }
                u32x2_t o;
                o[0] = tanh2_pack(acc[0], acc[1]);
                o[1] = tanh2_pack(acc[2], acc[3]);
                if (ok1) *h1p = o;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // conv2 patch: outputs (sy - 4 + dy, sx - 4 + dx), dy, dx in 0..4
            {
                const int oy2 = syk - 4 + pdy2, ox2 = sxk - 4 + pdx2;
                const bool ok2 = n2 < 25 && (unsigned)oy2 < (unsigned)G2 && (unsigned)ox2 < (unsigned)G2;
                const int cy2 = min(max(oy2, 0), G2 - 1), cx2 = min(max(ox2, 0), G2 - 1);
                const uint8_t* base = h1 + (cy2 * G1 + cx2) * 16 + hh * H1A;
                f32x16_t acc = bias2;
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const u32x4_t b = *(const u32x4_t*)(base + ((tap / 3) * G1 + (tap % 3)) * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w2[tap]), as_bf16x8(b), acc, 0, 0, 0);
                }
                if (ok2) {
                    const uint32_t lane_off = (uint32_t)(((3 + hh) * PP + oy2 * G2 + ox2) * 8);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        u32x2_t o;
                        o[0] = tanh2_pack(acc[4 * q], acc[4 * q + 1]);
                        o[1] = tanh2_pack(acc[4 * q + 2], acc[4 * q + 3]);
                        *(u32x2_t*)(arow + lane_off + (2 * q - 3) * PP * 8) = o;
                    }
                }
            }
            // restore the shared image (the reads above are done: LDS serves a wave's operations in order)
            if (ok1) *h1p = keep;
            if (lane == 0) *selfp = 0;
            POL_STAMP(5 + k);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// the network's tail, fused: tanh(fc1 out) -> fc2 -> tanh -> action / value heads -> mask -> sample
// ------------------------------------------------------------------------------------------------
// Input is fc1's pre-activation (the BLAS GEMM's bf16 output, bias included, scaled by 2 log2 e like the conv stages).
// A block of 4 waves takes 128 samples:
//   a. every wave applies tanh to its 32 rows and parks them as bf16 in LDS xs[128][256] (rows padded by 16 bytes: the
//      32x32x16 B-operand reads of 32 consecutive samples then fall on 16 distinct 16-byte slots per lane group);
//   b. fc2 as out[128 ch][128 samples]: wave w owns channels 32 w .. 32 w + 31 (its 16 A fragments stay in registers for
//      the whole launch) and runs the 4 sample tiles: 64 MFMAs 32x32x16;
//   c. tanh, bf16, into LDS hs[128 samples][128 ch] (aliasing xs; rows padded by 16 bytes);
//   d. heads as out[16][16 samples] 16x16x32 MFMAs (rows 0..A-1 action logits, row A the value), wave w on its 32 samples;
//   e. a sample's 16 outputs sit in 4 lanes (lane, +16, +32, +48): masked log-softmax (logits + (mask - 1) * 1e9,
//      agent_network.py:66-75), entropy, inverse-CDF sampling with one Philox4x32-10 uniform per sample, log-prob.
__global__ void __launch_bounds__(256) k_policy_head(HeadArgs a) {
    extern __shared__ uint32_t lds[];
    uint8_t* xs = (uint8_t*)lds;  // stage a/b image; stage c/d image aliases it
    uint8_t* hs = (uint8_t*)lds;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int n32 = lane & 31, hh = lane >> 5, g4 = lane >> 4;

    u32x4_t w2[16], wh[4];
#pragma unroll
    for (int s = 0; s < 16; s++) w2[s] = a.fc2_frag[(wave * 16 + s) * WAVE + lane];
#pragma unroll
    for (int s = 0; s < 4; s++) wh[s] = a.head_frag[s * WAVE + lane];
    f32x16_t bias2;
#pragma unroll
    for (int r = 0; r < 16; r++) bias2[r] = a.fc2_bias[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh];
    f32x4_t biash;
#pragma unroll
    for (int r = 0; r < 4; r++) biash[r] = a.head_bias[4 * g4 + r];

    const int64_t n_tiles = (a.B + HEAD_TILE - 1) / HEAD_TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * HEAD_TILE;
        // ---- a. tanh(fc1) of this wave's 32 rows -> xs
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const int r = 32 * wave + 2 * i + hh;  // row within the tile; lane n32 takes 16-byte chunk n32 of it
            const int64_t srow = s0 + r;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (srow < a.B) v = *(const u32x4_t*)(a.y1 + srow * 256 + n32 * 8);
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = tanh2_pack(__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xFFFF0000u));
            *(u32x4_t*)(xs + r * HEAD_XS_ROW + n32 * 16) = o;
        }
        __syncthreads();
        // ---- b-e. fc2, tanh, heads, distribution (ctf_policy_dev.h)
        head_stages_bcde(a, xs, hs, w2, wh, bias2, biash, wave, lane, [&](int r) { const int64_t smp = s0 + r; return smp < a.B ? smp : (int64_t)-1; });
        __syncthreads();  // hs is free again
    }
}

static thread_local char g_perr[256];
extern "C" const char* ctf_policy_last_error(void) { return g_perr; }
int ctf_policy_fail(const char* msg) {
    snprintf(g_perr, sizeof(g_perr), "%s", msg);
    return -1;
}
static int pfail(const char* msg) { return ctf_policy_fail(msg); }

// ---- deterministic mode: the per-device workspace and the ordered reduction of the blocks' partial sums
static DetWorkspace g_det[64];
extern "C" int ctf_policy_set_deterministic(int32_t device_id, float* workspace_dev, int64_t workspace_floats) {
    if (device_id < 0 || device_id >= 64) return pfail("device_id out of range");
    if (workspace_dev && (workspace_floats < 1 || ((uintptr_t)workspace_dev & 15))) return pfail("workspace: 16-byte aligned, at least one float");
    g_det[device_id].ptr = workspace_dev;
    g_det[device_id].floats = workspace_dev ? workspace_floats : 0;
    return 0;
}
extern "C" int64_t ctf_policy_deterministic_workspace(int32_t device_id) {
    return (device_id >= 0 && device_id < 64 && g_det[device_id].ptr) ? g_det[device_id].floats : 0;
}
DetWorkspace ctf_policy_det(int device_id) {
    if (device_id < 0 || device_id >= 64) return DetWorkspace{nullptr, 0};
    return g_det[device_id];
}
__global__ void __launch_bounds__(256) k_det_reduce(const float* part, int n_blocks, int64_t stride, int elems, float* dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const float* p = part + i;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 3 < n_blocks; b += 4) {
        s0 += p[(size_t)b * stride];
        s1 += p[(size_t)(b + 1) * stride];
        s2 += p[(size_t)(b + 2) * stride];
        s3 += p[(size_t)(b + 3) * stride];
    }
    for (; b < n_blocks; b++) s0 += p[(size_t)b * stride];
    dst[i] += (s0 + s1) + (s2 + s3);
}
hipError_t ctf_policy_det_reduce(const float* part, int n_blocks, int64_t stride, int elems, float* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_det_reduce, dim3((elems + 255) / 256), dim3(256), 0, st, part, n_blocks, stride, elems, dst);
    return hipGetLastError();
}

// compute units of a device, looked up once per device and thread (hipGetDeviceProperties is not free)
static int policy_n_cus(int device_id);
int ctf_policy_cus(int device_id) { return policy_n_cus(device_id); }
static int policy_n_cus(int device_id) {
    static thread_local int cus_of[64];
    if (device_id >= 0 && device_id < 64 && cus_of[device_id]) return cus_of[device_id];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return 0;
    if (device_id >= 0 && device_id < 64) cus_of[device_id] = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}

extern "C" int32_t ctf_policy_act_stride(int32_t grid_size, int32_t meta_len) {
    const int pp = (((grid_size - 4) * (grid_size - 4) + 31) >> 5) << 5;  // positions padded to whole 32-position tiles
    return (32 * pp + meta_len + 63) & ~63;                               // rows are whole 128-byte lines
}

extern "C" int ctf_policy_features(const uint8_t* codes_dev, const uint16_t* meta_dev, int32_t n_envs, int32_t n_agents,
                                   int32_t grid_size, int32_t meta_len, const int32_t* agent_sel, int32_t n_sel,
                                   const void* conv1_frag_dev, const float* conv1_bias_dev, const void* conv2_frag_dev,
                                   const float* conv2_bias_dev, uint16_t* act_dev, const uint16_t* shared_view_selfcell_dev,
                                   int32_t device_id, void* stream) {
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
    a.h0_out = nullptr; a.h1_out = nullptr;
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

    const int n_cus = policy_n_cus(device_id);
    if (!n_cus) return pfail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    const int per_wave = pol_h0_bytes(grid_size) + pol_h1_bytes(grid_size);
    int wpb = 4;
    while (wpb > 1 && wpb * per_wave > 64 * 1024) wpb >>= 1;
    if (const char* ov = getenv("CTF_POLICY_WPB")) {  // profiling only
        const int v = atoi(ov);
        if (v >= 1 && v <= wpb) wpb = v;
    }
    const size_t sh = (size_t)wpb * per_wave;
    int per_cu = (int)((160 * 1024) / sh);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * wpb > 12) per_cu = 12 / wpb;  // 3 waves per SIMD: what the register budget allows
    if (const char* ov = getenv("CTF_POLICY_BLOCKS_PER_CU")) {  // profiling only: occupancy scaling
        const int v = atoi(ov);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    const int S = n_envs * n_sel;
    int blocks = (S + wpb - 1) / wpb;
    if (blocks > n_cus * per_cu) blocks = n_cus * per_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    if (shared_view_selfcell_dev && n_sel <= 4 && (grid_size == 15 || grid_size == 11)) {
        // agents sharing a view: one wave per env does the shared work once and a small patch per agent
        TeamArgs ta;
        ta.p = a;
        ta.selfcells = shared_view_selfcell_dev;
        ta.A = n_sel;
        // ONE block per CU, one wave per SIMD: measured 1.12 ms for the two teams of an arena step against 1.32 / 1.36 with two /
        // three blocks — every wave of this kernel keeps A activation rows open at once, and the more such streams a CU runs
        // the worse its store path does (the per-agent kernel, one row per wave, is the other way round: 1.80 / 1.57 / 1.51)
        int team_per_cu = 1;
        if (const char* ov = getenv("CTF_POLICY_BLOCKS_PER_CU")) team_per_cu = atoi(ov) >= 1 ? atoi(ov) : 1;  // profiling only
        if (team_per_cu > per_cu) team_per_cu = per_cu;
        int tblocks = (n_envs + wpb - 1) / wpb;
        if (tblocks > n_cus * team_per_cu) tblocks = n_cus * team_per_cu;
        if (grid_size == 15) {
            if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features_team<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features_team<15>, dim3(tblocks), dim3(wpb * WAVE), sh, st, ta);
        } else {
            if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features_team<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL(k_policy_features_team<11>, dim3(tblocks), dim3(wpb * WAVE), sh, st, ta);
        }
    } else if (grid_size == 15) {
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

typedef short i16x4_t __attribute__((ext_vector_type(4)));
// two transposed 4-position blocks (4 positions apart) -> one 8-position MFMA operand of this lane's channel
__device__ __forceinline__ u32x4_t wgrad_tr_operand(const uint8_t* lds_addr, int second_block_bytes) {
    typedef __attribute__((address_space(3))) i16x4_t* lds_v4;
    const i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_addr));
    const i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_addr + second_block_bytes));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return (u32x4_t){l2[0], l2[1], h2[0], h2[1]};
}
// ---- tanh' * incoming gradient for two bf16 pairs (and their float32 values, for the bias-gradient sums)
__device__ __forceinline__ uint32_t tanh_grad2(uint32_t g2, uint32_t h2, float& s0, float& s1) {  // two bf16 pairs -> g * (1 - h * h)
    const float g0 = __builtin_bit_cast(float, g2 << 16), g1 = __builtin_bit_cast(float, g2 & 0xFFFF0000u);
    const float h0 = __builtin_bit_cast(float, h2 << 16), h1 = __builtin_bit_cast(float, h2 & 0xFFFF0000u);
    const float r0 = g0 * (1.0f - h0 * h0), r1 = g1 * (1.0f - h1 * h1);
    s0 += r0;
    s1 += r1;
    return pack_bf16(r0, r1);
}
// ---- the data path of the training front's backward in ONE kernel (grid_size 11 / 15): per sample
//   dz2 = d_act * (1 - act^2)            rows of the activation matrix in; out channels-last [S][P2][32] (the weight gradient of conv2
//                                        is the library's) and, zero-padded by two cells, into LDS as [4 channel octets][G x G][8]
//   dh1 = conv2's data gradient           dh1[i][y][x] = sum over taps, o of W2[o][i][tap] * dz2[o][y - dy][x - dx]: 16x16x32 MFMAs, the
//                                        transposed weights register-resident (A operand), a lane's B operand 8 consecutive o of one cell
//   dz1 = dh1 * (1 - h1^2)               h1 = tanh(conv1) as the forward saved it; out channels-last [S][P1][16]
// and the per-channel sums of dz2 / dz1 in float32 (both bias gradients).  One wave per sample at a time, 20 KB of LDS per wave.
struct DgradArgs {
    const uint16_t* d_act;   // bf16 [S][Kp]
    const uint16_t* act;     // bf16 [S][Kp]
    const uint16_t* h1;      // bf16 [S][P1][16]
    const u32x4_t* w2t;      // [9][64]: [tap][lane][j] = W2[o = 8 (lane >> 4) + j][i = lane & 15][tap], bf16, unscaled
    uint16_t* dz2;           // bf16 [S][P2][32]
    uint16_t* dz1;           // bf16 [S][P1][16]
    float* db2;              // float [32] += , or NULL
    float* db1;              // float [16] += , or NULL
    float* dw2;              // W2 instantiation: float [32][16][9] += conv2's weight gradient (dz2 is then not written at all)
    float* part;             // deterministic mode: float [blocks][DGRAD_PART] = every block's own (dw2 | db2 | db1), else NULL (atomics)
    int64_t S;
    int32_t Kp;
    uint32_t inv_g1, inv_g2;
};
#define DGRAD_PART (32 * 16 * 9 + 32 + 16)
// W2 = true fuses conv2's WEIGHT gradient into the same pass: dz2 never leaves the CU.  Separate kernels moved, per sample, dz2 out
// (7.7 KB) and back in (7.7 KB) and h1 in a second time (5.4 KB) — and the weight-gradient kernel, once its transposing stores were
// gone, ran at the speed of exactly that traffic (0.70 ms per 262 144 samples, 4.9 TB/s).  Both images get the padding the position
// contraction needs (16 columns per gradient row, 18 per activation row, rows past the image: zeros that nothing ever writes), and
// the operands come from the images as they lie — dz2 in its four octet arrays, h1 in its two halves — through the transposing read
// (k_policy_front_wgrad), whose lanes may point anywhere.
template <int TG, bool W2>
__global__ void __launch_bounds__(256) k_policy_front_dgrad(DgradArgs a) {
    extern __shared__ uint32_t lds[];
    constexpr int G = TG, G1 = G - 2, G2 = G - 4, P1 = G1 * G1, P2 = G2 * G2, PP = ((P2 + 31) >> 5) << 5;
    constexpr int RA = G2 + 1 + (G2 & 1 ? 0 : 1), KS = RA / 2;  // the weight gradient's K-steps: two gradient rows of 16 columns each
    constexpr int ZW = W2 ? 18 : G;                              // cells per row of the padded dz2 image (column x + 2 <= 17 when fused)
    constexpr int ZCELLS = G * ZW, ZB = 4 * ZCELLS * 16;         // bytes: the padded dz2 image (4 octet arrays)
    constexpr int H1W = W2 ? 18 : G1, H1R = W2 ? RA + 2 : G1;    // the h1 image: columns / rows (fused: a tap reads up to column 17, row RA + 1)
    constexpr int H1A = W2 ? H1R * H1W * 16 : (((P1 + 15) >> 4) << 4) * 16;  // bytes of one channel half of it
    static_assert(!W2 || (RA + 1 < G && H1R > G1), "the zero rows the position contraction reads exist");
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), wpb = blockDim.x / WAVE;
    uint8_t* zp = (uint8_t*)lds + wave * (ZB + 2 * H1A);
    uint8_t* h1 = zp + ZB;
    u32x4_t w2t[9];
#pragma unroll
    for (int t = 0; t < 9; t++) w2t[t] = a.w2t[t * WAVE + lane];
    {   // the border of the padded image (fused: also the h1 image's padding) stays zero for the whole launch
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int q = lane; q < (W2 ? (ZB + 2 * H1A) / 16 : 4 * ZCELLS); q += WAVE) ((u32x4_t*)zp)[q] = z;
    }
    f32x4_t acc2[W2 ? 9 : 1][2];
#pragma unroll
    for (int t = 0; t < (W2 ? 9 : 1); t++) acc2[t][0] = acc2[t][1] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    const int n2 = lane & 31, hh = lane >> 5;   // dz2 pass: position within a 32-position tile, channel-quad parity
    const int n1 = lane & 15, g1 = lane >> 4;   // data-gradient pass: position within a 16-position tile, octet of o / quad of i
    float s2[16], s1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; r++) s2[r] = 0.0f;
    // A sample's inputs — its rows of d_act and act (8 bytes per lane, tile and channel quad) and its h1 image (16-byte pieces) — are
    // requested one sample ahead, right after the registers of the current sample's rows have been consumed: they travel while the
    // MFMA section runs.  (Round 3 loaded them at the top of their own sample: two exposed HBM round trips per sample and wave.)
    constexpr int NT2 = PP / 32, NH1 = (2 * P1 + WAVE - 1) / WAVE;
    u32x2_t gq[NT2][4], hq[NT2][4];
    u32x4_t h1q[NH1];
    auto fetch = [&](int64_t sn) {
        const uint16_t* drow = a.d_act + (size_t)sn * a.Kp;
        const uint16_t* arow = a.act + (size_t)sn * a.Kp;
#pragma unroll
        for (int t = 0; t < NT2; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const size_t col = ((size_t)(2 * q + hh) * PP + 32 * t + n2) * 4;  // (positions >= P2 of the last tile: padding columns of the row)
                gq[t][q] = *(const u32x2_t*)(drow + col);
                hq[t][q] = *(const u32x2_t*)(arow + col);
            }
        const uint8_t* hsrc = (const uint8_t*)(a.h1 + (size_t)sn * P1 * 16);
#pragma unroll
        for (int u = 0; u < NH1; u++) h1q[u] = *(const u32x4_t*)(hsrc + (size_t)min(lane + WAVE * u, 2 * P1 - 1) * 16);
    };
    const int64_t s_first = (int64_t)blockIdx.x * wpb + wave, s_stride = (int64_t)gridDim.x * wpb;
    if (s_first < a.S) fetch(s_first);
    for (int64_t s = s_first; s < a.S; s += s_stride) {
        // h1 of this sample: channels-last [P1][16] -> [2 halves][rows][columns][8]
#pragma unroll
        for (int u = 0; u < NH1; u++) {
            const int idx = lane + WAVE * u;
            if (idx < 2 * P1) {
                const int pos = idx >> 1, y = (int)(((uint32_t)pos * a.inv_g1) >> 16), x = pos - y * G1;
                *(u32x4_t*)(h1 + (idx & 1) * H1A + (y * H1W + x) * 16) = h1q[u];
            }
        }
#pragma unroll
        for (int t = 0; t < NT2; t++) {
            const int pa = 32 * t + n2;
            const bool ok = pa < P2;
            const int pc = ok ? pa : P2 - 1;
            const int y = (int)(((uint32_t)pc * a.inv_g2) >> 16), x = pc - y * G2;
            uint8_t* cell = zp + ((y + 2) * ZW + (x + 2)) * 16 + hh * 8;
#pragma unroll
            for (int q = 0; q < 4; q++) {  // channels 8 q + 4 hh .. + 3: octet q, its half hh
                float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                const u32x2_t o = {tanh_grad2(gq[t][q][0], hq[t][q][0], d0, d1), tanh_grad2(gq[t][q][1], hq[t][q][1], d2, d3)};
                if (ok) {
                    *(u32x2_t*)(cell + q * ZCELLS * 16) = o;
                    s2[4 * q] += d0; s2[4 * q + 1] += d1; s2[4 * q + 2] += d2; s2[4 * q + 3] += d3;
                }
            }
        }
        fetch(s + s_stride < a.S ? s + s_stride : s);  // (behind the last sample it re-reads itself: every lane always issues the same loads)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // dz2 out, channels-last [P2][32]: 16-byte pieces (position p, octet q) read back from the padded image — consecutive lanes write
        // consecutive pieces (straight from the registers above they were 8-byte pieces 64 bytes apart)
        if (!W2) {
            uint8_t* z2out = (uint8_t*)(a.dz2 + (size_t)s * P2 * 32);
#pragma unroll
            for (int u = 0; u < (4 * P2 + WAVE - 1) / WAVE; u++) {
                const int idx = lane + WAVE * u;
                if (idx < 4 * P2) {
                    const int pos = idx >> 2, q = idx & 3;
                    const int y = (int)(((uint32_t)pos * a.inv_g2) >> 16), x = pos - y * G2;
                    *(u32x4_t*)(z2out + (size_t)idx * 16) = *(const u32x4_t*)(zp + q * ZCELLS * 16 + ((y + 2) * ZW + (x + 2)) * 16);
                }
            }
        }
        uint16_t* z1out = a.dz1 + (size_t)s * P1 * 16;
        constexpr int T1 = (P1 + 15) >> 4;
        // two tiles per pass: two independent accumulation chains (a chain of nine dependent MFMAs per tile leaves the pipe idle most of
        // the time when the wave is alone on its SIMD)
        auto dgrad_tile_base = [&](int t, int& p, int& yy, int& xx) {
            p = 16 * t + n1;
            const int pc = p < P1 ? p : P1 - 1;
            yy = (int)(((uint32_t)pc * a.inv_g1) >> 16);
            xx = pc - yy * G1;
            return zp + g1 * ZCELLS * 16 + ((yy + 2) * ZW + (xx + 2)) * 16;
        };
        auto dgrad_tile_out = [&](const f32x4_t& acc, int p, int yy, int xx) {
            const int pc = p < P1 ? p : P1 - 1;
            const u32x2_t hv = *(const u32x2_t*)(h1 + (g1 >> 1) * H1A + (W2 ? yy * H1W + xx : pc) * 16 + (g1 & 1) * 8);  // h1 channels 4 g1 .. + 3 of this position
            float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            const u32x2_t o = {tanh_grad2(pack_bf16(acc[0], acc[1]), hv[0], d0, d1), tanh_grad2(pack_bf16(acc[2], acc[3]), hv[1], d2, d3)};
            if (p < P1) {
                *(u32x2_t*)(z1out + (size_t)p * 16 + 4 * g1) = o;
                s1[0] += d0; s1[1] += d1; s1[2] += d2; s1[3] += d3;
            }
        };
        int t = 0;
#pragma unroll 1
        for (; t + 1 < T1; t += 2) {
            int pa_, ya_, xa_, pb_, yb_, xb_;
            const uint8_t* base_a = dgrad_tile_base(t, pa_, ya_, xa_);
            const uint8_t* base_b = dgrad_tile_base(t + 1, pb_, yb_, xb_);
            f32x4_t acc_a = {0.0f, 0.0f, 0.0f, 0.0f}, acc_b = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int off = ((tap / 3) * ZW + (tap % 3)) * 16;
                const u32x4_t ba = *(const u32x4_t*)(base_a - off);
                const u32x4_t bb = *(const u32x4_t*)(base_b - off);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w2t[tap]), as_bf16x8(ba), acc_a, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w2t[tap]), as_bf16x8(bb), acc_b, 0, 0, 0);
            }
            dgrad_tile_out(acc_a, pa_, ya_, xa_);
            dgrad_tile_out(acc_b, pb_, yb_, xb_);
        }
        if (t < T1) {
            int pa_, ya_, xa_;
            const uint8_t* base_a = dgrad_tile_base(t, pa_, ya_, xa_);
            f32x4_t acc_a = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const u32x4_t ba = *(const u32x4_t*)(base_a - ((tap / 3) * ZW + (tap % 3)) * 16);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w2t[tap]), as_bf16x8(ba), acc_a, 0, 0, 0);
            }
            dgrad_tile_out(acc_a, pa_, ya_, xa_);
        }
        if (W2) {
            // ---- conv2's weight gradient: dW2[o][i][tap] += sum over positions of dz2[o][y][x] * h1[i][y + dy][x + dx].  K-step ks = gradient
            // rows 2 ks, 2 ks + 1; lane 4 q + p of a 16-lane group supplies block row q (position x0 + q) and channels 4 p .. 4 p + 3:
            // octet (p >> 1) + 2 h of dz2 / half p >> 1 of h1, 8 bytes in.  (All 64 lanes are active, as the transposing read requires.)
            const int kg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
            const int za = (tp >> 1) * ZCELLS * 16 + (((kg >> 1) + 2) * ZW + 8 * (kg & 1) + tq + 2) * 16 + (tp & 1) * 8;
            const int hb = (tp >> 1) * H1A + ((kg >> 1) * H1W + 8 * (kg & 1) + tq) * 16 + (tp & 1) * 8;
#pragma unroll 1
            for (int ks = 0; ks < KS; ks++) {
                u32x4_t av[2];
#pragma unroll
                for (int h = 0; h < 2; h++) av[h] = wgrad_tr_operand(zp + za + h * 2 * ZCELLS * 16 + ks * 2 * ZW * 16, 4 * 16);
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const u32x4_t bv = wgrad_tr_operand(h1 + hb + ((ks * 2 + t / 3) * H1W + t % 3) * 16, 4 * 16);
#pragma unroll
                    for (int h = 0; h < 2; h++) acc2[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[h]), as_bf16x8(bv), acc2[t][h], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();  // the next sample rewrites both images
    }
    // bias gradients: lanes with the same hh (dz2: channel 8 q + 4 hh + r) / the same g1 (dz1: channel 4 g1 + r) hold the same channels
    __syncthreads();  // every wave is through with its images: the block's dynamic LDS (>= 16 KB) is the reduction's scratch now
    float* red = (float*)lds;
    if (a.db2) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[r * 256 + threadIdx.x] = s2[r];
        __syncthreads();
        if (threadIdx.x < 32) {  // channel c = 8 q + 4 hh + r2  <->  (r = 4 q + r2, hh)
            const int c = threadIdx.x, q = c >> 3, h2 = (c >> 2) & 1, r = 4 * q + (c & 3);
            float t = 0.0f;
            for (int k = 0; k < 256; k++)
                if (((k & 63) >> 5) == h2) t += red[r * 256 + k];
            if (a.part) a.part[(size_t)blockIdx.x * DGRAD_PART + 32 * 16 * 9 + c] = t;
            else atomicAdd(a.db2 + c, t);
        }
        __syncthreads();
    }
    if (a.db1) {
#pragma unroll
        for (int r = 0; r < 4; r++) red[r * 256 + threadIdx.x] = s1[r];
        __syncthreads();
        if (threadIdx.x < 16) {  // channel c = 4 g1 + r
            const int c = threadIdx.x, gg = c >> 2, r = c & 3;
            float t = 0.0f;
            for (int k = 0; k < 256; k++)
                if (((k & 63) >> 4) == gg) t += red[r * 256 + k];
            if (a.part) a.part[(size_t)blockIdx.x * DGRAD_PART + 32 * 16 * 9 + 32 + c] = t;
            else atomicAdd(a.db1 + c, t);
        }
    }
    if (W2) {  // the block's four partial weight gradients -> one; D tile: lane holds rows m = 4 (lane >> 4) + r of column n = lane & 15
        __syncthreads();
        const int mn = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[((wave * 9 + t) * 32 + 16 * h + 4 * kg + r) * 16 + mn] = acc2[W2 ? t : 0][h][r];
        __syncthreads();
        for (int e = threadIdx.x; e < 9 * 32 * 16; e += blockDim.x) {
            float v = 0.0f;
            for (int w = 0; w < wpb; w++) v += red[w * 9 * 32 * 16 + e];
            const int t = e / (32 * 16), oi = e - t * (32 * 16);
            if (a.part) a.part[(size_t)blockIdx.x * DGRAD_PART + (size_t)oi * 9 + t] = v;
            else atomicAdd(a.dw2 + (size_t)oi * 9 + t, v);  // [out][in][tap]
        }
    }
}

extern "C" int ctf_policy_front_dgrad(const uint16_t* d_act_dev, const uint16_t* act_dev, const uint16_t* h1_dev, const void* conv2_t_frag_dev,
                                      int64_t n_samples, int32_t grid_size, int32_t meta_len, uint16_t* dz2_dev, uint16_t* dz1_dev,
                                      float* bias2_grad_dev, float* bias1_grad_dev, int32_t device_id, void* stream) {
    if (!d_act_dev || !act_dev || !h1_dev || !conv2_t_frag_dev || !dz2_dev || !dz1_dev) return pfail("null argument");
    if (grid_size != 15 && grid_size != 11) return pfail("the training front is built for grid_size 11 and 15 (the reference's maps)");
    if (n_samples < 0) return pfail("n_samples out of range");
    if (((uintptr_t)d_act_dev | (uintptr_t)act_dev | (uintptr_t)dz2_dev | (uintptr_t)dz1_dev) & 7) return pfail("8-byte alignment");
    if (((uintptr_t)h1_dev | (uintptr_t)conv2_t_frag_dev) & 15) return pfail("h1_dev / conv2_t_frag_dev must be 16-byte aligned");
    if (!n_samples) return 0;
    DgradArgs a;
    a.d_act = d_act_dev; a.act = act_dev; a.h1 = h1_dev; a.w2t = (const u32x4_t*)conv2_t_frag_dev; a.dz2 = dz2_dev; a.dz1 = dz1_dev;
    a.db2 = bias2_grad_dev; a.db1 = bias1_grad_dev; a.dw2 = nullptr; a.S = n_samples; a.Kp = ctf_policy_act_stride(grid_size, meta_len);
    const int G1 = grid_size - 2, G2 = grid_size - 4;
    a.inv_g1 = (65536 + G1 - 1) / G1;
    a.inv_g2 = (65536 + G2 - 1) / G2;
    for (int p = 0; p < G1 * G1; p++)
        if ((int)(((uint32_t)p * a.inv_g1) >> 16) != p / G1) return pfail("internal: reciprocal of G-2 not exact");
    for (int p = 0; p < G2 * G2; p++)
        if ((int)(((uint32_t)p * a.inv_g2) >> 16) != p / G2) return pfail("internal: reciprocal of G-4 not exact");
    const int n_cus = policy_n_cus(device_id);
    if (!n_cus) return pfail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    const int wpb = 4;
    const int per_wave = 4 * grid_size * grid_size * 16 + 2 * ((((G1 * G1) + 15) >> 4) << 4) * 16;
    const size_t sh = (size_t)wpb * per_wave;
    int per_cu = (int)((160 * 1024) / sh);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    int64_t blocks = (n_samples + wpb - 1) / wpb;
    if (blocks > (int64_t)n_cus * per_cu) blocks = (int64_t)n_cus * per_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    const DetWorkspace det = ctf_policy_det(device_id);
    a.part = (det.ptr && (a.db2 || a.db1)) ? det.ptr : nullptr;
    if (a.part && blocks * DGRAD_PART > det.floats) err = hipErrorOutOfMemory;
    if (err != hipSuccess) {
    } else if (grid_size == 15) {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_front_dgrad<15, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_dgrad<15, false>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
    } else {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_front_dgrad<11, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_dgrad<11, false>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess && a.part && a.db2) err = ctf_policy_det_reduce(a.part + 32 * 16 * 9, (int)blocks, DGRAD_PART, 32, a.db2, st);
    if (err == hipSuccess && a.part && a.db1) err = ctf_policy_det_reduce(a.part + 32 * 16 * 9 + 32, (int)blocks, DGRAD_PART, 16, a.db1, st);
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err == hipErrorOutOfMemory) return pfail("deterministic mode: the registered workspace is too small for this launch (ctf_policy_set_deterministic)");
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}

// ---- the two WEIGHT gradients of the training front, contraction over positions on the matrix cores (grid_size 11 / 15)
//   dW2[o][i][tap] = sum over samples, conv2 output positions (y, x) of dz2[o][y][x] * h1[i][y + dy][x + dx]
//   dW1[o][c][tap] = sum over samples, conv1 output positions (y, x) of dz1[o][y][x] * x0[c][y + dy][x + dx]      (x0: the one-hot image)
// One 16x16x32 MFMA contracts 32 positions = two image rows of 16 columns (the columns past the row's end are zeros of the A operand).
// Both operands must then be POSITION-contiguous per lane (8 consecutive x of one channel), i.e. transposed against the channels-last
// tensors the other kernels exchange.  Round 3 transposed them into LDS with 2-byte stores (112 store instructions per sample and
// wave: 1.5 ms per launch for 0.2 ms of MFMAs).  Round 4: the images stay CHANNELS-LAST in LDS — whole 16-byte pieces as they arrive,
// 14 stores — and gfx950's transposing read delivers the operands: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i
// (= channel) of a block of 4 rows (= 4 consecutive positions), so an operand register pair is two such reads:
//   A  [RA rows][16 cols][CO channels]     the gradient image; rows >= GO and cols >= GO stay zero
//   B  [RB rows][18 cols][16 channels]     the activation image; a tap (dy, dx) reads it from row + dy, column + dx on
// Accumulators (one f32x4 tile per tap and 16 out channels) live in registers across all samples of the wave; at the end the block's
// waves are summed through LDS and added to the float32 result with one atomicAdd per element and block.
struct WgradArgs {
    const uint16_t* grad;    // bf16 channels-last [S][GO*GO][CO]: dz2 (CO = 32) or dz1 (CO = 16)
    const uint16_t* img;     // bf16 channels-last [S][GI*GI][16]: h1, or NULL when the image is built from codes
    const uint8_t* codes;    // uint8 [S][GI*GI] (conv1 only): the one-hot image's source
    float* dw;               // float [CO][16][9] += (the caller zeroes it)
    float* part;             // deterministic mode: float [blocks][CO * 16 * 9], every block's own sum, else NULL (atomics)
    int64_t S;
};
// GO: side of the gradient image, GI = GO + 2: side of the activation image, CO: out channels (16 or 32)
template <int GO, int CO, bool FROM_CODES>
__global__ void __launch_bounds__(128) k_policy_front_wgrad(WgradArgs a) {
    extern __shared__ uint32_t lds[];
    constexpr int GI = GO + 2, PO = GO * GO, PI = GI * GI;
    constexpr int RA = GO + 1 + (GO & 1 ? 0 : 1);      // rows of A: GO + at least one zero row, an even count
    constexpr int KS = RA / 2;                          // K-steps (two rows each)
    constexpr int RB = RA + 2;                          // rows of B read: up to RA - 1 + 2
    constexpr int WB = 18;                              // columns of B read: up to 15 + 2
    constexpr int A_POS = CO * 2, A_ROW = 16 * A_POS;  // bytes per position / per row of A
    constexpr int B_POS = 32, B_ROW = WB * B_POS;
    constexpr int A_BYTES = RA * A_ROW, B_BYTES = RB * B_ROW;
    constexpr int NH = CO / 16;                         // 16-channel halves of the out channels
    constexpr int NPA = PO * (CO / 8), NA = (NPA + WAVE - 1) / WAVE;             // 16-byte pieces of the gradient image, per lane
    constexpr int NPB = FROM_CODES ? PI : PI * 2, NB = (NPB + WAVE - 1) / WAVE;  // code bytes / 16-byte pieces of the activation image
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), wpb = blockDim.x / WAVE;
    uint8_t* A = (uint8_t*)lds + wave * (A_BYTES + B_BYTES);
    uint8_t* B = A + A_BYTES;
    {   // zeros everywhere once: the A rows / columns past the image and the B rows / columns past the image stay zero
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int q = lane; q < (A_BYTES + B_BYTES) / 16; q += WAVE) ((u32x4_t*)A)[q] = z;
    }
    f32x4_t acc[9][NH];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int h = 0; h < NH; h++) acc[t][h] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    const int mn = lane & 15, kg = lane >> 4;           // operand row / column, k-group: row parity kg >> 1, columns 8 (kg & 1) ..
    // the transposing read: lane 4 q + p of a 16-lane group supplies the address of block row q (position x0 + q), channels 4 p .. 4 p + 3
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int a_off = (kg >> 1) * A_ROW + (8 * (kg & 1) + tq) * A_POS + tp * 8;   // + h * 32 bytes, + ks * 2 * A_ROW
    const int b_off = (kg >> 1) * B_ROW + (8 * (kg & 1) + tq) * B_POS + tp * 8;   // + (dy * B_ROW + dx * B_POS), + ks * 2 * B_ROW
    // the next sample's inputs travel while this one is contracted: a lane's pieces sit in registers across the MFMA section
    u32x4_t ga[NA], gb[FROM_CODES ? 1 : NB];
    uint32_t cb[FROM_CODES ? NB : 1];
    const int64_t s0 = (int64_t)blockIdx.x * wpb + wave, stride = (int64_t)gridDim.x * wpb;
    auto fetch = [&](int64_t s) {
        const u32x4_t* gsrc = (const u32x4_t*)(a.grad + (size_t)s * PO * CO);
#pragma unroll
        for (int u = 0; u < NA; u++) ga[u] = gsrc[min(lane + WAVE * u, NPA - 1)];
        if (FROM_CODES) {
            const uint8_t* cp = a.codes + (size_t)s * PI;
#pragma unroll
            for (int u = 0; u < NB; u++) cb[u] = cp[min(lane + WAVE * u, NPB - 1)];
        } else {
            const u32x4_t* isrc = (const u32x4_t*)(a.img + (size_t)s * PI * 16);
#pragma unroll
            for (int u = 0; u < NB; u++) gb[u] = isrc[min(lane + WAVE * u, NPB - 1)];
        }
    };
    if (s0 < a.S) fetch(s0);
    for (int64_t s = s0; s < a.S; s += stride) {
        // ---- A: the gradient image, channels-last [PO][CO] -> [RA][16][CO]: every 16-byte piece as it is, at its (row, column)
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int idx = lane + WAVE * u;
            if (idx < NPA) {
                const int pos = idx / (CO / 8), oct = idx - pos * (CO / 8);
                const int y = pos / GO, x = pos - y * GO;
                *(u32x4_t*)(A + y * A_ROW + x * A_POS + oct * 16) = ga[u];
            }
        }
        // ---- B: the activation image [RB rows][WB columns][16 channels]
        if (FROM_CODES) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            for (int q = lane; q < B_BYTES / 16; q += WAVE) ((u32x4_t*)B)[q] = z;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int c = lane + WAVE * u;
                if (c < NPB) {
                    const uint32_t code = cb[u], ch = code & 0x7Fu;
                    const int y = c / GI, x = c - y * GI;
                    uint8_t* dst = B + y * B_ROW + x * B_POS;
                    if (ch != 0 && ch < 16) *(uint16_t*)(dst + ch * 2) = 0x3F80;
                    if (code >> 7) *(uint16_t*)(dst) = 0x3F80;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int idx = lane + WAVE * u;
                if (idx < NPB) {
                    const int pos = idx >> 1, oct = idx & 1;
                    const int y = pos / GI, x = pos - y * GI;
                    *(u32x4_t*)(B + y * B_ROW + x * B_POS + oct * 16) = gb[u];
                }
            }
        }
        {   // every lane issues the same loads, also behind the last sample (it re-reads itself)
            const int64_t sn = s + stride < a.S ? s + stride : s;
            fetch(sn);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // ---- contraction: K-step ks = gradient rows 2 ks, 2 ks + 1; tap (dy, dx) reads the activation rows + dy from column + dx on.
        // (All 64 lanes are active here, as the transposing read requires.)
#pragma unroll 1
        for (int ks = 0; ks < KS; ks++) {
            u32x4_t av[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) av[h] = wgrad_tr_operand(A + a_off + h * 32 + ks * 2 * A_ROW, 4 * A_POS);
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const u32x4_t bv = wgrad_tr_operand(B + b_off + (ks * 2 + t / 3) * B_ROW + (t % 3) * B_POS, 4 * B_POS);
#pragma unroll
                for (int h = 0; h < NH; h++) acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[h]), as_bf16x8(bv), acc[t][h], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();  // the next sample rewrites both images
    }
    // ---- the block's partial sums -> one; D tile: lane holds rows m = 4 (lane >> 4) + r of column n = lane & 15
    __syncthreads();
    float* red = (float*)lds;  // [wave][tap][CO][16]
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[((wave * 9 + t) * CO + 16 * h + 4 * kg + r) * 16 + mn] = acc[t][h][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * CO * 16; e += blockDim.x) {
        float v = 0.0f;
        for (int w = 0; w < wpb; w++) v += red[w * 9 * CO * 16 + e];
        const int t = e / (CO * 16), oi = e - t * (CO * 16);
        if (a.part) a.part[(size_t)blockIdx.x * (9 * CO * 16) + (size_t)oi * 9 + t] = v;
        else atomicAdd(a.dw + (size_t)oi * 9 + t, v);  // [out][in][tap]
    }
}

extern "C" int ctf_policy_front_wgrad(const uint16_t* dz2_dev, const uint16_t* h1_dev, const uint16_t* dz1_dev, const uint8_t* codes_dev,
                                      int64_t n_samples, int32_t grid_size, float* dw2_dev, float* dw1_dev, int32_t device_id, void* stream) {
    if (!dz2_dev || !h1_dev || !dz1_dev || !codes_dev || !dw2_dev || !dw1_dev) return pfail("null argument");
    if (grid_size != 15 && grid_size != 11) return pfail("the training front is built for grid_size 11 and 15 (the reference's maps)");
    if (n_samples < 0) return pfail("n_samples out of range");
    if (((uintptr_t)dz2_dev | (uintptr_t)h1_dev | (uintptr_t)dz1_dev) & 15) return pfail("16-byte alignment");
    if (!n_samples) return 0;
    const int n_cus = policy_n_cus(device_id);
    if (!n_cus) return pfail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    const int wpb = 2;
    auto launch = [&](auto kernel, int go, int co, const WgradArgs& a) {
        const int ra = go + 1 + ((go & 1) ? 0 : 1), rb = ra + 2;
        size_t sh = (size_t)wpb * ((size_t)ra * 16 * co * 2 + (size_t)rb * 18 * 32);  // k_policy_front_wgrad's A_BYTES + B_BYTES per wave
        const size_t red = (size_t)wpb * 9 * co * 16 * 4;
        if (sh < red) sh = red;
        int per_cu = (int)((160 * 1024) / sh);
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;  // 8 waves per CU = 2 per SIMD
        int64_t blocks = (a.S + wpb - 1) / wpb;
        if (blocks > (int64_t)n_cus * per_cu) blocks = (int64_t)n_cus * per_cu;
        WgradArgs b = a;
        const DetWorkspace det = ctf_policy_det(device_id);
        b.part = det.ptr;
        if (err == hipSuccess && b.part && blocks * 9 * co * 16 > det.floats) err = hipErrorOutOfMemory;
        if (err == hipSuccess && sh > 48 * 1024) err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, b);
        if (err == hipSuccess && b.part) err = ctf_policy_det_reduce(b.part, (int)blocks, 9 * co * 16, 9 * co * 16, b.dw, st);
    };
    WgradArgs a2, a1;
    a2.grad = dz2_dev; a2.img = h1_dev; a2.codes = nullptr; a2.dw = dw2_dev; a2.S = n_samples;
    a1.grad = dz1_dev; a1.img = nullptr; a1.codes = codes_dev; a1.dw = dw1_dev; a1.S = n_samples;
    if (grid_size == 15) {
        launch(k_policy_front_wgrad<11, 32, false>, 11, 32, a2);
        launch(k_policy_front_wgrad<13, 16, true>, 13, 16, a1);
    } else {
        launch(k_policy_front_wgrad<7, 32, false>, 7, 32, a2);
        launch(k_policy_front_wgrad<9, 16, true>, 9, 16, a1);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err == hipErrorOutOfMemory) return pfail("deterministic mode: the registered workspace is too small for this launch (ctf_policy_set_deterministic)");
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}

// The whole backward of the training front: conv2's data AND weight gradient in one pass (k_policy_front_dgrad<.., true>: dz2 never
// leaves the CU), then conv1's weight gradient from the dz1 that pass wrote and the code bytes.
extern "C" int ctf_policy_front_backward(const uint16_t* d_act_dev, const uint16_t* act_dev, const uint16_t* h1_dev, const uint8_t* codes_dev,
                                         const void* conv2_t_frag_dev, int64_t n_samples, int32_t grid_size, int32_t meta_len,
                                         uint16_t* dz1_dev, float* dw2_dev, float* dw1_dev, float* bias2_grad_dev, float* bias1_grad_dev,
                                         int32_t device_id, void* stream) {
    if (!d_act_dev || !act_dev || !h1_dev || !codes_dev || !conv2_t_frag_dev || !dz1_dev || !dw2_dev || !dw1_dev) return pfail("null argument");
    if (grid_size != 15 && grid_size != 11) return pfail("the training front is built for grid_size 11 and 15 (the reference's maps)");
    if (n_samples < 0) return pfail("n_samples out of range");
    if (((uintptr_t)d_act_dev | (uintptr_t)act_dev) & 7) return pfail("8-byte alignment");
    if (((uintptr_t)h1_dev | (uintptr_t)conv2_t_frag_dev | (uintptr_t)dz1_dev) & 15) return pfail("h1_dev / conv2_t_frag_dev / dz1_dev must be 16-byte aligned");
    if (!n_samples) return 0;
    DgradArgs a;
    a.d_act = d_act_dev; a.act = act_dev; a.h1 = h1_dev; a.w2t = (const u32x4_t*)conv2_t_frag_dev; a.dz2 = nullptr; a.dz1 = dz1_dev;
    a.db2 = bias2_grad_dev; a.db1 = bias1_grad_dev; a.dw2 = dw2_dev; a.S = n_samples; a.Kp = ctf_policy_act_stride(grid_size, meta_len);
    const int G1 = grid_size - 2, G2 = grid_size - 4;
    a.inv_g1 = (65536 + G1 - 1) / G1;
    a.inv_g2 = (65536 + G2 - 1) / G2;
    for (int p = 0; p < G1 * G1; p++)
        if ((int)(((uint32_t)p * a.inv_g1) >> 16) != p / G1) return pfail("internal: reciprocal of G-2 not exact");
    for (int p = 0; p < G2 * G2; p++)
        if ((int)(((uint32_t)p * a.inv_g2) >> 16) != p / G2) return pfail("internal: reciprocal of G-4 not exact");
    const int n_cus = policy_n_cus(device_id);
    if (!n_cus) return pfail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    const DetWorkspace det = ctf_policy_det(device_id);
    {   // the fused pass: four waves per block, one block per CU (the padded images are 25 KB per wave, the registers one wave per SIMD)
        const int wpb = 4;
        const int ra = G2 + 1 + ((G2 & 1) ? 0 : 1);
        const size_t per_wave = (size_t)4 * grid_size * 18 * 16 + (size_t)2 * (ra + 2) * 18 * 16;  // k_policy_front_dgrad<.., true>: ZB + 2 H1A
        size_t sh = (size_t)wpb * per_wave;
        const size_t red = (size_t)wpb * 9 * 32 * 16 * 4;
        if (sh < red) sh = red;
        int64_t blocks = (n_samples + wpb - 1) / wpb;
        if (blocks > (int64_t)n_cus) blocks = n_cus;
        a.part = det.ptr;
        if (a.part && blocks * DGRAD_PART > det.floats) err = hipErrorOutOfMemory;
        if (err != hipSuccess) {
        } else if (grid_size == 15) {
            err = hipFuncSetAttribute((const void*)k_policy_front_dgrad<15, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_dgrad<15, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
        } else {
            err = hipFuncSetAttribute((const void*)k_policy_front_dgrad<11, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_dgrad<11, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
        }
        if (err == hipSuccess && a.part) err = ctf_policy_det_reduce(a.part, (int)blocks, DGRAD_PART, 32 * 16 * 9, a.dw2, st);
        if (err == hipSuccess && a.part && a.db2) err = ctf_policy_det_reduce(a.part + 32 * 16 * 9, (int)blocks, DGRAD_PART, 32, a.db2, st);
        if (err == hipSuccess && a.part && a.db1) err = ctf_policy_det_reduce(a.part + 32 * 16 * 9 + 32, (int)blocks, DGRAD_PART, 16, a.db1, st);
    }
    {   // conv1's weight gradient (as in ctf_policy_front_wgrad)
        const int wpb = 2, go = G1, co = 16;
        const int ra = go + 1 + ((go & 1) ? 0 : 1), rb = ra + 2;
        size_t sh = (size_t)wpb * ((size_t)ra * 16 * co * 2 + (size_t)rb * 18 * 32);
        const size_t red = (size_t)wpb * 9 * co * 16 * 4;
        if (sh < red) sh = red;
        int per_cu = (int)((160 * 1024) / sh);
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;
        int64_t blocks = (n_samples + wpb - 1) / wpb;
        if (blocks > (int64_t)n_cus * per_cu) blocks = (int64_t)n_cus * per_cu;
        WgradArgs a1;
        a1.grad = dz1_dev; a1.img = nullptr; a1.codes = codes_dev; a1.dw = dw1_dev; a1.S = n_samples;
        a1.part = det.ptr;
        if (err == hipSuccess && a1.part && blocks * 9 * co * 16 > det.floats) err = hipErrorOutOfMemory;
        if (grid_size == 15) {
            if (err == hipSuccess && sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_front_wgrad<13, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_wgrad<13, 16, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a1);
        } else {
            if (err == hipSuccess && sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_front_wgrad<9, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            if (err == hipSuccess) hipLaunchKernelGGL((k_policy_front_wgrad<9, 16, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a1);
        }
        if (err == hipSuccess && a1.part) err = ctf_policy_det_reduce(a1.part, (int)blocks, 9 * co * 16, 9 * co * 16, a1.dw, st);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err == hipErrorOutOfMemory) return pfail("deterministic mode: the registered workspace is too small for this launch (ctf_policy_set_deterministic)");
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}

extern "C" int ctf_policy_features_train(const uint8_t* codes_dev, const uint16_t* meta_dev, int64_t n_samples, int32_t grid_size,
                                         int32_t meta_len, const void* conv1_frag_dev, const float* conv1_bias_dev,
                                         const void* conv2_frag_dev, const float* conv2_bias_dev, uint16_t* act_dev, uint16_t* h0_dev,
                                         uint16_t* h1_dev, int32_t device_id, void* stream) {
    if (!codes_dev || !meta_dev || !conv1_frag_dev || !conv1_bias_dev || !conv2_frag_dev || !conv2_bias_dev || !act_dev || !h1_dev)
        return pfail("null argument");
    if (grid_size != 15 && grid_size != 11) return pfail("the training front is built for grid_size 11 and 15 (the reference's maps)");
    if (n_samples < 1 || n_samples > 0x7FFFFFFF) return pfail("n_samples out of range");
    if (meta_len < 2 || (meta_len & 1)) return pfail("meta_len must be even (2N + 6)");
    if (((uintptr_t)act_dev & 15) || ((uintptr_t)h0_dev & 15) || ((uintptr_t)h1_dev & 15) || ((uintptr_t)meta_dev & 3))
        return pfail("act_dev / h0_dev / h1_dev must be 16-byte, meta_dev 4-byte aligned");
    PolicyArgs a;
    a.codes = codes_dev; a.meta = meta_dev; a.act = act_dev;
    a.w1frag = (const u32x4_t*)conv1_frag_dev; a.b1 = conv1_bias_dev;
    a.w2frag = (const u32x4_t*)conv2_frag_dev; a.b2 = conv2_bias_dev;
    a.n_envs = (int32_t)n_samples; a.N = 1; a.G = grid_size; a.M = meta_len; a.n_sel = 1;  // every row of codes_dev is one sample
    a.Kp = ctf_policy_act_stride(grid_size, meta_len);
    a.sel_pack = 0;
    a.h0_out = h0_dev; a.h1_out = h1_dev;
    const int G1 = grid_size - 2, G2 = grid_size - 4;
    a.inv_g1 = (65536 + G1 - 1) / G1;
    a.inv_g2 = (65536 + G2 - 1) / G2;
    for (int p = 0; p < G1 * G1; p++)
        if ((int)(((uint32_t)p * a.inv_g1) >> 16) != p / G1) return pfail("internal: reciprocal of G-2 not exact");
    for (int p = 0; p < G2 * G2; p++)
        if ((int)(((uint32_t)p * a.inv_g2) >> 16) != p / G2) return pfail("internal: reciprocal of G-4 not exact");
    const int n_cus = policy_n_cus(device_id);
    if (!n_cus) return pfail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    const int per_wave = pol_h0_bytes(grid_size) + pol_h1_bytes(grid_size);
    const int wpb = 4;
    const size_t sh = (size_t)wpb * per_wave;
    int per_cu = (int)((160 * 1024) / sh);
    if (per_cu * wpb > 12) per_cu = 12 / wpb;  // 3 waves per SIMD, as in ctf_policy_features
    int64_t blocks = (n_samples + wpb - 1) / wpb;
    if (blocks > (int64_t)n_cus * per_cu) blocks = (int64_t)n_cus * per_cu;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    if (grid_size == 15) {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features<15, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_features<15, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
    } else {
        if (sh > 48 * 1024) err = hipFuncSetAttribute((const void*)k_policy_features<11, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL((k_policy_features<11, true>), dim3((unsigned)blocks), dim3(wpb * WAVE), sh, st, a);
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}

extern "C" int ctf_policy_head(const uint16_t* fc1_out_dev, int64_t n_samples, const void* fc2_frag_dev, const float* fc2_bias_dev,
                               const void* head_frag_dev, const float* head_bias_dev, const float* mask_decision_dev,
                               const int32_t* given_action_dev, int32_t n_actions, uint64_t seed, uint64_t offset,
                               int32_t* action_dev, float* logprob_dev, float* entropy_dev, float* value_dev, float* logits_dev,
                               int32_t device_id, void* stream) {
    if (!fc1_out_dev || !fc2_frag_dev || !fc2_bias_dev || !head_frag_dev || !head_bias_dev || !action_dev || !logprob_dev ||
        !entropy_dev || !value_dev)
        return pfail("null argument");
    if (n_samples < 1) return pfail("n_samples must be >= 1");
    if (n_actions < 1 || n_actions > 15) return pfail("n_actions outside 1..15");
    if ((uintptr_t)fc1_out_dev & 15) return pfail("fc1_out_dev must be 16-byte aligned");
    HeadArgs a;
    a.y1 = fc1_out_dev; a.fc2_frag = (const u32x4_t*)fc2_frag_dev; a.fc2_bias = fc2_bias_dev;
    a.head_frag = (const u32x4_t*)head_frag_dev; a.head_bias = head_bias_dev;
    a.mask = mask_decision_dev; a.given = given_action_dev;
    a.action = action_dev; a.logprob = logprob_dev; a.entropy = entropy_dev; a.value = value_dev; a.logits = logits_dev;
    a.B = n_samples; a.A = n_actions; a.seed = seed; a.offset = offset;
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return pfail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return pfail("hipSetDevice failed");
    const int n_cus = policy_n_cus(device_id);
    hipError_t err = n_cus ? hipSuccess : hipErrorInvalidDevice;
    if (err == hipSuccess) {
        const size_t sh = (size_t)HEAD_TILE * HEAD_XS_ROW;  // 66 KB: the stage a/b image; the stage c/d image is smaller
        static_assert(HEAD_TILE * HEAD_XS_ROW >= HEAD_TILE * HEAD_HS_ROW, "hs aliases xs");
        err = hipFuncSetAttribute((const void*)k_policy_head, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) {
            const int64_t tiles = (n_samples + HEAD_TILE - 1) / HEAD_TILE;
            const int64_t cap = (int64_t)n_cus * 2;
            hipLaunchKernelGGL(k_policy_head, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(256), sh, (hipStream_t)stream, a);
            err = hipGetLastError();
        }
    }
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err != hipSuccess) return pfail(hipGetErrorString(err));
    return 0;
}
