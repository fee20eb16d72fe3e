// ctf_policy_tail.hip — the weight / bias gradients of the network's small dense layers (fc2: 256 -> 128, the action and value heads:
// 128 -> 9 + 1; agent_network.py:16-18) in the learner's backward (ppo.py:231-233 -> autograd).
//
//     dW[n][k] = sum over the M samples of dy[m][n] * x[m][k]        db[n] = sum over m of dy[m][n]
//
// are reductions over a quarter of a million to a million samples into a few thousand numbers.  The BLAS library runs them as GEMMs with
// a tiny output and an enormous K: 0.75 ms for fc2's (0.017 TFLOP), 0.25 + 0.18 ms for the heads', and the bias gradients as column
// reductions of narrow matrices (0.13 ms for a [262 144 x 9] sum) — 1.9 of the 8.6 ms of a 262 144-sample pass.  Here: one kernel per
// layer, HBM-bound (it reads dy and x once).  The contraction runs over SAMPLES, so both MFMA operands must be sample-contiguous per
// lane — transposed against the row-major tensors; the tiles are staged in LDS as they are and read through ds_read_b64_tr_b16 (the
// weight-gradient kernels of the conv front do the same over positions, ctf_policy.hip).
#include "ctf_policy_dev.h"

typedef short tail_i16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t tail_tr_operand(const uint8_t* lds_addr, int second_block_bytes) {
    typedef __attribute__((address_space(3))) tail_i16x4_t* lds_v4;
    const tail_i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_addr));
    const tail_i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_addr + second_block_bytes));
    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
    return (u32x4_t){l2[0], l2[1], h2[0], h2[1]};
}

struct TailWgradArgs {
    const uint16_t* dy;   // bf16 [M][N]
    const uint16_t* x;    // bf16 [M][x_stride]: the kernel's K columns start at column k_base + slab * K
    float* dw;            // float [N][dw_stride] += (same column offset)
    float* db;            // float [N] +=, or NULL
    int64_t M;
    int32_t x_stride, dw_stride, k_base, n_slabs;  // n_slabs > 1 (fc1): block b works on column slab b % n_slabs, sample range b / n_slabs
};

#define TAIL_CH 64     // samples per chunk (two K-steps of 32)
#define TAIL_PAD 32    // bytes of padding per staged row (the four rows of a transposed block then fall on distinct banks)

// NT n-tiles x KT k-tiles of 16 x 16.  NT >= 4: wave w owns n-tiles w * NT/4 .. and every k-tile; else every n-tile and k-tiles w * KT/4 ..
template <int NT, int KT>
__global__ void __launch_bounds__(256) k_tail_wgrad(TailWgradArgs a) {
    constexpr int N = 16 * NT, K = 16 * KT;
    constexpr bool NSPLIT = NT >= 4;
    constexpr int WN = NSPLIT ? NT / 4 : NT, WK = NSPLIT ? KT : KT / 4;   // tiles of one wave
    static_assert((NSPLIT ? NT : KT) % 4 == 0, "the split dimension is a multiple of four tiles");
    constexpr int RA = N * 2 + TAIL_PAD, RB = K * 2 + TAIL_PAD;           // staged row strides
    constexpr int PA = N / 8, PB = K / 8;                                  // 16-byte pieces per row
    extern __shared__ __attribute__((aligned(16))) uint8_t tail_lds[];
    uint8_t* A = tail_lds;                    // [TAIL_CH][RA]: dy rows
    uint8_t* B = tail_lds + TAIL_CH * RA;     // [TAIL_CH][RB]: x rows
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int n0 = NSPLIT ? wave * WN : 0, k0 = NSPLIT ? 0 : wave * WK;
    f32x4_t acc[WN][WK];
#pragma unroll
    for (int i = 0; i < WN; i++)
#pragma unroll
        for (int j = 0; j < WK; j++) acc[i][j] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
    // staging: thread t takes pieces t, t + 256, ... of a chunk's dy (TAIL_CH * PA pieces) and x (TAIL_CH * PB pieces)
    constexpr int IA = (TAIL_CH * PA + 255) / 256, IB = (TAIL_CH * PB + 255) / 256;
    static_assert(256 % PA == 0, "a thread always stages the same eight columns of dy (its bias-gradient partial sums)");
    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int kg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int ra = (8 * kg + tq) * RA + tp * 8, rb = (8 * kg + tq) * RB + tp * 8;
    const int64_t n_chunks = (a.M + TAIL_CH - 1) / TAIL_CH;
    const int slab = blockIdx.x % a.n_slabs, c_first = blockIdx.x / a.n_slabs, c_stride = gridDim.x / a.n_slabs;
    const int kcol = a.k_base + slab * K;  // first column of x / dw this block works on
    u32x4_t pa[IA], pb[IB];
    auto fetch = [&](int64_t c) {
        const int64_t m0 = c * TAIL_CH;
#pragma unroll
        for (int i = 0; i < IA; i++) {
            const int idx = tid + 256 * i, row = idx / PA, oct = idx - row * PA;
            const int64_t m = m0 + row;
            pa[i] = (idx < TAIL_CH * PA && m < a.M) ? *(const u32x4_t*)(a.dy + (size_t)m * N + oct * 8) : (u32x4_t){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < IB; i++) {
            const int idx = tid + 256 * i, row = idx / PB, oct = idx - row * PB;
            const int64_t m = m0 + row;
            pb[i] = (idx < TAIL_CH * PB && m < a.M) ? *(const u32x4_t*)(a.x + (size_t)m * a.x_stride + kcol + oct * 8) : (u32x4_t){0u, 0u, 0u, 0u};
        }
    };
    int64_t c = c_first;
    if (c < n_chunks) fetch(c);
    for (; c < n_chunks; c += c_stride) {
#pragma unroll
        for (int i = 0; i < IA; i++) {
            const int idx = tid + 256 * i, row = idx / PA, oct = idx - row * PA;
            if (idx < TAIL_CH * PA) {
                *(u32x4_t*)(A + row * RA + oct * 16) = pa[i];
                if (a.db) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        bsum[2 * j] += __uint_as_float(pa[i][j] << 16);
                        bsum[2 * j + 1] += __uint_as_float(pa[i][j] & 0xFFFF0000u);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < IB; i++) {
            const int idx = tid + 256 * i, row = idx / PB, oct = idx - row * PB;
            if (idx < TAIL_CH * PB) *(u32x4_t*)(B + row * RB + oct * 16) = pb[i];
        }
        if (c + c_stride < n_chunks) fetch(c + c_stride);  // the next chunk travels while this one is contracted
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < TAIL_CH / 32; ks++) {
            u32x4_t av[WN];
#pragma unroll
            for (int i = 0; i < WN; i++) av[i] = tail_tr_operand(A + ra + ks * 32 * RA + (n0 + i) * 32, 4 * RA);
#pragma unroll
            for (int j = 0; j < WK; j++) {
                const u32x4_t bv = tail_tr_operand(B + rb + ks * 32 * RB + (k0 + j) * 32, 4 * RB);
#pragma unroll
                for (int i = 0; i < WN; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[i]), as_bf16x8(bv), acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();  // the next chunk overwrites the tiles
    }
    // D tile: lane holds rows (n) 4 (lane >> 4) + r of column (k) lane & 15
#pragma unroll
    for (int i = 0; i < WN; i++)
#pragma unroll
        for (int j = 0; j < WK; j++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                atomicAdd(a.dw + (size_t)(16 * (n0 + i) + 4 * (lane >> 4) + r) * a.dw_stride + kcol + 16 * (k0 + j) + (lane & 15), acc[i][j][r]);
    if (a.db) {  // a thread's eight columns: octet tid % PA; the block's partial sums meet in LDS, one atomic per column and block
        float* red = (float*)tail_lds;
        __syncthreads();
        for (int q = tid; q < N; q += 256) red[q] = 0.0f;
        __syncthreads();
        const int oct = tid % PA;
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&red[oct * 8 + j], bsum[j]);
        __syncthreads();
        for (int q = tid; q < N; q += 256) atomicAdd(a.db + q, red[q]);
    }
}

extern "C" int ctf_policy_linear_wgrad(const uint16_t* dy_dev, const uint16_t* x_dev, int64_t n_samples, int32_t n_out, int32_t n_in, float* dw_dev,
                                       float* db_dev, int32_t device_id, void* stream) {
    if (!dy_dev || !x_dev || !dw_dev) return ctf_policy_fail("null argument");
    if (n_samples < 0) return ctf_policy_fail("n_samples out of range");
    if (((uintptr_t)dy_dev | (uintptr_t)x_dev) & 15) return ctf_policy_fail("dy / x must be 16-byte aligned");
    if (!n_samples) return 0;
    const int n_cus = ctf_policy_cus(device_id);
    if (!n_cus) return ctf_policy_fail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return ctf_policy_fail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return ctf_policy_fail("hipSetDevice failed");
    TailWgradArgs a;
    a.dy = dy_dev; a.x = x_dev; a.dw = dw_dev; a.db = db_dev; a.M = n_samples;
    a.x_stride = n_in; a.dw_stride = n_in; a.k_base = 0; a.n_slabs = 1;
    const int64_t n_chunks = (n_samples + TAIL_CH - 1) / TAIL_CH;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipSuccess;
    auto launch = [&](auto kernel, int n, int k, int64_t blocks) {
        const size_t sh = (size_t)TAIL_CH * (n * 2 + TAIL_PAD) + (size_t)TAIL_CH * (k * 2 + TAIL_PAD);
        if (err == hipSuccess && sh > 48 * 1024) err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), sh, st, a);
    };
    auto capped = [&](int per_cu) { return n_chunks < (int64_t)n_cus * per_cu ? n_chunks : (int64_t)n_cus * per_cu; };
    if (n_out == 128 && n_in == 256) launch(k_tail_wgrad<8, 16>, 128, 256, capped(2));       // fc2
    else if (n_out == 16 && n_in == 128) launch(k_tail_wgrad<1, 8>, 16, 128, capped(4));      // the two heads, padded to 16 outputs
    else if (n_out == 256 && n_in % 64 == 0 && n_in >= 128 && db_dev == nullptr) {
        // fc1: every block takes all 256 outputs of a SLAB of 128 input columns over a range of the samples — x (the big operand, 8 KB per
        // sample) is read exactly once, dy (512 B per sample) once per slab, by blocks that run side by side (slab = block % n_slabs);
        // a remainder of 64 columns gets a launch of its own
        const int slabs = n_in / 128;
        int64_t ranges = ((int64_t)n_cus * 2 + slabs - 1) / slabs;
        if (ranges > n_chunks) ranges = n_chunks;
        if (ranges < 1) ranges = 1;
        a.n_slabs = slabs;
        launch(k_tail_wgrad<16, 8>, 256, 128, ranges * slabs);
        if (n_in % 128) {
            a.k_base = slabs * 128; a.n_slabs = 1;
            launch(k_tail_wgrad<16, 4>, 256, 64, capped(1) < 64 ? capped(1) : 64);
        }
    } else err = hipErrorInvalidValue;
    if (err == hipSuccess) err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err == hipErrorInvalidValue) return ctf_policy_fail("ctf_policy_linear_wgrad is built for (n_out, n_in) = (128, 256), (16, 128) and (256, a multiple of 64 without bias)");
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the rollout collector's per-step bookkeeping (PPOTrainer.get_single_rollout, ppo.py:74-93: what is stored per trained agent and the
// joint action handed to env.step) as ONE launch
// ------------------------------------------------------------------------------------------------
// Per env step the collector stores, for every trained agent, the observation the policy saw (code bytes, metadata), its action,
// log-prob and value, and assembles the env's joint action with team-1 agents' actions mapped back through the flip (REVERSED_ACTION_MAP,
// ppo.py:80-83,90-93).  As tensor expressions that is ~17 small kernels per step (0.15 ms of a 2.1 ms step at 65 536 envs); here one wave
// per (trained agent, env) row copies the row's 225 code bytes and converts its metadata, and the waves of agent slot 0 also write their
// env's joint action.
struct RolloutStoreArgs {
    const uint8_t* codes;        // u8 [E][N][cells]
    const uint16_t* meta;        // f16 [E][N][M]
    const int32_t* act_trained;  // i32 [A][E]
    const float* logprob;        // f32 [A][E]
    const float* value;          // f32 [A][E]
    const int32_t* act_other;    // i32 [B][E]
    uint8_t* grid_codes;         // u8 [A][E][cells]
    float* metadata;             // f32 [A][E][M]
    float* actions;              // f32 [A][E]
    float* logprobs;             // f32 [A][E]
    float* values;               // f32 [A][E]
    int8_t* env_actions;         // i8 [E][N]
    int32_t E, N, cells, M, A;
    uint64_t sel_pack;           // nibble k: the agent of trained slot k
    uint64_t src_pack;           // nibble n: where agent n's action comes from: bit 3 = the other list, bits 0..2 = its slot
    uint64_t lut_pack;           // nibble a: the env action a team-1 agent's sampled action a stands for
    uint32_t team1_mask;         // bit n: agent n's action goes through the LUT
};

__global__ void __launch_bounds__(256) k_rollout_store(RolloutStoreArgs a) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t rows = (int64_t)a.A * a.E;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int k = (int)(row / a.E), e = (int)(row - (int64_t)k * a.E);
        const int agent = (int)((a.sel_pack >> (4 * k)) & 15u);
        const uint8_t* src = a.codes + ((size_t)e * a.N + agent) * a.cells;
        uint8_t* dst = a.grid_codes + (size_t)row * a.cells;
        for (int c = lane; c < a.cells; c += WAVE) dst[c] = src[c];
        if (lane < a.M)
            a.metadata[(size_t)row * a.M + lane] = (float)__builtin_bit_cast(_Float16, a.meta[((size_t)e * a.N + agent) * a.M + lane]);
        if (lane == 0) {
            a.actions[row] = (float)a.act_trained[row];
            a.logprobs[row] = a.logprob[row];
            a.values[row] = a.value[row];
        }
        if (k == 0 && lane < a.N) {  // the env's joint action
            const uint32_t s = (uint32_t)((a.src_pack >> (4 * lane)) & 15u);
            int act = (s & 8u) ? a.act_other[(size_t)(s & 7u) * a.E + e] : a.act_trained[(size_t)(s & 7u) * a.E + e];
            if ((a.team1_mask >> lane) & 1u) act = (int)((a.lut_pack >> (4 * (act & 15))) & 15u);
            a.env_actions[(size_t)e * a.N + lane] = (int8_t)act;
        }
    }
}

extern "C" int ctf_rollout_store_step(const uint8_t* codes_dev, const uint16_t* meta_dev, int32_t n_envs, int32_t n_agents, int32_t cells,
                                      int32_t meta_len, const int32_t* trained_sel, int32_t n_trained, const int32_t* other_sel, int32_t n_other,
                                      const int32_t* act_trained_dev, const float* logprob_dev, const float* value_dev, const int32_t* act_other_dev,
                                      const uint8_t* reversed_action_lut, uint32_t team1_mask, uint8_t* grid_codes_out, float* metadata_out,
                                      float* actions_out, float* logprobs_out, float* values_out, int8_t* env_actions_out, int32_t device_id,
                                      void* stream) {
    if (!codes_dev || !meta_dev || !trained_sel || !other_sel || !act_trained_dev || !logprob_dev || !value_dev || !act_other_dev ||
        !reversed_action_lut || !grid_codes_out || !metadata_out || !actions_out || !logprobs_out || !values_out || !env_actions_out)
        return ctf_policy_fail("null argument");
    if (n_envs < 1 || n_agents < 1 || n_agents > 16 || n_trained < 1 || n_trained > 8 || n_other < 0 || n_other > 8 || n_trained + n_other != n_agents)
        return ctf_policy_fail("n_envs / n_agents / selections out of range (every agent is in exactly one of the two lists, at most 8 each)");
    if (cells < 1 || meta_len < 1 || meta_len > 64) return ctf_policy_fail("cells / meta_len out of range");
    RolloutStoreArgs a;
    a.codes = codes_dev; a.meta = meta_dev; a.act_trained = act_trained_dev; a.logprob = logprob_dev; a.value = value_dev; a.act_other = act_other_dev;
    a.grid_codes = grid_codes_out; a.metadata = metadata_out; a.actions = actions_out; a.logprobs = logprobs_out; a.values = values_out;
    a.env_actions = env_actions_out; a.E = n_envs; a.N = n_agents; a.cells = cells; a.M = meta_len; a.A = n_trained; a.team1_mask = team1_mask;
    a.sel_pack = 0; a.src_pack = 0; a.lut_pack = 0;
    uint32_t seen = 0;
    for (int k = 0; k < n_trained; k++) {
        if (trained_sel[k] < 0 || trained_sel[k] >= n_agents || ((seen >> trained_sel[k]) & 1u)) return ctf_policy_fail("trained_sel entry out of range or repeated");
        seen |= 1u << trained_sel[k];
        a.sel_pack |= (uint64_t)trained_sel[k] << (4 * k);
        a.src_pack |= (uint64_t)k << (4 * trained_sel[k]);
    }
    for (int k = 0; k < n_other; k++) {
        if (other_sel[k] < 0 || other_sel[k] >= n_agents || ((seen >> other_sel[k]) & 1u)) return ctf_policy_fail("other_sel entry out of range or repeated");
        seen |= 1u << other_sel[k];
        a.src_pack |= (uint64_t)(8 | k) << (4 * other_sel[k]);
    }
    for (int i = 0; i < 9; i++) {
        if (reversed_action_lut[i] > 8) return ctf_policy_fail("reversed_action_lut entry out of range");
        a.lut_pack |= (uint64_t)reversed_action_lut[i] << (4 * i);
    }
    const int n_cus = ctf_policy_cus(device_id);
    if (!n_cus) return ctf_policy_fail("hipGetDeviceProperties failed");
    int dev_prev = 0;
    if (hipGetDevice(&dev_prev) != hipSuccess) return ctf_policy_fail("hipGetDevice failed");
    if (dev_prev != device_id && hipSetDevice(device_id) != hipSuccess) return ctf_policy_fail("hipSetDevice failed");
    int64_t blocks = ((int64_t)n_trained * n_envs + 3) / 4;
    if (blocks > (int64_t)n_cus * 16) blocks = (int64_t)n_cus * 16;
    hipLaunchKernelGGL(k_rollout_store, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    const hipError_t err = hipGetLastError();
    if (dev_prev != device_id) (void)hipSetDevice(dev_prev);
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}
