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
    float* part;          // deterministic mode: float [blocks][part_stride], every block's own tile [N][K] then its bias sums [N]; NULL: atomics
    int64_t part_stride;
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
    float* part = a.part ? a.part + (size_t)blockIdx.x * a.part_stride : nullptr;
#pragma unroll
    for (int i = 0; i < WN; i++)
#pragma unroll
        for (int j = 0; j < WK; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = 16 * (n0 + i) + 4 * (lane >> 4) + r, k = 16 * (k0 + j) + (lane & 15);
                if (part) part[n * K + k] = acc[i][j][r];
                else atomicAdd(a.dw + (size_t)n * a.dw_stride + kcol + k, acc[i][j][r]);
            }
    if (a.db) {  // a thread's eight columns: octet tid % PA.  The block's 256 / PA partial sums per column meet in LDS and are added in
                 // thread order (a fixed order: the block's bias sum does not depend on how its waves were scheduled)
        float* red = (float*)tail_lds;  // [256][8]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) red[tid * 8 + j] = bsum[j];
        __syncthreads();
        for (int q = tid; q < N; q += 256) {
            float t = 0.0f;
            for (int u = q >> 3; u < 256; u += PA) t += red[u * 8 + (q & 7)];
            if (part) part[N * K + q] = t;
            else atomicAdd(a.db + q, t);
        }
    }
}

// deterministic mode: dw[n][k_base + slab * K + k] += the blocks' tiles of that slab, sample ranges in ascending order
__global__ void __launch_bounds__(256) k_tail_det_reduce(const float* part, int64_t stride, int n_ranges, int n_slabs, int N, int K, float* dw,
                                                         int dw_stride, int k_base) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_slabs * N * K) return;
    const int slab = idx / (N * K), rem = idx - slab * (N * K), n = rem / K, k = rem - n * K;
    const float* p = part + (size_t)slab * stride + rem;
    const size_t step = (size_t)n_slabs * stride;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int r = 0;
    for (; r + 3 < n_ranges; r += 4) {
        s0 += p[(size_t)r * step];
        s1 += p[(size_t)(r + 1) * step];
        s2 += p[(size_t)(r + 2) * step];
        s3 += p[(size_t)(r + 3) * step];
    }
    for (; r < n_ranges; r++) s0 += p[(size_t)r * step];
    dw[(size_t)n * dw_stride + k_base + slab * K + k] += (s0 + s1) + (s2 + s3);
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
    const DetWorkspace det = ctf_policy_det(device_id);
    bool too_small = false;
    auto launch = [&](auto kernel, int n, int k, int64_t blocks) {
        const size_t sh = (size_t)TAIL_CH * (n * 2 + TAIL_PAD) + (size_t)TAIL_CH * (k * 2 + TAIL_PAD);
        a.part = det.ptr;
        a.part_stride = (int64_t)n * k + n;
        if (a.part && blocks * a.part_stride > det.floats) { too_small = true; return; }
        if (err == hipSuccess && sh > 48 * 1024) err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (err == hipSuccess) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), sh, st, a);
        if (err == hipSuccess && a.part) {  // the blocks' tiles -> dw (and db), in block order
            const int ranges = (int)(blocks / a.n_slabs);
            hipLaunchKernelGGL(k_tail_det_reduce, dim3((unsigned)((a.n_slabs * n * k + 255) / 256)), dim3(256), 0, st, (const float*)a.part, a.part_stride,
                               ranges, a.n_slabs, n, k, a.dw, a.dw_stride, a.k_base);
            if (a.db) err = ctf_policy_det_reduce(a.part + (size_t)n * k, (int)blocks, a.part_stride, n, a.db, st);
        }
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
    if (too_small) return ctf_policy_fail("deterministic mode: the registered workspace is too small for this launch (ctf_policy_set_deterministic)");
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

// ------------------------------------------------------------------------------------------------
// fc1's data gradient: d_act[M][Kp] = dy[M][256] x W[256][Kp]   (ppo.py:231-233 -> autograd through agent_network.py:16)
// ------------------------------------------------------------------------------------------------
// 2.18 GB of bf16 written per 262 144 samples against 0.56 TFLOP: HBM-write bound; the library's GEMM takes 1.06-1.08 ms
// (profiles/r04_learner_roofline.md), this kernel 0.76 (same bits).  The short contraction (256) lets a wave keep ITS 32 rows of dy in
// registers for the whole kernel (16 B-operand fragments); a block of 8 waves = 256 rows walks over the Kp columns in chunks of 64:
// the chunk of W^T ([64 columns][256 k], 32 KB, L2-resident) comes global -> LDS by DMA into a ring of three stages, two chunks ahead —
//     chunk c:   s_waitcnt vmcnt(12)  this wave's 4 pieces of chunk c have landed (younger: 4 stores, the 4 pieces of c + 1, 4 stores)
//                s_barrier            everybody's have, and everybody has finished reading chunk c - 1
//                4 x glds             chunk c + 2 -> the stage chunk c - 1 was read from
//                32 ds_read_b128 + 32 MFMA (32x32x16; W^T rows as the A operand: D[column][row])
//                the wave's [32 rows][64 columns] through its own LDS patch -> four stores of whole 128-byte lines per row
// (loads, stores and LDS-DMA are counted together per wave, in issue order, which is what makes the 12 exact).  An LDS-DMA piece is
// 1 KiB = two W^T rows of 512 B; slot s of row r holds the row's 16-byte piece s ^ (r & 15), so that the 16 rows of a ds_read_b128
// lane group fall on the 16 slots of a 256-byte bank row.
// Measured around it (profiles/r04_fc1_dgrad.md): the stores alone 0.59 ms (3.7 TB/s: 128-byte pieces at a row stride of 8 320 B), the
// arithmetic alone 0.54; fragments prefetched four k-steps ahead, tiles mapped XCD-contiguously, the next chunk's DMA ahead of the
// stores, and two loader waves of their own (compute waves that never wait for vector memory) all landed within 0.76-0.82.
struct Fc1DgradArgs {
    const uint16_t* dy;   // bf16 [M][256]
    const uint16_t* wt;   // bf16 [Kp][256]: W^T
    uint16_t* out;        // bf16 [M][Kp]
    int32_t M, Kp;
};
#define FD_STAGE 32768
#define FD_OROW 144      // bytes of a row of the wave's output patch: 64 columns + 16 of padding
__device__ __forceinline__ void fd_glds16(const uint8_t* gsrc, uint32_t lds_dst) {
    uint32_t keep;   // (hand-placed: the loop's own counted s_waitcnt covers it; M0 is put back)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_fc1_dgrad(Fc1DgradArgs g) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t fd_lds[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int n32 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * 256 + 32 * wave;       // this wave's rows
    const int n_chunks = g.Kp >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)fd_lds;
    uint8_t* const opatch = fd_lds + 3 * FD_STAGE + wave * 32 * FD_OROW;
    // piece i of this wave = W^T rows 2 (4 wave + i), + 1 of the chunk: lane -> row + (lane >> 5), slot lane & 31
    const uint8_t* src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = 2 * (4 * wave + i) + (lane >> 5);
        src[i] = (const uint8_t*)g.wt + (size_t)r * 512 + (((lane & 31) ^ (r & 15)) * 16);
    }
    const uint32_t dst0 = lds0 + 4 * wave * 1024;
#define FD_ISSUE(C, ST)  _Pragma("unroll") for (int i = 0; i < 4; i++) fd_glds16(src[i] + (size_t)(C) * (64 * 512), dst0 + (ST) * FD_STAGE + i * 1024)
    FD_ISSUE(0, 0);
    FD_ISSUE(min(1, n_chunks - 1), 1);
    // the wave's dy rows as B-operand fragments: lane -> row n32, k = 16 u + 8 hh ..
    u32x4_t dyf[16];
    {
        const uint8_t* dyrow = (const uint8_t*)g.dy + (size_t)min(row0 + n32, g.M - 1) * 512 + hh * 16;
#pragma unroll
        for (int u = 0; u < 16; u++) dyf[u] = *(const u32x4_t*)(dyrow + u * 32);
    }
    // stores of the patch: pass t -> patch row (lane >> 3) + 8 t, 16-byte piece lane & 7
    const int orow = lane >> 3, opc = lane & 7;
    uint8_t* const gout = (uint8_t*)g.out + (size_t)(row0 + orow) * g.Kp * 2 + opc * 16;
    const size_t gstep = (size_t)8 * g.Kp * 2;
    const uint8_t* const wrow = fd_lds + n32 * 512;
    const int sw = n32 & 15;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the dy fragments; the two first chunks too — once)
    const bool live = row0 < g.M;   // (M is a multiple of 32: a wave's rows exist or do not; a wave without rows still loads its pieces of W^T)
    int st = 0, st2 = 2;
#pragma unroll 1
    for (int c = 0; c < n_chunks; c++) {
        if (live) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // (c = 0: nothing is in flight)
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // no stores in between
        asm volatile("s_barrier" ::: "memory");
        FD_ISSUE(min(c + 2, n_chunks - 1), st2);   // (past the end: the last chunk once more, into a stage nobody reads any longer)
        const uint8_t* W = wrow + st * FD_STAGE;
        f32x16_t acc[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const u32x4_t wf = *(const u32x4_t*)(W + i * (32 * 512) + (((2 * u + hh) ^ sw) * 16));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf), as_bf16x8(dyf[u]), acc[i], 0, 0, 0);
            }
        }
        // D[column 32 i + 8 q + 4 hh + r][row n32] -> the patch [row][64 columns]
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32x2_t v;
                v[0] = pack_bf16(acc[i][4 * q], acc[i][4 * q + 1]);
                v[1] = pack_bf16(acc[i][4 * q + 2], acc[i][4 * q + 3]);
                *(u32x2_t*)(opatch + n32 * FD_OROW + (32 * i + 8 * q + 4 * hh) * 2) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        u32x4_t o[4];
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = *(const u32x4_t*)(opatch + (orow + 8 * t) * FD_OROW + opc * 16);
        if (live) {   // exactly four store instructions a chunk: the counted wait above
#pragma unroll
            for (int t = 0; t < 4; t++) *(u32x4_t*)(gout + t * gstep + (size_t)c * 128) = o[t];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is read before the next chunk overwrites it
        __builtin_amdgcn_wave_barrier();
        st = st == 2 ? 0 : st + 1;
        st2 = st2 == 2 ? 0 : st2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two refetches past the end: landed before the wave may go
#undef FD_ISSUE
}

extern "C" int ctf_policy_fc1_dgrad(const uint16_t* dy_dev, const uint16_t* wt_dev, int32_t n_samples, int32_t kp, uint16_t* d_act_dev,
                                    int32_t device_id, void* stream) {
    if (!dy_dev || !wt_dev || !d_act_dev) return ctf_policy_fail("null argument");
    if (n_samples < 32 || (n_samples & 31)) return ctf_policy_fail("n_samples must be a positive multiple of 32");
    if (kp < 64 || (kp & 63)) return ctf_policy_fail("kp must be a multiple of 64");
    if (((uintptr_t)dy_dev | (uintptr_t)wt_dev | (uintptr_t)d_act_dev) & 15) return ctf_policy_fail("16-byte alignment");
    Fc1DgradArgs g;
    g.dy = dy_dev; g.wt = wt_dev; g.out = d_act_dev; g.M = n_samples; g.Kp = kp;
    DeviceScope scope(device_id);
    if (!scope.ok) return ctf_policy_fail("hipSetDevice failed");
    const int sh = 3 * FD_STAGE + 8 * 32 * FD_OROW;
    hipError_t err = hipFuncSetAttribute((const void*)k_fc1_dgrad, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
    if (err == hipSuccess) hipLaunchKernelGGL(k_fc1_dgrad, dim3((n_samples + 255) / 256), dim3(512), sh, (hipStream_t)stream, g);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) return ctf_policy_fail(hipGetErrorString(err));
    return 0;
}
