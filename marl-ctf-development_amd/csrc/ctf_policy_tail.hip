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
