// ctf_kernels.hip — hand-written gfx950 (CDNA4) kernels of the batched GridworldCtf hot path.
//
//   k_seed            twin MT19937 seeding per env (CPython init_by_array / NumPy init_genrand)
//   k_reset           GridworldCtf.reset()                       (reference gridworld_ctf.py:383-477)
//   k_step            GridworldCtf.step(actions)                 (reference gridworld_ctf.py:849-918)
//   k_observe         standardise_state + get_env_metadata, all agents (gridworld_ctf.py:975-1069)
//   k_random_actions  synthetic Philox4x32-10 action stream for bench / tests
//
// Integer / byte work, HBM-bound: no MFMA anywhere.  Wavefront = 64 lanes is assumed throughout.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ctf_device.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return (uint32_t)(((uint64_t)n * d.m) >> d.s); }
__device__ __forceinline__ int iabs_(int x) { return x < 0 ? -x : x; }
__device__ __forceinline__ int cheb(int r0, int c0, int r1, int c1) {
    int a = iabs_(r0 - r1), b = iabs_(c0 - c1);
    return a > b ? a : b;
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// MT19937 with LAZY in-place regeneration (used by MtWin below): instead of rewriting all 624 words when
// the block is exhausted (a 624-iteration burst that would serialise a divergent wave), word i of the
// next block is produced from a[i], a[i+1], a[i+397] at the moment it is consumed.  Words [0,pos) then
// belong to the new block and [pos,624) to the old one, which is exactly the order the standard in-place
// algorithm visits them in, so the output stream is identical.  The lazy flag is 0 only between a state
// import (words [pos,624) are output as they stand) and the first wrap.

// NumPy npy_double_to_half: direct round-to-nearest-even f64 -> binary16 bits
__device__ __forceinline__ uint16_t f64_to_f16(double d) {
    uint64_t b = (uint64_t)__double_as_longlong(d);
    uint32_t sign = (uint32_t)((b >> 48) & 0x8000u);
    int32_t be = (int32_t)((b >> 52) & 0x7FF);
    uint64_t m = b & 0xFFFFFFFFFFFFFull;
    if (be == 0x7FF) return (uint16_t)(sign | 0x7C00u | (m ? (0x200u | (uint32_t)(m >> 42)) : 0u));
    if (be == 0) return (uint16_t)sign;
    int32_t E = be - 1023;
    if (E > 15) return (uint16_t)(sign | 0x7C00u);
    if (E >= -14) {
        uint32_t h = (uint32_t)((E + 15) << 10) | (uint32_t)(m >> 42);
        uint64_t rem = m & ((1ull << 42) - 1), half = 1ull << 41;
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    if (E < -25) return (uint16_t)sign;
    uint64_t full = m | (1ull << 52);
    int shift = 28 - E;
    uint64_t h = full >> shift, rem = full & ((1ull << shift) - 1), half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | (uint32_t)h);
}

// ------------------------------------------------------------------------------------------------
// seeding
// ------------------------------------------------------------------------------------------------
__device__ void mt_init_genrand(uint32_t* mt, uint32_t s) {
    mt[0] = s;
    uint32_t prev = s;
    for (int i = 1; i < CTF_MT_N; i++) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
        mt[i] = prev;
    }
}
__device__ void mt_init_by_array(uint32_t* mt, const uint32_t* key, int len) {
    mt_init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    uint32_t prev = mt[0];
    for (int k = CTF_MT_N > len ? CTF_MT_N : len; k; k--) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        mt[i] = prev;
        i++; j++;
        if (i >= CTF_MT_N) { mt[0] = prev; i = 1; }
        if (j >= len) j = 0;
    }
    for (int k = CTF_MT_N - 1; k; k--) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
        mt[i] = prev;
        i++;
        if (i >= CTF_MT_N) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// py_seeds / np_seeds: device arrays [E].  After this, env e == random.seed(py) ; np.random.seed(np).
extern "C" __global__ void k_seed(DevCfg cfg, DevPtrs p, const uint64_t* py_seeds, const uint64_t* np_seeds) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.n_envs) return;
    uint64_t ps = py_seeds[e];
    uint32_t key[2] = {(uint32_t)ps, (uint32_t)(ps >> 32)};
    mt_init_by_array(p.mt_py + (size_t)e * CTF_MT_N, key, key[1] ? 2 : 1);
    mt_init_genrand(p.mt_np + (size_t)e * CTF_MT_N, (uint32_t)np_seeds[e]);
    p.rngpos[2 * e + 0] = CTF_MT_N;  // both generators start exhausted: first draw regenerates
    p.rngpos[2 * e + 1] = CTF_MT_N;
}

// ------------------------------------------------------------------------------------------------
// reset
// ------------------------------------------------------------------------------------------------
// Writes the reset record of one env at `sr` (any address space) — everything except `perm`.
template <typename BytePtr>
__device__ __forceinline__ void reset_record(const DevCfg& cfg, BytePtr sr) {
    for (int i = 0; i < cfg.N; i++) {
        uint64_t hb = (uint64_t)__double_as_longlong(cfg.type_hp[cfg.type[i]]);
        ((uint32_t*)(sr))[2 * i] = (uint32_t)hb;
        ((uint32_t*)(sr))[2 * i + 1] = (uint32_t)(hb >> 32);
        sr[cfg.off_pos + 2 * i] = (uint8_t)cfg.start_pos[i][0];
        sr[cfg.off_pos + 2 * i + 1] = (uint8_t)cfg.start_pos[i][1];
        sr[cfg.off_flag + i] = 0;
        *(uint16_t*)(sr + cfg.off_inv + 2 * i) = 0;
    }
    int32_t* misc = (int32_t*)(sr + cfg.off_misc);
    misc[0] = 0;  // env_step_count
    misc[1] = 0;  // team_flag_captures[0]
    misc[2] = 0;  // team_flag_captures[1]
    misc[3] = CTF_F_BASE_ZERO;  // not done; visitation = zero maps + the start cells (:473), log empty
}

// One 64-lane block per env; mask == nullptr resets every env.  init_perm is set only by ctf_create.
extern "C" __global__ void __launch_bounds__(WAVE) k_reset(DevCfg cfg, DevPtrs p, const uint8_t* mask, int init_perm) {
    int e = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[e]) return;
    uint32_t* g = (uint32_t*)(p.grid + (size_t)e * cfg.GS);
    const uint32_t* src = (const uint32_t*)p.init_grid;
    for (int w = lane; w < cfg.GS / 4; w += WAVE) g[w] = src[w];
    uint8_t* sr = p.rec + (size_t)e * cfg.RS;
    if (lane == 0) {
        reset_record(cfg, sr);
        if (init_perm)
            for (int i = 0; i < cfg.N; i++) sr[cfg.off_perm + i] = (uint8_t)i;
    }
    if (cfg.log_metrics) {
        int32_t* m = p.metrics + (size_t)e * CTF_N_METRICS * cfg.N;
        for (int w = lane; w < CTF_N_METRICS * cfg.N; w += WAVE) m[w] = 0;
        // visitation: reset_record flagged the base maps as zero and emptied the log — nothing to clear
    }
}

// ------------------------------------------------------------------------------------------------
// lane-divergent config lookups: bit-field extracts / selects on SGPR-resident values (no memory traffic)
// ------------------------------------------------------------------------------------------------
// pin*(): pass a kernel-argument value through an empty asm with an SGPR constraint so that it is an opaque SGPR value.  Without
// this the compiler rewrites "team ? cfg.x[1] : cfg.x[0]" into ONE load from a lane-selected kernarg
// address, i.e. a dependent vector-memory access in the middle of the per-agent loop.
__device__ __forceinline__ int pin(int v) {
    asm("" : "+s"(v));  // zero instructions: just makes the value an opaque SGPR operand
    return v;
}
__device__ __forceinline__ uint64_t pin64(uint64_t v) {
    asm("" : "+s"(v));
    return v;
}
__device__ __forceinline__ double pind(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    asm("" : "+s"(b));
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ int cfg_team(const DevCfg& c, int a) { return (int)(((uint32_t)pin((int)c.team_mask) >> a) & 1u); }
__device__ __forceinline__ int cfg_type(const DevCfg& c, int a) { return (int)(((uint32_t)pin((int)c.type_pack) >> (2 * a)) & 3u); }
__device__ __forceinline__ int cfg_opp(const DevCfg& c, int team, int q) {
    const uint64_t p0 = pin64(c.opp_pack[0]), p1 = pin64(c.opp_pack[1]);
    return (int)(((team ? p1 : p0) >> (4 * q)) & 15u);
}
__device__ __forceinline__ int cfg_nopp(const DevCfg& c, int team) { return team ? pin(c.n_opp[1]) : pin(c.n_opp[0]); }
__device__ __forceinline__ double sel4(const double* t, int k) {
    const double t0 = pind(t[0]), t1 = pind(t[1]), t2 = pind(t[2]), t3 = pind(t[3]);
    return k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3));
}
#define TSEL(arr, team, k) ((team) ? pin((int)(arr)[1][k]) : pin((int)(arr)[0][k]))

// ------------------------------------------------------------------------------------------------
// step — W lanes per env (W = 1, 2, 4 or 8 >= opponents per team), 64 / W envs per 64-thread block
// ------------------------------------------------------------------------------------------------
// The per-agent loop of GridworldCtf.step is inherently sequential (later agents see earlier agents' moves, tags and
// respawns), so parallelism is across envs — but one lane per env leaves one wave per SIMD and a kernel bound by the
// dependent LDS / VALU chain.  Here the W lanes of a group run the env's control flow redundantly (state reads are LDS
// broadcasts; state WRITES are done by sub-lane 0 only) and split the work that is parallel inside an agent's turn:
//   - tagging: sub-lane q draws the np.random.rand() of opponent q from its own two window words and tests range;
//     __ballot finds the first hit, which is applied (possibly respawning, which consumes extra words) before the
//     remaining opponents are re-evaluated from the shifted stream position — exactly the reference's draw order;
//   - adjacency / zone metrics, healing, rewards, visitation: one agent per sub-lane;
//   - MT window refills and write-backs: one word per sub-lane per pass.
// 64 / W envs per wave means W times more waves (4 per SIMD for the arena) to hide the latency chain.
//
// LDS slot of one env (bytes):
//   [grid GS][rec RS][actions 16][py window 64][np window 64][metric deltas u8 13*N (METRICS)]
// The slot stride in dwords is odd so that different groups' same-offset accesses fall in distinct banks.
#ifndef WCAP
#define WCAP 16  // MT words per window (>= 2 * max opponents per team)
#endif
// Profiling-only ablations of the step kernel (results become wrong; never defined in the shipped build):
//   bit0 no tagging, bit1 no metric section, bit2 no shuffles, bit3 no act, bit4 no visitation atomics,
//   bit5 no metric flush, bit6 no state write-back, bit7 MT refills without their loads, bit8 MT flushes without their stores
#ifndef STEP_ABLATE
#define STEP_ABLATE 0
#endif
// Profiling-only phase trace of the step kernel (tools/trace_step.py): lane 0 of every block stamps the 100 MHz
// wall clock at fixed points of the step (1: every phase, 2: start / end only); never defined in the shipped build.
#ifndef STEP_TRACE
#define STEP_TRACE 0
#endif
#if STEP_TRACE
__device__ unsigned long long g_step_trace[8192][40];
__device__ unsigned long long g_step_span[8192][3];  // every block: start, end, HW_ID
#define STEP_STAMP(k) do { if (threadIdx.x == 0 && (STEP_TRACE == 1 || (k) == 0 || (k) == 32)) { const unsigned long long t_ = wall_clock64(); \
        if (STEP_TRACE == 1 && blockIdx.x < 8192) g_step_trace[blockIdx.x][(k)] = t_; \
        if ((k) == 0 && blockIdx.x < 8192) { g_step_span[blockIdx.x][0] = t_; g_step_span[blockIdx.x][2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); } \
        if ((k) == 32 && blockIdx.x < 8192) g_step_span[blockIdx.x][1] = t_; } } while (0)
extern "C" int ctf_debug_step_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_trace), sizeof(g_step_trace));
}
extern "C" int ctf_debug_step_span(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_span), sizeof(g_step_span));
}
#else
#define STEP_STAMP(k) do { } while (0)
#endif

__host__ __device__ inline int step_slot_bytes(int GS, int RS, int N, bool metrics) {
    int b = GS + RS + 16 + 2 * WCAP * 4 + (metrics ? ((CTF_N_METRICS * N + 3) & ~3) : 0);
    if (((b / 4) & 1) == 0) b += 4;
    return b;
}

// One MT19937 stream of one env: state words in HBM, a WCAP-word window of the next words in LDS (untempered, already
// regenerated).  All fields are replicated in the registers of the group's W lanes and updated identically.
// Lazy in-place regeneration (see above); a window may span the end of a block: word idx >= 624 is word idx-624 of
// the next block and is always regenerated.
struct MtWin {
    uint32_t* a;    // 624 state words (global)
    uint32_t* win;  // WCAP words (LDS)
    uint32_t pos;   // stream position of win[0], 0..624
    uint32_t lazy;
    uint32_t n, cur;
};
__device__ __forceinline__ MtWin mtw_open(uint32_t* base, uint32_t* win, uint32_t packed) {
    MtWin g;
    g.a = base; g.win = win;
    g.pos = packed & CTF_POS_MASK;
    g.lazy = (packed & CTF_LAZY_BIT) ? 1u : 0u;
    g.n = 0; g.cur = 0;
    return g;
}
// ---- the window's traffic with the state array.  Sub-lane j of the group owns the T = WCAP / W CONSECUTIVE window words
// [T j, T j + T): its state words are one (4-byte aligned) multi-dword access, so that a group reads / writes whole 64-byte
// runs instead of 16-byte pieces — the MT streams are 40 % of the step kernel's time, all of it memory requests.
typedef uint32_t mt_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t mt_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
// words a[i0 .. i0 + T) with indices taken modulo 624 (i0 < 624)
template <int T>
__device__ __forceinline__ void mt_load_span(const uint32_t* a, uint32_t i0, uint32_t* out) {
    if (i0 + T <= CTF_MT_N) {
        if constexpr (T % 4 == 0) {
#pragma unroll
            for (int q = 0; q < T / 4; q++) {
                const mt_u32x4 v = *(const mt_u32x4*)(a + i0 + 4 * q);
                out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
            }
        } else if constexpr (T == 2) {
            const mt_u32x2 v = *(const mt_u32x2*)(a + i0);
            out[0] = v.x; out[1] = v.y;
        } else {
#pragma unroll
            for (int k = 0; k < T; k++) out[k] = a[i0 + k];
        }
    } else {  // the span crosses the end of the block
#pragma unroll
        for (int k = 0; k < T; k++) out[k] = a[i0 + k >= CTF_MT_N ? i0 + k - CTF_MT_N : i0 + k];
    }
}
// stores the consumed, regenerated words of the window (words [0, cur)) back to the state array
template <int W>
__device__ __forceinline__ void mtw_store_consumed(const MtWin& g, int j) {
    constexpr int T = WCAP / W;
    if (STEP_ABLATE & 256) return;  // no flush stores
    const int c = (int)g.cur - T * j;  // how many of this lane's words were consumed
    if (c <= 0) return;
    const uint32_t idx0 = g.pos + (uint32_t)(T * j);
    const bool crossing = idx0 < CTF_MT_N && idx0 + T > CTF_MT_N;
    if (T % 4 == 0 && c >= T && !crossing && (g.lazy || idx0 >= CTF_MT_N)) {
        uint32_t* dst = g.a + (idx0 >= CTF_MT_N ? idx0 - CTF_MT_N : idx0);
#pragma unroll
        for (int q = 0; q < T / 4; q++) {
            const mt_u32x4 v = {g.win[T * j + 4 * q], g.win[T * j + 4 * q + 1], g.win[T * j + 4 * q + 2], g.win[T * j + 4 * q + 3]};
            *(mt_u32x4*)(dst + 4 * q) = v;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < T; k++) {
        const uint32_t idx = idx0 + (uint32_t)k;
        const bool wrapped = idx >= CTF_MT_N;
        if (k < c && (g.lazy | (uint32_t)wrapped)) g.a[wrapped ? idx - CTF_MT_N : idx] = g.win[T * j + k];
    }
}
// write the consumed words back and advance; unconsumed window words are simply dropped — they are recomputed from
// unchanged state words by the next refill
template <int W>
__device__ __forceinline__ void mtw_flush(MtWin& g, int j) {
    mtw_store_consumed<W>(g, j);
    g.pos += g.cur;
    if (g.pos > CTF_MT_N) { g.pos -= CTF_MT_N; g.lazy = 1; }
    g.n = 0; g.cur = 0;
}
// flush + refill in one: the NEW window's loads are issued first, then the consumed words of the old window are stored, then
// the new window is written.  The words loaded are never among the words stored: the loads touch [npos, npos + WCAP], their
// + 397 partners and, beyond 227, words regenerated >= 211 positions ago; the stores [npos - cur, npos).
template <int W>
__device__ __forceinline__ void mtw_cycle(MtWin& g, int j, int gshift) {
    constexpr int T = WCAP / W;
    uint32_t npos = g.pos + g.cur, nlazy = g.lazy;
    if (npos > CTF_MT_N) { npos -= CTF_MT_N; nlazy = 1; }
    const uint32_t idx0 = npos + (uint32_t)(T * j);                       // stream index of this lane's first word
    const uint32_t i0 = idx0 >= CTF_MT_N ? idx0 - CTF_MT_N : idx0;        // ... as a state word
    uint32_t x0[T], x1n, m[T];
    if (STEP_ABLATE & 128) {  // no refill loads
#pragma unroll
        for (int k = 0; k < T; k++) { x0[k] = (i0 + k) * 2654435761u; m[k] = x0[k] >> 3; }
        x1n = x0[0] ^ 0x55u;
    } else {
        mt_load_span<T>(g.a, i0, x0);
        mt_load_span<T>(g.a, i0 + 397 >= CTF_MT_N ? i0 + 397 - CTF_MT_N : i0 + 397, m);
        // a[i + 1] of the lane's last word = the next lane's first word; the group's last lane loads it
        const uint32_t inext = i0 + T >= CTF_MT_N ? i0 + T - CTF_MT_N : i0 + T;
        x1n = (j == W - 1) ? g.a[inext] : 0u;
    }
    mtw_store_consumed<W>(g, j);  // the old window's consumed, regenerated words go back (LDS reads -> global stores)
    if (W > 1) {
        const uint32_t from_next = (uint32_t)__shfl((int)x0[0], gshift + ((j + 1) & (W - 1)), WAVE);
        if (j != W - 1) x1n = from_next;
    }
#pragma unroll
    for (int k = 0; k < T; k++) {
        const uint32_t idx = idx0 + (uint32_t)k;
        const bool lz = (nlazy | (uint32_t)(idx >= CTF_MT_N)) != 0;
        const uint32_t nx = (k + 1 < T) ? x0[k + 1 < T ? k + 1 : 0] : x1n;
        const uint32_t y = (x0[k] & 0x80000000u) | (nx & 0x7fffffffu);
        const uint32_t v = m[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g.win[T * j + k] = lz ? v : x0[k];
    }
    g.pos = npos; g.lazy = nlazy;
    g.n = WCAP; g.cur = 0;
}
// make sure `need` (<= WCAP) words are in the window; group-uniform, so the W lanes refill together
template <int W>
__device__ __forceinline__ void mtw_ensure(MtWin& g, int j, int gshift, uint32_t need) {
    if (g.n - g.cur < need) mtw_cycle<W>(g, j, gshift);
}
template <int W>
__device__ __forceinline__ uint32_t mtw_next(MtWin& g, int j, int gshift) {
    mtw_ensure<W>(g, j, gshift, 1);
    return mt_temper(g.win[g.cur++]);
}
// CPython random._randbelow_with_getrandbits(n): k = n.bit_length(); draw k bits until < n.
// The W lanes of the group look at the next W words of the stream at once (sub-lane q at word cur + q): the first accepted
// one is the draw and everything up to it is consumed — the same words in the same order as the one-word-at-a-time loop, in
// ~1.3 rounds instead of the ~5 that the slowest of a wave's 16 groups needs when the acceptance probability is 1/2.
template <int W>
__device__ __forceinline__ uint32_t py_randbelow(MtWin& g, int j, int gshift, uint32_t n) {
    const uint32_t sh = (uint32_t)__clz((int)n);
    if (W == 1) {
        uint32_t r;
        do { r = mtw_next<W>(g, j, gshift) >> sh; } while (r >= n);
        return r;
    }
    for (;;) {
        mtw_ensure<W>(g, j, gshift, (uint32_t)W);
        const uint32_t r = mt_temper(g.win[g.cur + j]) >> sh;
        const uint32_t ok = (uint32_t)(__ballot(r < n) >> gshift) & ((1u << W) - 1u);
        if (ok) {
            const int first = __ffs((int)ok) - 1;
            g.cur += (uint32_t)first + 1u;
            return (uint32_t)__shfl((int)r, gshift + first, WAVE);
        }
        g.cur += W;
    }
}
// NumPy legacy randint(k), k >= 1: masked rejection on one 32-bit word; k == 1 draws nothing
template <int W>
__device__ __forceinline__ uint32_t np_randint(MtWin& g, int j, int gshift, uint32_t k) {
    const uint32_t rng = k - 1;
    if (rng == 0) return 0;
    const uint32_t mask = 0xFFFFFFFFu >> __clz((int)rng);
    uint32_t v;
    do { v = mtw_next<W>(g, j, gshift) & mask; } while (v > rng);
    return v;
}

template <int W>
struct StepCtx {
    uint8_t* sg;   // grid  (LDS)
    uint8_t* sr;   // record (LDS)
    uint8_t* sm;   // metric deltas of this step (LDS) or nullptr
    int j;         // sub-lane within the env's group
    int gshift;    // first lane of the group
    bool lead;     // j == 0: performs the state writes
    __device__ __forceinline__ uint32_t ballot(bool p) const {
        return (uint32_t)(__ballot(p) >> gshift) & ((1u << W) - 1u);
    }
};

template <int W>
__device__ __forceinline__ double ld_hp(const StepCtx<W>& s, int a) {
    const uint32_t* q = (const uint32_t*)(s.sr + 8 * a);
    return __hiloint2double((int)q[1], (int)q[0]);
}
template <int W>
__device__ __forceinline__ void st_hp(const StepCtx<W>& s, int a, double v) {
    uint32_t* q = (uint32_t*)(s.sr + 8 * a);
    q[0] = (uint32_t)__double2loint(v);
    q[1] = (uint32_t)__double2hiint(v);
}
template <bool METRICS, int W>
__device__ __forceinline__ void metric_add(const DevCfg& cfg, const StepCtx<W>& s, int m, int a, int v) {
    if (METRICS && s.lead) s.sm[m * cfg.N + a] += (uint8_t)v;  // per-step deltas stay far below 256
}
// a counter that agent a's turn touches exactly once per step: a plain store instead of an LDS read-modify-write (the deltas
// start the step at zero), so that consecutive updates do not wait for each other's reads
template <bool METRICS, int W>
__device__ __forceinline__ void metric_set(const DevCfg& cfg, const StepCtx<W>& s, int m, int a, int v) {
    if (METRICS && s.lead) s.sm[m * cfg.N + a] = (uint8_t)v;
}

// respawn, gridworld_ctf.py:761-794 (all lanes of the group compute; sub-lane 0 writes)
template <int W>
__device__ __forceinline__ void respawn(const DevCfg& cfg, const StepCtx<W>& s, MtWin& np_, int o, uint32_t& flagm, uint32_t& status) {
    const int G = cfg.G, team = cfg_team(cfg, o), type = cfg_type(cfg, o);
    const int x = TSEL(cfg.spawn_pos, team, 0), y = TSEL(cfg.spawn_pos, team, 1);
    const int r0 = x - 1 > 0 ? x - 1 : 0, c0 = y - 1 > 0 ? y - 1 : 0;
    const int r1 = x + 2 < G ? x + 2 : G, c1 = y + 2 < G ? y + 2 : G;
    // open cells of the (clipped) 3x3 window as a bitmask in row-major candidate order
    uint32_t open = 0;
    int k = 0;
    for (int r = r0; r < r1; r++)
        for (int c = c0; c < c1; c++)
            if (s.sg[r * G + c] == 0) { open |= 1u << ((r - r0) * 3 + (c - c0)); k++; }
    if (k == 0) { status |= CTF_ST_NO_RESPAWN; return; }
    const uint32_t rnd = np_randint<W>(np_, s.j, s.gshift, (uint32_t)k);
    uint32_t bits = open;
    for (uint32_t t = 0; t < rnd; t++) bits &= bits - 1;  // drop the rnd lowest candidates
    const int sel = __ffs((int)bits) - 1;
    int nr = x + sel / 3 - 1, nc = y + sel % 3 - 1;  // "-1" even when the window was clipped (:775)
    if (nr < 0 || nc < 0) { status |= CTF_ST_SPAWN_EDGE; nr = nr < 0 ? nr + G : nr; nc = nc < 0 ? nc + G : nc; }
    int8_t* ps = (int8_t*)(s.sr + cfg.off_pos);
    const int orow = ps[2 * o], ocol = ps[2 * o + 1];
    const bool carrying = (flagm >> o) & 1u;
    if (s.lead) {
        s.sg[orow * G + ocol] = 0;
        s.sg[nr * G + nc] = (uint8_t)(4 + type + 4 * team);
        ps[2 * o] = (int8_t)nr;
        ps[2 * o + 1] = (int8_t)nc;
        st_hp(s, o, sel4(cfg.type_hp, type));
        if (carrying) {
            if (cfg.drop_flag) s.sg[orow * G + ocol] = (uint8_t)(12 + (1 - team));
            else s.sg[TSEL(cfg.flag_pos, 1 - team, 0) * G + TSEL(cfg.flag_pos, 1 - team, 1)] = (uint8_t)(12 + (1 - team));
        }
    }
    flagm &= ~(1u << o);
}

// GridworldCtf.step for ONE env (state in LDS), executed by the W lanes of its group.
template <bool METRICS, int W>
__device__ __forceinline__ void env_step(const DevCfg& cfg, const DevPtrs& p, const StepCtx<W>& s, const int8_t* act, MtWin& py,
                                         MtWin& np_, uint32_t& status, int e, float* __restrict__ rw32, double* __restrict__ rw64,
                                         uint8_t* __restrict__ done_out) {
    const int N = cfg.N, G = cfg.G, j = s.j;
    int32_t* misc = (int32_t*)(s.sr + cfg.off_misc);
    int8_t* ps = (int8_t*)(s.sr + cfg.off_pos);
    int16_t* inv = (int16_t*)(s.sr + cfg.off_inv);

    // replicated register copies of the small per-env state: step, captures, has_flag bits, _arr as nibbles
    const int step = misc[0] + 1;
    int caps[2] = {misc[1], misc[2]};
    int vis_flags = misc[3];
    if (METRICS && (step - 1) - (vis_flags >> CTF_F_FOLDED_SHIFT) >= CTF_VIS_LOG - 1) {
        // the env went 511 steps without a reset: fold its log into the base maps before entry `step` reuses a slot
        uint32_t* base = p.vis + (size_t)e * N * cfg.GS;
        if (vis_flags & CTF_F_BASE_ZERO) {
            for (int w = j; w < N * cfg.GS; w += W) base[w] = 0;
            __builtin_amdgcn_s_waitcnt(0);
            for (int i = j; i < N; i += W) atomicAdd(base + i * cfg.GS + cfg.start_pos[i][0] * G + cfg.start_pos[i][1], 1u);
        }
        for (int st = (vis_flags >> CTF_F_FOLDED_SHIFT) + 1; st <= step - 1; st++)
            for (int i = j; i < N; i += W)
                atomicAdd(base + i * cfg.GS + p.vislog[((size_t)(st & (CTF_VIS_LOG - 1)) * cfg.n_envs + e) * N + i], 1u);
        vis_flags = (vis_flags & CTF_F_DONE) | ((step - 1) << CTF_F_FOLDED_SHIFT);
    }
    uint32_t flagm = 0;
    uint64_t perm = 0;
#pragma unroll
    for (int i = 0; i < CTF_MAX_AGENTS; i++) {
        if (i < N) {
            flagm |= (uint32_t)(s.sr[cfg.off_flag + i] & 1u) << i;
            perm |= (uint64_t)(s.sr[cfg.off_perm + i] & 15u) << (4 * i);
        }
    }
    uint32_t cap_mask = 0, resp_mask = 0, cap_team = 0;

    STEP_STAMP(33);
    // Two shuffles per step: dice_roll (:734-742) before the agents act and the one inside heal_agents (:839-847)
    // after.  One rolled loop holds both so that the RNG refill code exists once.
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {
        // top the CPython window up while the whole wave is at the same point (rejection sampling lets the groups'
        // stream positions diverge; refilling on demand would re-run the refill per group)
        if (!(STEP_ABLATE & 4)) mtw_cycle<W>(py, j, s.gshift);
        STEP_STAMP(phase == 0 ? 2 : 28);
#pragma unroll 1
        for (int i = (STEP_ABLATE & 4) ? 0 : N - 1; i >= 1; i--) {  // random.shuffle(self._arr)
            const uint32_t r = py_randbelow<W>(py, j, s.gshift, (uint32_t)i + 1u);
            const uint64_t vi = (perm >> (4 * i)) & 15u, vr = (perm >> (4 * r)) & 15u;
            perm = (perm & ~((uint64_t)15u << (4 * i)) & ~((uint64_t)15u << (4 * r))) | (vr << (4 * i)) | (vi << (4 * r));
        }
        STEP_STAMP(phase == 0 ? 3 : 29);
        if (phase == 1) break;

#pragma unroll 1
        for (int k = 0; k < N; k++) {
            const int a = (int)((perm >> (4 * k)) & 15u);
            const int type = cfg_type(cfg, a), team = cfg_team(cfg, a);
            int action = act[a];
            if (action < 0 || action >= CTF_N_ACTIONS) { status |= CTF_ST_BAD_ACTION; action = 4; }

            // ---- act (:700-732); ACTION_DELTAS (:100-145): vaulter jumps 2, miner acts at distance 1 on 5..8
            const int base = action <= 4 ? action : action - 5;
            const int scale = action <= 4 ? 1 : (type == 2 ? 2 : (type == 3 ? 1 : 0));
            const int dr = (base == 0 ? -1 : (base == 1 ? 1 : 0)) * scale;
            const int dc = (base == 2 ? 1 : (base == 3 ? -1 : 0)) * scale;
            int pr = ps[2 * a], pc = ps[2 * a + 1];
            const int nr = pr + dr, nc = pc + dc;
            if (!(STEP_ABLATE & 8) && nr >= 0 && nr < G && nc >= 0 && nc < G) {
                const int cell = s.sg[nr * G + nc];
                if (cell == 0 && (action <= 3 || (action >= 5 && type == 2 && (ld_hp(s, a) - cfg.vault_cost) > cfg.vault_min))) {
                    // movement_handler (:569-612)
                    const int ofr = TSEL(cfg.flag_pos, 1 - team, 0), ofc = TSEL(cfg.flag_pos, 1 - team, 1);
                    const int hfr = TSEL(cfg.flag_pos, team, 0), hfc = TSEL(cfg.flag_pos, team, 1);
                    const int opp_flag_cell = s.sg[ofr * G + ofc], home_flag_cell = s.sg[hfr * G + hfc];  // neither is the moved-from / moved-to cell
                    if (s.lead) {
                        s.sg[pr * G + pc] = 0;
                        s.sg[nr * G + nc] = (uint8_t)(4 + type + 4 * team);
                        ps[2 * a] = (int8_t)nr;
                        ps[2 * a + 1] = (int8_t)nc;
                    }
                    pr = nr; pc = nc;
                    if (cheb(nr, nc, ofr, ofc) <= 1 && opp_flag_cell == 12 + (1 - team)) {  // pickup: flag cell -> BLOCK
                        flagm |= 1u << a;
                        if (s.lead) s.sg[ofr * G + ofc] = 1;
                        metric_add<METRICS>(cfg, s, CTF_M_FLAG_PICKUPS, a, 1);
                    }
                    if (cheb(nr, nc, hfr, hfc) <= 1 && ((flagm >> a) & 1u)) {  // capture
                        if (!cfg.home_flag_capture || home_flag_cell == 12 + team) {
                            flagm &= ~(1u << a);
                            if (s.lead) s.sg[ofr * G + ofc] = (uint8_t)(12 + (1 - team));
                            caps[0] += team == 0;
                            caps[1] += team == 1;
                            metric_add<METRICS>(cfg, s, CTF_M_FLAG_CAPTURES, a, 1);
                            cap_mask |= 1u << a;
                            cap_team |= 1u << team;
                        }
                    }
                    if (action >= 5 && type == 2) {  // update_vaulter_hp
                        const double h = ld_hp(s, a) - cfg.vault_cost;
                        if (s.lead) st_hp(s, a, h);
                    }
                } else if (action >= 5 && type == 3 && inv[a] > 0 && cell == 0 &&
                           cheb(nr, nc, TSEL(cfg.spawn_pos, team, 0), TSEL(cfg.spawn_pos, team, 1)) > 1 &&
                           cheb(nr, nc, TSEL(cfg.spawn_pos, 1 - team, 0), TSEL(cfg.spawn_pos, 1 - team, 1)) > 1) {
                    if (s.lead) {
                        s.sg[nr * G + nc] = 2;  // add_block (:614-634)
                        inv[a] -= 1;
                    }
                    if (METRICS) {
                        metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID, a, 1);
                        metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID_DIST_OWN_FLAG, a,
                                            cheb(pr, pc, TSEL(cfg.capture_pos, team, 0), TSEL(cfg.capture_pos, team, 1)));
                        metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID_DIST_OPP_FLAG, a,
                                            cheb(pr, pc, TSEL(cfg.capture_pos, 1 - team, 0), TSEL(cfg.capture_pos, 1 - team, 1)));
                    }
                } else if (action < 5 && type == 3 && (cell == 2 || cell == 3)) {
                    if (cell == 2) {
                        if (s.lead) s.sg[nr * G + nc] = 3;  // mine_block (:677-690)
                    } else {
                        if (s.lead) {
                            s.sg[nr * G + nc] = 0;
                            if (inv[a] < 1000) inv[a] += 1;
                        }
                        metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_MINED, a, 1);
                    }
                }
            }

            STEP_STAMP(4 + 3 * (k & 7));
            // ---- tagging_logic (:796-837): sub-lane q evaluates opponent q0+q; hits are applied in opponent order
            const double dmg = sel4(cfg.type_damage, type);
            if (!(STEP_ABLATE & 1) && dmg > 0) {
                double mult = 1.0;
                if (type == 1 && cheb(pr, pc, TSEL(cfg.flag_pos, team, 0), TSEL(cfg.flag_pos, team, 1)) <= 3) mult = cfg.guard_mult;
                const double hit = dmg * mult;
                const int no = cfg_nopp(cfg, team);
                int q0 = 0;
#pragma unroll 1
                while (q0 < no) {
                    const int left = no - q0;
                    const int cnt = left < W ? left : W;  // opponents evaluated in this pass
                    mtw_ensure<W>(np_, j, s.gshift, (uint32_t)(2 * cnt));
                    bool is_hit = false;
                    if (j < cnt) {
                        const int o = cfg_opp(cfg, team, q0 + j);
                        const uint32_t wa = mt_temper(np_.win[np_.cur + 2 * j]) >> 5, wb = mt_temper(np_.win[np_.cur + 2 * j + 1]) >> 6;
                        const double u = ((double)wa * 67108864.0 + (double)wb) * (1.0 / 9007199254740992.0);  // np.random.rand()
                        is_hit = (u < cfg.tag_p) && cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1;
                    }
                    const uint32_t hits = s.ballot(is_hit);
                    if (hits == 0) {  // nobody tagged: all cnt doubles consumed
                        np_.cur += 2 * cnt;
                        q0 += cnt;
                        continue;
                    }
                    const int first = __ffs((int)hits) - 1;
                    np_.cur += 2 * (first + 1);  // doubles up to and including the tagged opponent's
                    const int o = cfg_opp(cfg, team, q0 + first);
                    const double h = ld_hp(s, o) - hit;
                    if (s.lead) st_hp(s, o, h);
                    metric_add<METRICS>(cfg, s, CTF_M_TAG_COUNT, a, 1);
                    if (h <= 0) {
                        if ((flagm >> o) & 1u) metric_add<METRICS>(cfg, s, CTF_M_FLAG_DISPOSSESSIONS, a, 1);
                        respawn<W>(cfg, s, np_, o, flagm, status);  // may draw randint words right here
                        resp_mask |= 1u << a;
                        metric_add<METRICS>(cfg, s, CTF_M_RESPAWN_TAG_COUNT, a, 1);
                    }
                    q0 += first + 1;
                }
            }

            STEP_STAMP(5 + 3 * (k & 7));
            // ---- metric-only section (:879-902): one teammate / opponent per sub-lane
            if (METRICS && !(STEP_ABLATE & 2)) {
                if (cheb(pr, pc, TSEL(cfg.capture_pos, team, 0), TSEL(cfg.capture_pos, team, 1)) <= 3)
                    metric_set<METRICS>(cfg, s, CTF_M_STEPS_DEFENDING_ZONE, a, 1);
                if (cheb(pr, pc, TSEL(cfg.capture_pos, 1 - team, 0), TSEL(cfg.capture_pos, 1 - team, 1)) <= 3)
                    metric_set<METRICS>(cfg, s, CTF_M_STEPS_ATTACKING_ZONE, a, 1);
                const int n_own = cfg_nopp(cfg, 1 - team), n_opp = cfg_nopp(cfg, team);
                int adj_own = 0, adj_opp = 0;
                for (int q = j; q < 8; q += W) {  // lists hold at most 8 agents
                    bool near_own = false, near_opp = false;
                    if (q < n_own) {  // OPPONENTS[1-team]: own team, self included
                        const int mt = cfg_opp(cfg, 1 - team, q);
                        near_own = cheb(pr, pc, ps[2 * mt], ps[2 * mt + 1]) <= 1;
                    }
                    if (q < n_opp) {
                        const int o = cfg_opp(cfg, team, q);
                        near_opp = cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1;
                    }
                    adj_own += __popc(s.ballot(near_own));
                    adj_opp += __popc(s.ballot(near_opp));
                    if (q - j + W >= (n_own > n_opp ? n_own : n_opp)) break;  // group-uniform exit
                }
                metric_set<METRICS>(cfg, s, CTF_M_STEPS_ADJ_TEAMMATE, a, adj_own);
                metric_set<METRICS>(cfg, s, CTF_M_STEPS_ADJ_OPPONENT, a, adj_opp);
            }
            STEP_STAMP(6 + 3 * (k & 7));
        }
    }  // phase

    // heal_agents (:839-847), after its shuffle above; order is irrelevant to the result: one agent per sub-lane
    for (int a = j; a < N; a += W) {
        const double mx = sel4(cfg.type_hp, cfg_type(cfg, a));
        double h = ld_hp(s, a);
        if (h < mx) {
            h += cfg.heal;
            st_hp(s, a, h > mx ? mx : h);
        }
    }

    // rewards: act() reward, + tagging reward, adjusted (:957-966), terminal (:920-940) — same op order; one agent per sub-lane
    const bool terminal = (step == cfg.game_steps);
    int winner = -1, margin = 0;
    if (terminal) {
        margin = iabs_(caps[0] - caps[1]);
        winner = caps[0] > caps[1] ? 0 : (caps[0] < caps[1] ? 1 : -1);
    }
    const int flags_in = misc[3];
    const int done_now = ((flags_in & CTF_F_DONE) != 0) || terminal;
    for (int i = j; i < N; i += W) {
        const int team = cfg_team(cfg, i);
        double r = 0.0 + cfg.r_step;
        if ((cap_mask >> i) & 1u) r += cfg.r_capture;
        r += ((resp_mask >> i) & 1u) ? cfg.r_tag : 0.0;
        if (cfg.use_adjusted) r -= (((cap_team >> (1 - team)) & 1u) ? 1.0 : 0.0) * cfg.r_capture * cfg.punish;
        if (winner >= 0) {
            if (team == winner) r += margin * cfg.win_scalar;
            else r -= margin * cfg.loss_scalar;
        }
        if (rw32) rw32[(size_t)e * N + i] = (float)r;
        if (rw64) rw64[(size_t)e * N + i] = r;
        if (METRICS && !(STEP_ABLATE & 16)) {  // update_visitation_map (:479-486) as a log entry: slot step % 512
            p.vislog[((size_t)(step & (CTF_VIS_LOG - 1)) * cfg.n_envs + e) * N + i] = (uint16_t)(ps[2 * i] * G + ps[2 * i + 1]);
        }
    }

    STEP_STAMP(30);
    // ---- the replicated registers go back to the record (sub-lane 0)
    if (s.lead) {
        misc[0] = step;
        misc[1] = caps[0];
        misc[2] = caps[1];
        misc[3] = (vis_flags & ~CTF_F_DONE) | (done_now ? CTF_F_DONE : 0);
        if (done_out) done_out[e] = (uint8_t)done_now;
#pragma unroll
        for (int i = 0; i < CTF_MAX_AGENTS; i++) {
            if (i < N) {
                s.sr[cfg.off_flag + i] = (uint8_t)((flagm >> i) & 1u);
                s.sr[cfg.off_perm + i] = (uint8_t)((perm >> (4 * i)) & 15u);
            }
        }
    }
}

template <bool METRICS, int W>
__global__ void __launch_bounds__(WAVE)
#if STEP_TRACE
__attribute__((amdgpu_waves_per_eu(4, 4)))  // a trace must keep the shipped kernel's residency (16 blocks per CU)
#endif
k_step(DevCfg cfg, DevPtrs p, const int8_t* __restrict__ actions,
                                                float* __restrict__ rw32, double* __restrict__ rw64,
                                                uint8_t* __restrict__ done_out, uint32_t flags) {
    constexpr int EPW = WAVE / W;  // envs per wave
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    STEP_STAMP(0);
    const int g = lane / W, j = lane % W;
    const int env0 = blockIdx.x * EPW;
    const int nvalid = min(EPW, cfg.n_envs - env0);
    const int SLB = step_slot_bytes(cfg.GS, cfg.RS, cfg.N, METRICS);
    const int SLW = SLB / 4, GW = cfg.GS / 4, RW = cfg.RS / 4;
    const int AW = 4, WW = 2 * WCAP;  // action words, RNG window words per slot
    const int N = cfg.N;
    const int e = env0 + g;
    // the two stream positions: asked for before the staging loads, needed right behind the barrier
    uint32_t rp_py = 0, rp_np = 0;
    if (g < nvalid) {
        rp_py = p.rngpos[2 * e];
        rp_np = p.rngpos[2 * e + 1];
    }

    // ---- stage the wave's envs' grids, records and actions into LDS.  Flat, coalesced 16-byte loads, unrolled so
    // that every lane has several independent loads in flight (a rolled per-env loop pays one memory latency per env).
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* gsrc = (const u32x4*)(p.grid + (size_t)env0 * cfg.GS);
        const u32x4* rsrc = (const u32x4*)(p.rec + (size_t)env0 * cfg.RS);
        const int GQ = GW / 4, RQ = RW / 4;  // 16-byte quads per env (GS and RS are multiples of 16)
        const int ng = nvalid * GQ, nr = nvalid * RQ;
#pragma unroll 4
        for (int q = lane; q < ng; q += WAVE) {
            const u32x4 v = gsrc[q];
            const int el = (int)fdiv((uint32_t)q, cfg.div_gq), w = (q - el * GQ) * 4;
            uint32_t* slot = lds + el * SLW + w;
            slot[0] = v.x; slot[1] = v.y; slot[2] = v.z; slot[3] = v.w;
        }
#pragma unroll 2
        for (int q = lane; q < nr; q += WAVE) {
            const u32x4 v = rsrc[q];
            const int el = (int)fdiv((uint32_t)q, cfg.div_rq), w = (q - el * RQ) * 4;
            uint32_t* slot = lds + el * SLW + GW + w;
            slot[0] = v.x; slot[1] = v.y; slot[2] = v.z; slot[3] = v.w;
        }
        const int8_t* asrc = actions + (size_t)env0 * N;
#pragma unroll 2
        for (int idx = lane; idx < nvalid * N; idx += WAVE) {
            const int el = (int)fdiv((uint32_t)idx, cfg.div_n), i = idx - el * N;
            ((int8_t*)(lds + el * SLW + GW + RW))[i] = asrc[idx];
        }
        if (METRICS) {
            const int MW = (CTF_N_METRICS * N + 3) / 4;
            for (int idx = lane; idx < nvalid * MW; idx += WAVE) {
                const int el = (int)fdiv((uint32_t)idx, cfg.div_mw), w = idx - el * MW;
                lds[el * SLW + GW + RW + AW + WW + w] = 0;
            }
        }
    }
    __syncthreads();
    STEP_STAMP(1);

    if (g < nvalid) {
        StepCtx<W> s;
        s.sg = (uint8_t*)(lds + g * SLW);
        s.sr = s.sg + cfg.GS;
        uint32_t* wins = (uint32_t*)(s.sr + cfg.RS + 16);
        s.sm = METRICS ? (uint8_t*)(wins + WW) : nullptr;
        s.j = j;
        s.gshift = g * W;
        s.lead = (j == 0);
        const int8_t* act = (const int8_t*)(s.sr + cfg.RS);
        int32_t* misc = (int32_t*)(s.sr + cfg.off_misc);

        MtWin py = mtw_open(p.mt_py + (size_t)e * CTF_MT_N, wins, rp_py);
        MtWin npg = mtw_open(p.mt_np + (size_t)e * CTF_MT_N, wins + WCAP, rp_np);

        if ((flags & CTF_STEP_AUTO_RESET) && (misc[3] & CTF_F_DONE)) {
            // reset() of this env inside the step launch (not in the reference: opt-in flag); the group's lanes share the copies
            const uint32_t* src = (const uint32_t*)p.init_grid;
            for (int w = j; w < GW; w += W) ((uint32_t*)s.sg)[w] = src[w];
            if (s.lead) reset_record(cfg, s.sr);
            if (METRICS) {
                int32_t* m = p.metrics + (size_t)e * CTF_N_METRICS * N;
                for (int w = j; w < CTF_N_METRICS * N; w += W) m[w] = 0;
                // (visitation: reset_record flagged the base maps as zero and emptied the log)
            }
        }

        uint32_t status = 0;
        env_step<METRICS, W>(cfg, p, s, act, py, npg, status, e, rw32, rw64, done_out);
        mtw_flush<W>(py, j);
        mtw_flush<W>(npg, j);
        if (s.lead) {
            p.rngpos[2 * e] = py.pos | (py.lazy ? CTF_LAZY_BIT : 0u);
            p.rngpos[2 * e + 1] = npg.pos | (npg.lazy ? CTF_LAZY_BIT : 0u);
            if (status) atomicOr(p.status, status);
        }
        STEP_STAMP(31);
    }
    __syncthreads();

    // ---- write the envs back (flat, coalesced 16-byte stores)
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4* gdst = (u32x4*)(p.grid + (size_t)env0 * cfg.GS);
        u32x4* rdst = (u32x4*)(p.rec + (size_t)env0 * cfg.RS);
        const int GQ = GW / 4, RQ = RW / 4;
        const int ng = nvalid * GQ, nr = nvalid * RQ;
#pragma unroll 2
        for (int q = lane; q < ((STEP_ABLATE & 64) ? 0 : ng); q += WAVE) {
            const int el = (int)fdiv((uint32_t)q, cfg.div_gq), w = (q - el * GQ) * 4;
            const uint32_t* slot = lds + el * SLW + w;
            const u32x4 v = {slot[0], slot[1], slot[2], slot[3]};
            gdst[q] = v;
        }
#pragma unroll 2
        for (int q = lane; q < nr; q += WAVE) {
            const int el = (int)fdiv((uint32_t)q, cfg.div_rq), w = (q - el * RQ) * 4;
            const uint32_t* slot = lds + el * SLW + GW + w;
            const u32x4 v = {slot[0], slot[1], slot[2], slot[3]};
            rdst[q] = v;
        }
        if (METRICS && !(STEP_ABLATE & 32)) {
            // this step's u8 deltas are added to the i32 counters with no-return atomics (nothing to wait for); two
            // neighbouring counters share one 64-bit add (a counter never carries out of its 32 bits)
            const int MN = CTF_N_METRICS * N;
            int32_t* mdst = p.metrics + (size_t)env0 * MN;
            if ((MN & 1) == 0) {
                for (int idx = lane; idx < nvalid * MN / 2; idx += WAVE) {
                    const int el = (int)fdiv((uint32_t)(2 * idx), cfg.div_mn), w = 2 * idx - el * MN;
                    const uint8_t* d = (const uint8_t*)(lds + el * SLW + GW + RW + AW + WW) + w;
                    const unsigned long long inc = (unsigned long long)d[0] | ((unsigned long long)d[1] << 32);
                    if (inc) atomicAdd((unsigned long long*)(mdst + 2 * idx), inc);
                }
            } else {
                for (int idx = lane; idx < nvalid * MN; idx += WAVE) {
                    const int el = (int)fdiv((uint32_t)idx, cfg.div_mn), w = idx - el * MN;
                    const uint8_t inc = ((const uint8_t*)(lds + el * SLW + GW + RW + AW + WW))[w];
                    if (inc) atomicAdd(mdst + idx, (int32_t)inc);
                }
            }
        }
    }
    STEP_STAMP(32);
}

// ------------------------------------------------------------------------------------------------
// observe — one wave per env at a time; the env's whole one-hot block is first built as a BITMAP in LDS
// ------------------------------------------------------------------------------------------------
// The observation block u8 [N][C][G][G] of one env is ~98 % zeros.  Per env a wave
//   1. zeroes a bitmap of N*C*G*G bits in LDS (one bit per output byte),
//   2. sets the hot bits: for every non-empty grid cell and every agent, bit
//      i*C*G*G + channel(viewer team, tile)*G*G + (reversed ? flip(cell) : cell), plus the own-position
//      bit of plane 0 — LDS atomic ORs, relabelling via the SGPR-resident channel LUT,
//   3. streams the block out: 16-byte chunk k of the output is halfword k of the bitmap with every bit
//      expanded to a byte (3 full-rate VALU ops per 4 bytes), one 1-KiB-aligned coalesced store
//      instruction per 64 chunks.
// Metadata rows (f16 [N][2N+6]) are assembled in LDS and leave as 8-byte stores.
//
// Per-wave LDS (bytes): [rec RS][mvals 96][meta staging][meta source LUT][bitmap]
#define OBS_MV_BYTES 96
__host__ __device__ inline int obs_meta_stage_bytes(int N, int M) { return (N * M * 2 + 15) & ~15; }
__host__ __device__ inline int obs_bitmap_bytes(int obs_bytes) { return ((((obs_bytes + 31) / 32 + 1) * 4) + 15) & ~15; }
__host__ __device__ inline int obs_meta_lut_bytes(int N, int M) { return (N * M + 15) & ~15; }
__host__ __device__ inline int obs_wave_bytes(int RS, int N, int M, int obs_bytes) {
    return RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M) + obs_meta_lut_bytes(N, M) + obs_bitmap_bytes(obs_bytes);
}

#ifndef OBS_TILES_DEFAULT
#define OBS_TILES_DEFAULT 1  // ctf_launch_observe takes the tile render whenever it applies (CTF_OBS_TILES=0 / 1 overrides)
#endif
#define LGKM_ONLY 0xC07F  // s_waitcnt lgkmcnt(0): LDS traffic only — never drain the wave's outstanding stores
// Profiling-only ablations (never defined in the shipped build; see tools/ablate.sh):
//   bit0 no bit expansion, bit1 no chunk stores, bit2 no bitmap build, bit3 no metadata, bit4 metadata computed but not stored
#ifndef OBS_ABLATE
#define OBS_ABLATE 0
#endif
// experiment (G <= 16 only, results wrong beyond): no compiler-tracked load anywhere in the render loop
#ifndef OBS_NODRAIN
#define OBS_NODRAIN 0
#endif

template <int ALIGN>
struct OutVec;
template <>
struct OutVec<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <>
struct OutVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int flip_cell(const DevCfg& cfg, int cell, int r, int c) {
    // destination of cell (r, c) under reverse_grid (gridworld_ctf.py:1003-1007)
    const int G = cfg.G;
    if (cfg.flip_axis == -1) return cfg.GG - 1 - cell;      // np.flip over both axes
    if (cfg.flip_axis == 0) return (G - 1 - r) * G + c;     // np.flip(plane, 0)
    if (cfg.flip_axis == 1) return r * G + (G - 1 - c);     // np.flip(plane, 1)
    return (G - 1 - c) * G + (G - 1 - r);                   // np.rot90(plane.T, 2)
}
// A load the compiler does not track: issued one env ahead of its use, waited for by obs_prefetch_wait.
// (vmcnt retires in issue order, so a compiler-placed wait for next env's state would first drain every
// store of the current env; issued BEFORE those stores and waited for with a counted vmcnt a few
// stores later, the data is simply there when the wave reaches the next env.)
__device__ __forceinline__ uint32_t obs_prefetch_dword(const uint32_t* ptr) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
#ifndef OBS_PREFETCH
#define OBS_PREFETCH 1  // 0 = profiling comparison only (tools/ablate.sh)
#endif
#ifndef OBS_UNROLL
#define OBS_UNROLL 8    // store instructions per pass of the stream loop
#endif
#define OBS_PF_WAIT 16  // stream iteration (a multiple of OBS_UNROLL) at which the prefetched state is waited for ...
// ... with vmcnt(8): the two prefetch loads are older than the >= OBS_PF_WAIT stores issued since
#define OBS_PREFETCH_WAIT(a, b) asm volatile("s_waitcnt vmcnt(8)" : "+v"(a), "+v"(b)::"memory")
#define OBS_PREFETCH_DRAIN(a, b) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)::"memory")

// 4 bits -> 4 bytes of 0/1: the four shifted copies of the nibble do not overlap, so no carries
__device__ __forceinline__ uint32_t expand4(uint32_t h, int j) {
    return (((h >> (4 * j)) & 15u) * 0x00204081u) & 0x01010101u;
}

struct ObsSlots {  // which agents look through each (viewer team, reversed?) view — uniform over the launch
    uint32_t a[4];
};
__device__ __forceinline__ ObsSlots obs_slots(const DevCfg& cfg, uint32_t reverse_mask) {
    ObsSlots s = {{0, 0, 0, 0}};
    for (int i = 0; i < cfg.N; i++) {
        const int sl = cfg.team[i] * 2 + (int)((reverse_mask >> i) & 1u);
        s.a[0] |= (sl == 0) ? (1u << i) : 0u;
        s.a[1] |= (sl == 1) ? (1u << i) : 0u;
        s.a[2] |= (sl == 2) ? (1u << i) : 0u;
        s.a[3] |= (sl == 3) ? (1u << i) : 0u;
    }
    return s;
}

// Which of the (at most 36 distinct + two constant) values mv[] each element of the N x M metadata block shows
// (gridworld_ctf.py:1044-1067): mv[0] step fraction, mv[1 + t] capture ratio for a viewer of team t, mv[4 + j] the
// uint8-truncated hp of agent j, mv[20 + j] has_flag[j], mv[40] = 1.0, mv[41] = 0.0.
__device__ __forceinline__ void obs_meta_lut(const DevCfg& cfg, uint8_t* mlut, uint16_t* mv, int lane) {
    const int N = cfg.N, M = cfg.M;
    for (int idx = lane; idx < N * M; idx += WAVE) {
        const int i = (int)fdiv((uint32_t)idx, cfg.div_m), k = idx - i * M;
        const int team = cfg_team(cfg, i);
        int src = 41;
        if (k == 0) src = 0;
        else if (k == 1) src = 1 + team;
        else if (k < 6) src = (k - 2 == cfg_type(cfg, i)) ? 40 : 41;
        else {
            // rows 6,7: the agent itself; then own-team list minus self, then the opponents list (:1053-1067)
            int who = i;
            if (k >= 8) {
                const int pidx = (k - 8) >> 1;
                const int n_own = cfg_nopp(cfg, 1 - team), n_op = cfg_nopp(cfg, team);
                const int self_idx = (int)((pin64(cfg.self_idx_pack) >> (4 * i)) & 15u);
                const int n_mates = n_own - (self_idx < n_own ? 1 : 0);
                if (pidx < n_mates) who = cfg_opp(cfg, 1 - team, pidx + (pidx >= self_idx ? 1 : 0));
                else if (pidx - n_mates < n_op) who = cfg_opp(cfg, team, pidx - n_mates);
                else who = -1;
            }
            if (who >= 0) src = ((k & 1) ? 20 : 4) + who;
        }
        mlut[idx] = (uint8_t)src;
    }
    if (lane == 0) { mv[40] = 0x3C00u; mv[41] = 0u; }
}

// Everything of one env except the streaming: bitmap (zero + hot bits + own-position bits) into `bits`, metadata rows
// to global memory.  `recw` / `cells` are the lane's dword of the env's record / grid (lane-clamped loads).
// srec / mv / mstage are the calling wave's scratch.  Ends with the wave's LDS traffic drained.
__device__ __forceinline__ void obs_build_env(const DevCfg& cfg, const DevPtrs& p, int e, uint32_t recw, uint32_t cells,
                                              uint8_t* srec, uint16_t* mv, uint16_t* mstage, const uint8_t* mlut, uint32_t* bits,
                                              const ObsSlots& slots, uint32_t reverse_mask, int lane, bool obs,
                                              uint16_t* __restrict__ meta) {
    const int N = cfg.N, G = cfg.G, GG = cfg.GG, M = cfg.M;
    const int BQ = obs_bitmap_bytes(cfg.obs_bytes) / 16;
    const int GW = cfg.GS / 4;  // <= 256 dwords: up to 4 passes of 64 lanes
    const bool has_cells = obs && !(OBS_ABLATE & 4);
    const uint32_t* slot_agents = slots.a;
    // ---- zero the bitmap, park the record in LDS
    if (obs) {
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int q = lane; q < BQ; q += WAVE) ((u32x4_t*)bits)[q] = z;
    }
    if (lane < cfg.RS / 4) ((uint32_t*)srec)[lane] = recw;
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
    __builtin_amdgcn_wave_barrier();

    // ---- hot bits of every tile plane
    if (has_cells) {
        for (int w = lane; w < GW; w += WAVE) {
#if !OBS_NODRAIN
            if (w >= WAVE) cells = ((const uint32_t*)(p.grid + (size_t)e * cfg.GS))[w];  // G > 16 only
#endif
            int r = (int)fdiv((uint32_t)(w * 4), cfg.div_g), c = w * 4 - r * G;
            #pragma unroll
            for (int b = 0; b < 4; b++) {
                const int cell = w * 4 + b;
                const uint32_t v = (cells >> (8 * b)) & 0xFFu;
                if (v != 0 && cell < GG) {
                    const int fl = flip_cell(cfg, cell, r, c);
                    #pragma unroll
                    for (int slot = 0; slot < 4; slot++) {
                        if (slot_agents[slot]) {  // uniform
                            // nibble v of the 64-bit LUT with 32-bit ops (a 64-bit variable shift is several times slower)
                            const uint64_t lut = pin64(cfg.chan_lut[slot >> 1]);
                            const uint32_t code = (((v & 8u) ? (uint32_t)(lut >> 32) : (uint32_t)lut) >> (4 * (v & 7u))) & 15u;
                            if (code != CTF_TILE_NONE) {
                                const uint32_t q = code * (uint32_t)GG + (uint32_t)((slot & 1) ? fl : cell);
                                for (uint32_t m = slot_agents[slot]; m; m &= m - 1) {  // uniform loop over the slot's agents
                                    const uint32_t bit = (uint32_t)(__ffs((int)m) - 1) * (uint32_t)cfg.CGG + q;
                                    atomicOr(bits + (bit >> 5), 1u << (bit & 31u));
                                }
                            }
                        }
                    }
                }
                if (++c == G) { c = 0; r++; }
            }
        }
    }
    // ---- plane 0: the viewer's own position
    if (obs && lane < N) {
        const int8_t* ps = (const int8_t*)(srec + cfg.off_pos);
        const int r = ps[2 * lane], c = ps[2 * lane + 1];
        const int cell = ((reverse_mask >> lane) & 1u) ? flip_cell(cfg, r * G + c, r, c) : r * G + c;
        const uint32_t bit = (uint32_t)lane * (uint32_t)cfg.CGG + (uint32_t)cell;
        atomicOr(bits + (bit >> 5), 1u << (bit & 31u));
    }

    // ---- metadata (gridworld_ctf.py:1027-1069).  A: the few distinct values, as f16 bits
    if (meta && !(OBS_ABLATE & 8)) {
        const int32_t* misc = (const int32_t*)(srec + cfg.off_misc);
        if (lane < 36) {
            double val = 0.0;
            if (lane == 0) val = (double)misc[0] / (double)cfg.game_steps;
            else if (lane < 3) val = (double)(misc[lane] + 1) / (double)(misc[3 - lane] + 1);  // viewer team lane-1
            else if (lane >= 4 && lane < 4 + N) {
                // the quirk at :1039-1041: hp of the agent whose INDEX is type(j), over max hp of type(j), as uint8
                const int tv = cfg_type(cfg, lane - 4);
                double q = 0.0;
                if (tv < N) {
                    const uint32_t* hq = (const uint32_t*)(srec + 8 * tv);
                    q = __hiloint2double((int)hq[1], (int)hq[0]) / sel4(cfg.type_hp, tv);
                }
                val = (double)(uint8_t)(long long)q;
            } else if (lane >= 20 && lane < 20 + N) val = (double)srec[cfg.off_flag + lane - 20];
            mv[lane] = f64_to_f16(val);
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        // B: every element of the N x M block is one of those values (which one: obs_meta_lut, built once per wave)
        for (int idx = lane; idx < N * M; idx += WAVE) mstage[idx] = mv[mlut[idx]];
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        // C: N*M*2 = 4N(N+3) bytes, always a multiple of 8
        u32x2_t* mdst = (u32x2_t*)(meta + (size_t)e * N * M);
        if (!(OBS_ABLATE & 16))
        for (int q = lane; q < N * M / 4; q += WAVE) mdst[q] = ((const u32x2_t*)mstage)[q];
    }
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);  // the bitmap's atomic ORs have landed
    __builtin_amdgcn_wave_barrier();

}

// one 16-byte chunk of the observation block: h = halfword k of the bitmap, every bit expanded to a byte
template <int ALIGN>
__device__ __forceinline__ void obs_store_chunk(uint8_t* out, uint32_t h, int k, int nfull, int tail, uint32_t& ablate_acc) {
    const uint32_t o = (uint32_t)k << 4;
    uint32_t x[4];
    if (OBS_ABLATE & 1) { x[0] = x[1] = x[2] = x[3] = h; }
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = expand4(h, j);
    }
    if (OBS_ABLATE & 2) { ablate_acc ^= x[0] ^ x[1] ^ x[2] ^ x[3]; return; }
    if (k < nfull) {
        if (ALIGN >= 4) {
            typedef typename OutVec<(ALIGN >= 16 ? 16 : 4)>::type V;
            const V v = {x[0], x[1], x[2], x[3]};
            *(V*)(out + o) = v;  // plain store: measured faster than nontemporal for this pattern
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) out[o + j] = (uint8_t)(x[j >> 2] >> ((j & 3) * 8));
        }
    } else {
        for (int j = 0; j < tail; j++) out[o + j] = (uint8_t)(x[j >> 2] >> ((j & 3) * 8));
    }
}

// Every wave builds and streams its own envs.  (A builder / streamer split — one wave of a block building the next
// three envs' bitmaps while the other three stream — was tried and measured 15-25 % slower: a single builder wave's
// dependent chain is too long, and 24 streaming waves per CU drive the store path less well than 32.)
template <int ALIGN>
__global__ void __launch_bounds__(256) k_observe(DevCfg cfg, DevPtrs p, uint8_t* __restrict__ obs,
                                                 uint16_t* __restrict__ meta, uint32_t reverse_mask, uint32_t xcd_map) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    const int N = cfg.N, M = cfg.M;
    uint8_t* wl = (uint8_t*)lds + wave * obs_wave_bytes(cfg.RS, N, M, cfg.obs_bytes);
    uint8_t* srec = wl;
    uint16_t* mv = (uint16_t*)(wl + cfg.RS);
    uint16_t* mstage = (uint16_t*)(wl + cfg.RS + OBS_MV_BYTES);
    uint8_t* mlut = wl + cfg.RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M);
    uint32_t* bits = (uint32_t*)(mlut + obs_meta_lut_bytes(N, M));
    const ObsSlots slots = obs_slots(cfg, reverse_mask);
    if (meta) obs_meta_lut(cfg, mlut, mv, lane);
    const int GW = cfg.GS / 4;
    const int rec_lane = min(lane, cfg.RS / 4 - 1), grid_lane = min(lane, GW - 1);
    // Consecutive workgroups go to consecutive XCDs: with xcd_map (grid a multiple of 8 blocks) XCD x renders ITS OWN contiguous
    // eighth of the envs instead of every XCD writing into every page of the buffer (see k_observe_tiles).
    int e_first = blockIdx.x * wpb + wave, e_stride = gridDim.x * wpb, e_end = cfg.n_envs;
    if (xcd_map) {
        const int chunk = (cfg.n_envs + 7) >> 3, lo = (int)(blockIdx.x & 7u) * chunk;
        e_first = lo + (int)(blockIdx.x >> 3) * wpb + wave;
        e_stride = (int)(gridDim.x >> 3) * wpb;
        e_end = min(lo + chunk, cfg.n_envs);
    }
    // the first env's state: ordinary loads; every later env's state arrives through the prefetch below
    uint32_t recw = 0, cells = 0;
#if OBS_NODRAIN
    if (e_first >= e_end) return;
    recw = obs_prefetch_dword((const uint32_t*)(p.rec + (size_t)e_first * cfg.RS) + rec_lane);
    cells = obs_prefetch_dword((const uint32_t*)(p.grid + (size_t)e_first * cfg.GS) + grid_lane);
    OBS_PREFETCH_DRAIN(recw, cells);
#else
    if (e_first < e_end) {
        recw = ((const uint32_t*)(p.rec + (size_t)e_first * cfg.RS))[rec_lane];
        cells = ((const uint32_t*)(p.grid + (size_t)e_first * cfg.GS))[grid_lane];
    }
#endif

    for (int e = e_first; e < e_end; e += e_stride) {
        obs_build_env(cfg, p, e, recw, cells, srec, mv, mstage, mlut, bits, slots, reverse_mask, lane, obs != nullptr, meta);

        // ---- stream the observation block: 16 bytes per lane per store.  Wave store instructions are
        // aligned to 1 KiB of the flat output (k starts negative), so only an env's first and last
        // instruction touch a partial line.
        if (obs) {
            const size_t base = (size_t)e * cfg.obs_bytes;
            uint8_t* out = obs + base;
            const uint16_t* hb = (const uint16_t*)bits;
            const int nfull = cfg.obs_bytes >> 4;
            const int tail = cfg.obs_bytes & 15;
            const int nchunks = nfull + (tail ? 1 : 0);
            const int k0 = (ALIGN >= 16) ? -(int)(((base + (uintptr_t)obs) >> 4) & 63) : 0;
            uint32_t ablate_acc = 0;
            const int niter = (nchunks - k0 + WAVE - 1) / WAVE;
            const int e_next = min(e + e_stride, e_end - 1);
            uint32_t nrec = 0, ncells = 0;
            // OBS_UNROLL store instructions per pass, their bitmap halfwords read first: the stores then issue back to back
            // instead of each waiting for its own LDS round trip
            for (int it0 = 0; it0 < niter; it0 += OBS_UNROLL) {
                if (OBS_PREFETCH && it0 == 0) {  // next env's state: issued before this env's first store
                    nrec = obs_prefetch_dword((const uint32_t*)(p.rec + (size_t)e_next * cfg.RS) + rec_lane);
                    ncells = obs_prefetch_dword((const uint32_t*)(p.grid + (size_t)e_next * cfg.GS) + grid_lane);
                }
                if (OBS_PREFETCH && it0 == OBS_PF_WAIT) OBS_PREFETCH_WAIT(nrec, ncells);
                uint32_t h[OBS_UNROLL];
#pragma unroll
                for (int u = 0; u < OBS_UNROLL; u++) {
                    const int k = k0 + lane + (it0 + u) * WAVE;
                    h[u] = (k >= 0 && k < nchunks) ? hb[k] : 0u;
                }
#pragma unroll
                for (int u = 0; u < OBS_UNROLL; u++) {
                    const int k = k0 + lane + (it0 + u) * WAVE;
                    if (k >= 0 && k < nchunks) obs_store_chunk<ALIGN>(out, h[u], k, nfull, tail, ablate_acc);
                }
            }
            if ((OBS_ABLATE & 2) && ablate_acc == 0x12345678u) out[lane] = 1;  // keeps the ablated work alive
            if (OBS_PREFETCH) {
                if (niter <= OBS_PF_WAIT) OBS_PREFETCH_DRAIN(nrec, ncells);
                recw = nrec;
                cells = ncells;
            } else if (e + e_stride < e_end) {
                recw = ((const uint32_t*)(p.rec + (size_t)(e + e_stride) * cfg.RS))[rec_lane];
                cells = ((const uint32_t*)(p.grid + (size_t)(e + e_stride) * cfg.GS))[grid_lane];
            }
        } else if (e + e_stride < e_end) {
#if OBS_NODRAIN
            recw = obs_prefetch_dword((const uint32_t*)(p.rec + (size_t)(e + e_stride) * cfg.RS) + rec_lane);
            OBS_PREFETCH_DRAIN(recw, cells);
#else
            recw = ((const uint32_t*)(p.rec + (size_t)(e + e_stride) * cfg.RS))[rec_lane];
#endif
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);  // this env's LDS reads are done before the next env reuses the bitmap
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// observe as one-shot TILES — the same bitmap idea, organised by output address instead of by env
// ------------------------------------------------------------------------------------------------
// Wave t of the grid renders CTF_OBS_TILE (8 192) consecutive bytes of the FLAT observation buffer and exits: it loads the
// grid of the (at most two) envs its tile touches, builds only the bits of its own tile — for each of the agent views whose
// blocks intersect the tile, every non-empty cell gives one candidate bit — and issues eight store instructions.  Why:
// the chip then writes a compact window that moves linearly through the buffer and a wave issues few stores.  On MI355X a
// wave that issues 1 / 4 / 8 / 25 store instructions in a row sustains 6.9 / 5.9 / 5.7 / 5.5 TB/s chip-wide, and the long
// per-wave streams of k_observe lose another 15 % on the "slow" kind of allocation (on some boxes: every allocation) while
// one-shot tiles do not (tools/store_bw8.hip, profiles/r02_store_bw8_tiles.txt).  The price: the cell scan is repeated by
// every tile of an env (3.1 tiles per arena env) for the views the tile holds, and every wave pays the prologue — so both are
// kept lean: envs come in groups whose blocks fill a whole number of tiles (no 64-bit division), a cell's flipped index and
// its channel code for either viewing team are worked out once per grid dword, a view costs ~7 instructions per cell, and
// the metadata LUT comes from a table the host built.
// The 1-D grid is walked XCD-contiguously (block b -> logical block (b % 8) * (blocks / 8) + b / 8): each of the 8 XCDs writes
// its own eighth of the buffer front to back, 0.306 -> 0.259 ms on the arena (tools/store_bw9.hip for the bare pattern).
// Metadata rows: written by the wave whose tile holds an env's first byte.
// Used when an env's block is a multiple of 16 bytes and >= one tile and the buffer is 16-byte aligned (ctf_launch_observe).
#define OBS_TILE CTF_OBS_TILE
__host__ __device__ inline int tiles_wave_bytes(int RS, int N, int M) {
    return OBS_TILE / 8 + RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M);
}

__global__ void __launch_bounds__(256) k_observe_tiles(DevCfg cfg, DevPtrs p, uint8_t* __restrict__ obs, uint16_t* __restrict__ meta,
                                                       uint32_t reverse_mask, uint32_t xcd_map) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int N = cfg.N, M = cfg.M, G = cfg.G, GG = cfg.GG, CGG = cfg.CGG, OB = cfg.obs_bytes;
    // ---- which tile of which group of envs (uniform, 32-bit)
    // The launch is 1-D and consecutive workgroups go to consecutive XCDs (8 of them).  Block b therefore takes logical
    // block (b % 8) * (blocks / 8) + b / 8: every XCD writes ITS OWN contiguous eighth of the buffer front to back instead of
    // every XCD touching every page — a constant 8 KiB-tile fill measures 6.2 instead of 5.6 TB/s that way (tools/store_bw9.hip).
    const uint32_t b = blockIdx.x;
    const uint32_t lb = xcd_map ? (b & 7u) * ((uint32_t)cfg.tile_nb >> 3) + (b >> 3) : b;
    const uint32_t grp = fdiv(lb, cfg.div_tile_bx);
    const int tt = (int)(lb - grp * (uint32_t)cfg.tile_bx) * CTF_OBS_TILE_WPB + wave;
    if (tt >= cfg.tile_tpg) return;
    const int env_base = (int)grp * cfg.tile_k;
    if (env_base >= cfg.n_envs) return;
    const uint32_t lo_local = (uint32_t)tt * OBS_TILE;                  // < tile_k * obs_bytes
    const int el = (int)fdiv(lo_local, cfg.div_ob_tile);
    const int off0 = (int)(lo_local - (uint32_t)el * (uint32_t)OB);    // the tile starts at byte off0 of env e0's block
    const int e0 = env_base + el;
    if (e0 >= cfg.n_envs) return;
    const bool two = off0 + OBS_TILE > OB && e0 + 1 < cfg.n_envs;       // the tile runs into env e0 + 1 (obs_bytes >= tile: never further)
    uint8_t* wl = (uint8_t*)lds + wave * tiles_wave_bytes(cfg.RS, N, M);
    uint32_t* bits = (uint32_t*)wl;
    const int GW = cfg.GS / 4;
    // ---- the views (flat agent blocks) the tile intersects; the last ones may lie in env e0 + 1
    const int ia0 = (int)fdiv((uint32_t)off0, cfg.div_cgg);
    const int last = off0 + OBS_TILE - 1;  // relative to env e0's block
    const int ia1 = last < OB ? (int)fdiv((uint32_t)last, cfg.div_cgg) : N + (int)fdiv((uint32_t)(last - OB), cfg.div_cgg);
    // ---- loads, all issued before anything waits: the grids' dwords and (lane k: view k) the viewer's two position bytes
    const uint32_t* g0 = (const uint32_t*)(p.grid + (size_t)e0 * cfg.GS);
    const uint32_t* g1 = (const uint32_t*)(p.grid + (size_t)(two ? e0 + 1 : e0) * cfg.GS);
    uint32_t c0[4], c1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {  // G <= 32: at most 256 grid dwords (uniform conditions)
        c0[j] = (GW > WAVE * j) ? g0[min(lane + WAVE * j, GW - 1)] : 0u;
        c1[j] = (GW > WAVE * j && two) ? g1[min(lane + WAVE * j, GW - 1)] : 0u;
    }
    uint32_t own_bit = 0xFFFFFFFFu;
    {
        const int fa = ia0 + lane;
        const bool nxt = fa >= N;
        if (fa <= ia1 && !(nxt && !two)) {
            const int ik = nxt ? fa - N : fa;
            const uint16_t rc = *(const uint16_t*)(p.rec + (size_t)(nxt ? e0 + 1 : e0) * cfg.RS + cfg.off_pos + 2 * ik);
            const int r = (int)(int8_t)(rc & 0xFFu), c = (int)(int8_t)(rc >> 8);
            const int cell = ((reverse_mask >> ik) & 1u) ? flip_cell(cfg, r * G + c, r, c) : r * G + c;
            own_bit = (uint32_t)((nxt ? OB : 0) + ik * CGG - off0 + cell);
        }
    }
    for (int q = lane; q < OBS_TILE / 8 / 16; q += WAVE) ((u32x4_t*)bits)[q] = u32x4_t{0u, 0u, 0u, 0u};
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
    __builtin_amdgcn_wave_barrier();
    if (own_bit < (uint32_t)OBS_TILE) atomicOr(bits + (own_bit >> 5), 1u << (own_bit & 31u));
    if (!(OBS_ABLATE & 4)) {
        const uint32_t lut0_lo = (uint32_t)cfg.chan_lut[0], lut0_hi = (uint32_t)(cfg.chan_lut[0] >> 32);
        const uint32_t lut1_lo = (uint32_t)cfg.chan_lut[1], lut1_hi = (uint32_t)(cfg.chan_lut[1] >> 32);
#pragma unroll 1
        for (int j = 0; j * WAVE < GW; j++) {  // uniform; one pass for G <= 16
            const int w = lane + WAVE * j;
            const uint32_t cw0 = j == 0 ? c0[0] : (j == 1 ? c0[1] : (j == 2 ? c0[2] : c0[3]));
            const uint32_t cw1 = j == 0 ? c1[0] : (j == 1 ? c1[1] : (j == 2 ? c1[2] : c1[3]));
            // once per grid dword: for its 4 cells the plain and the flipped index and, per env, the channel code for either
            // viewing team (code NONE also for an empty cell)
            uint32_t plain[4], flipped[4], code0 = 0, code1 = 0;  // code0 / code1: byte b = (team-1 code << 4 | team-0 code) of env e0 / e0 + 1
            {
                int r = (int)fdiv((uint32_t)(w * 4), cfg.div_g), c = w * 4 - r * G;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int cell = w * 4 + b;
                    const bool in = w < GW && cell < GG;
                    plain[b] = (uint32_t)cell;
                    flipped[b] = (uint32_t)flip_cell(cfg, cell, r, c);
                    const uint32_t v0 = in ? (cw0 >> (8 * b)) & 0xFFu : 0u, v1 = in ? (cw1 >> (8 * b)) & 0xFFu : 0u;
                    const uint32_t sh0 = 4 * (v0 & 7u), sh1 = 4 * (v1 & 7u);
                    const uint32_t k00 = v0 ? (((v0 & 8u) ? lut0_hi : lut0_lo) >> sh0) & 15u : CTF_TILE_NONE;
                    const uint32_t k01 = v0 ? (((v0 & 8u) ? lut1_hi : lut1_lo) >> sh0) & 15u : CTF_TILE_NONE;
                    const uint32_t k10 = v1 ? (((v1 & 8u) ? lut0_hi : lut0_lo) >> sh1) & 15u : CTF_TILE_NONE;
                    const uint32_t k11 = v1 ? (((v1 & 8u) ? lut1_hi : lut1_lo) >> sh1) & 15u : CTF_TILE_NONE;
                    code0 |= (k00 | (k01 << 4)) << (8 * b);
                    code1 |= (k10 | (k11 << 4)) << (8 * b);
                    if (++c == G) { c = 0; r++; }
                }
            }
#pragma unroll 1
            for (int fa = ia0; fa <= ia1; fa++) {  // uniform: the views
                const bool nxt = fa >= N;
                if (nxt && !two) break;
                const int ik = nxt ? fa - N : fa;
                const int tsh = (int)((cfg.team_mask >> ik) & 1u) * 4;
                const bool rev = (reverse_mask >> ik) & 1u;
                const int base = (nxt ? OB : 0) + ik * CGG - off0;  // the view's block starts at tile bit `base` (may be negative)
                const uint32_t codes = nxt ? code1 : code0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t code = (codes >> (8 * b + tsh)) & 15u;
                    const uint32_t bit = (uint32_t)(base + (int)(code * (uint32_t)GG + (rev ? flipped[b] : plain[b])));
                    if (code != CTF_TILE_NONE && bit < (uint32_t)OBS_TILE) atomicOr(bits + (bit >> 5), 1u << (bit & 31u));
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(LGKM_ONLY);  // the atomic ORs have landed
    __builtin_amdgcn_wave_barrier();
    // ---- stream: TILE / 1 KiB store instructions, their bitmap halfwords read first (obs_bytes % 16 == 0: whole chunks)
    {
        const uint16_t* hb = (const uint16_t*)bits;
        const unsigned long long lo = (unsigned long long)env_base * (unsigned long long)OB + lo_local;
        const unsigned long long left = (unsigned long long)cfg.n_envs * (unsigned long long)OB - lo;
        const int nchunk = (int)((left < (unsigned long long)OBS_TILE ? left : (unsigned long long)OBS_TILE) >> 4);
        uint8_t* out = obs + lo;
        uint32_t h[OBS_TILE / 1024];
#pragma unroll
        for (int u = 0; u < OBS_TILE / 1024; u++) h[u] = hb[u * WAVE + lane];
#pragma unroll
        for (int u = 0; u < OBS_TILE / 1024; u++) {
            const int k = u * WAVE + lane;
            const u32x4_t v = {expand4(h[u], 0), expand4(h[u], 1), expand4(h[u], 2), expand4(h[u], 3)};
            if (k < nchunk && !(OBS_ABLATE & 2)) *(u32x4_t*)(out + ((size_t)k << 4)) = v;
        }
    }
    // ---- metadata rows of the env whose block starts in this tile (behind the stores: off the stream's path)
    const bool starts0 = off0 == 0;
    if (meta && (starts0 || two) && !(OBS_ABLATE & 8)) {
        const int em = starts0 ? e0 : e0 + 1;
        uint8_t* srec = wl + OBS_TILE / 8;
        uint16_t* mv = (uint16_t*)(srec + cfg.RS);
        uint16_t* mstage = (uint16_t*)(srec + cfg.RS + OBS_MV_BYTES);
        const uint32_t recw = ((const uint32_t*)(p.rec + (size_t)em * cfg.RS))[min(lane, cfg.RS / 4 - 1)];
        if (lane == 0) { mv[40] = 0x3C00u; mv[41] = 0u; }
        const ObsSlots none = {{0u, 0u, 0u, 0u}};
        obs_build_env(cfg, p, em, recw, 0u, srec, mv, mstage, p.meta_lut, nullptr, none, reverse_mask, lane, false, meta);
    }
}

// ------------------------------------------------------------------------------------------------
// compact observation: one byte per (agent, cell)
// ------------------------------------------------------------------------------------------------
// standardise_state's planes 1..C-1 are one-hot per cell (plane k+1 = (relabelled grid == TILES_USED[k]),
// gridworld_ctf.py:990-1001) and plane 0 has the single own-position bit, so a whole [C][G][G] block is carried by
// G*G bytes: codes[e][i][d] = index of the tile plane that is 1 at cell d of agent i's view (0 = none), bit 7 = plane 0.
// A policy that consumes the codes directly (ctf_policy.hip) never needs the 14x larger one-hot block.
// Per env a wave builds the (viewer team, reversed?) code maps that are in use (at most 4) in LDS; every agent's row is
// its map plus the own-position bit: an output dword is two aligned dwords of the map funnel-shifted (v_alignbyte_b32).
// The metadata rows leave in the same launch.  Per-wave LDS: obs_build_env's metadata scratch ...
// ... and the metadata scratch of obs_build_env when the launch also writes the metadata rows:
// [rec RS][mvals 96][meta staging][meta LUT][grid GS][self cell u16[16]][maps 4 x (GGp + 4)]
__host__ __device__ inline int codes_map_stride(int GG) { return ((GG + 3) & ~3) + 4; }  // + 4: the funnel read's second dword
__host__ __device__ inline int codes_wave_bytes(int GS, int RS, int GG, int N, int M) {
    return RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M) + obs_meta_lut_bytes(N, M) + GS + 32 + 4 * codes_map_stride(GG);
}

template <bool DWORDS>
__global__ void __launch_bounds__(256) k_observe_codes(DevCfg cfg, DevPtrs p, uint8_t* __restrict__ codes,
                                                       uint16_t* __restrict__ meta, uint16_t* __restrict__ selfcells,
                                                       uint32_t reverse_mask) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int wpb = blockDim.x / WAVE;
    const int N = cfg.N, M = cfg.M, G = cfg.G, GG = cfg.GG, MS = codes_map_stride(GG);
    uint8_t* wl = (uint8_t*)lds + wave * codes_wave_bytes(cfg.GS, cfg.RS, GG, N, M);
    uint8_t* srec = wl;
    uint16_t* mv = (uint16_t*)(wl + cfg.RS);
    uint16_t* mstage = (uint16_t*)(wl + cfg.RS + OBS_MV_BYTES);
    uint8_t* mlut = wl + cfg.RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M);
    uint8_t* sgrid = mlut + obs_meta_lut_bytes(N, M);
    uint16_t* selfc = (uint16_t*)(sgrid + cfg.GS);
    uint8_t* maps = (uint8_t*)(selfc + 16);
    const ObsSlots slots = obs_slots(cfg, reverse_mask);
    uint32_t slot_pack = 0;  // 2 bits per agent
    for (int i = 0; i < N; i++) slot_pack |= (uint32_t)(cfg.team[i] * 2 + (int)((reverse_mask >> i) & 1u)) << (2 * i);
    if (meta) obs_meta_lut(cfg, mlut, mv, lane);
    const int row = N * GG;
    const int GW = cfg.GS / 4;
    const int rec_lane = min(lane, cfg.RS / 4 - 1), grid_lane = min(lane, GW - 1);
    const int e_first = blockIdx.x * wpb + wave, e_stride = gridDim.x * wpb;
    // a wave's envs are a dependent chain of load -> build -> store: the next env's state is loaded while this one is built
    uint32_t recw_n = 0, cells_n = 0;
    if (e_first < cfg.n_envs) {
        recw_n = ((const uint32_t*)(p.rec + (size_t)e_first * cfg.RS))[rec_lane];
        cells_n = ((const uint32_t*)(p.grid + (size_t)e_first * cfg.GS))[grid_lane];
    }
    for (int e = e_first; e < cfg.n_envs; e += e_stride) {
        const uint32_t recw = recw_n, cells = cells_n;
        const int e_next = min(e + e_stride, cfg.n_envs - 1);
        recw_n = ((const uint32_t*)(p.rec + (size_t)e_next * cfg.RS))[rec_lane];
        cells_n = ((const uint32_t*)(p.grid + (size_t)e_next * cfg.GS))[grid_lane];
        if (lane < GW) ((uint32_t*)sgrid)[lane] = cells;
        for (int w = lane + WAVE; w < GW; w += WAVE) ((uint32_t*)sgrid)[w] = ((const uint32_t*)(p.grid + (size_t)e * cfg.GS))[w];  // G > 16
        // parks the record in LDS and, when asked for, writes this env's metadata rows
        obs_build_env(cfg, p, e, recw, 0u, srec, mv, mstage, mlut, nullptr, slots, reverse_mask, lane, false, meta);
        if (!codes && !selfcells) continue;
        if (lane < N) {
            const int8_t* ps = (const int8_t*)(srec + cfg.off_pos);
            const int r = ps[2 * lane], c = ps[2 * lane + 1];
            const uint16_t sc = (uint16_t)(((reverse_mask >> lane) & 1u) ? flip_cell(cfg, r * G + c, r, c) : r * G + c);
            selfc[lane] = sc;
            if (selfcells) selfcells[(size_t)e * N + lane] = sc;  // where bit 7 sits in this agent's row
        }
        if (!codes) continue;
#pragma unroll
        for (int slot = 0; slot < 4; slot++) {
            if (!slots.a[slot]) continue;  // uniform
            const uint64_t lut = pin64(cfg.chan_lut[slot >> 1]);
            for (int d = lane; d < GG; d += WAVE) {
                const int r = (int)fdiv((uint32_t)d, cfg.div_g), c = d - r * G;
                const int src = (slot & 1) ? flip_cell(cfg, d, r, c) : d;  // every flip is an involution
                const uint32_t v = sgrid[src];
                uint32_t code = (((v & 8u) ? (uint32_t)(lut >> 32) : (uint32_t)lut) >> (4 * (v & 7u))) & 15u;
                if (v == 0 || code == CTF_TILE_NONE) code = 0;
                maps[slot * MS + d] = (uint8_t)code;
            }
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
        uint8_t* out = codes + (size_t)e * row;
        for (int q = lane; q < (row + 3) / 4; q += WAVE) {
            const int f0 = 4 * q;
            const int i0 = (int)fdiv((uint32_t)f0, cfg.div_gg_row), d0 = f0 - i0 * GG;
            if (DWORDS) {
                // bytes d0 .. d0 + 3 of agent i0's row: two aligned dwords of its map, funnel-shifted (past the row's end the
                // map's padding shows up and is masked off below)
                const uint32_t* m32 = (const uint32_t*)(maps + ((slot_pack >> (2 * i0)) & 3u) * MS);
                uint32_t word = __builtin_amdgcn_alignbyte(m32[(d0 >> 2) + 1], m32[d0 >> 2], (uint32_t)(d0 & 3));
                const uint32_t sd = (uint32_t)((int)selfc[i0] - d0);
                if (sd < 4u) word |= 0x80u << (8 * sd);
                const int nb = GG - d0;  // bytes left in this agent's row
                if (nb < 4) {            // the dword runs into the next agent's row (never past the env: N * GG is a multiple of 4)
                    const int i1 = i0 + 1;
                    uint32_t nxt = *(const uint32_t*)(maps + ((slot_pack >> (2 * i1)) & 3u) * MS);
                    const uint32_t s1 = selfc[i1];
                    if (s1 < (uint32_t)(4 - nb)) nxt |= 0x80u << (8 * s1);
                    word = (word & ((1u << (8 * nb)) - 1u)) | (nxt << (8 * nb));
                }
                ((uint32_t*)out)[q] = word;
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int f = f0 + b;
                    if (f < row) {
                        const int i = (d0 + b >= GG) ? i0 + 1 : i0, d = (d0 + b >= GG) ? d0 + b - GG : d0 + b;
                        uint32_t v = maps[((slot_pack >> (2 * i)) & 3u) * MS + d];
                        if (d == (int)selfc[i]) v |= 0x80u;
                        out[f] = (uint8_t)v;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(LGKM_ONLY);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// bulk export of the evaluation counters
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void k_export_counters(DevCfg cfg, DevPtrs p, int32_t* metrics, int32_t* captures, int32_t* steps) {
    const int MN = CTF_N_METRICS * cfg.N;
    const size_t total = (size_t)cfg.n_envs * MN;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        if (metrics) metrics[idx] = cfg.log_metrics ? p.metrics[idx] : 0;
        if (idx < (size_t)cfg.n_envs) {
            const int32_t* misc = (const int32_t*)(p.rec + idx * cfg.RS + cfg.off_misc);
            if (captures) { captures[2 * idx] = misc[1]; captures[2 * idx + 1] = misc[2]; }
            if (steps) steps[idx] = misc[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bulk hand-over of the twin MT19937 states, stream-ordered (the facade's global-RNG contract; checkpoints)
// ------------------------------------------------------------------------------------------------
// Standard form per env and generator: 624 state words + the position (0..624), as random.getstate()[1] /
// np.random.get_state()[1:3] give them.  One block per env.
extern "C" __global__ void __launch_bounds__(256) k_import_rng(DevCfg cfg, DevPtrs p, const uint32_t* __restrict__ py,
                                                               const uint32_t* __restrict__ np_) {
    const int e = blockIdx.x;
    const uint32_t* src[2] = {py, np_};
    uint32_t* dst[2] = {p.mt_py, p.mt_np};
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;
        const uint32_t* in = src[k] + (size_t)e * (CTF_MT_N + 1);
        uint32_t* out = dst[k] + (size_t)e * CTF_MT_N;
        for (int i = threadIdx.x; i < CTF_MT_N; i += blockDim.x) out[i] = in[i];
        // lazy flag clear: words [pos, 624) are output as they stand
        if (threadIdx.x == 0) p.rngpos[2 * e + k] = in[CTF_MT_N] > CTF_MT_N ? CTF_MT_N : in[CTF_MT_N];
    }
}
// A lazily regenerated block ([0, pos) new, [pos, 624) old) is finished first: word i needs the OLD a[i], a[i + 1] and
// a[i + 397] for i < 227, the NEW a[i - 227] beyond — chunks of 227 words, every chunk read completely before it is written.
extern "C" __global__ void __launch_bounds__(256) k_export_rng(DevCfg cfg, DevPtrs p, uint32_t* __restrict__ py, uint32_t* __restrict__ np_) {
    __shared__ uint32_t a[CTF_MT_N];
    const int e = blockIdx.x, t = threadIdx.x;
    uint32_t* dst[2] = {py, np_};
    const uint32_t* src[2] = {p.mt_py, p.mt_np};
    for (int k = 0; k < 2; k++) {
        if (!dst[k]) continue;  // uniform
        const uint32_t packed = p.rngpos[2 * e + k];
        const int pos = (int)(packed & CTF_POS_MASK);
        for (int i = t; i < CTF_MT_N; i += blockDim.x) a[i] = src[k][(size_t)e * CTF_MT_N + i];
        __syncthreads();
        if (packed & CTF_LAZY_BIT) {
            for (int c = pos; c < CTF_MT_N; c += 227) {
                const int i = c + t;
                const bool on = t < 227 && i < CTF_MT_N;
                uint32_t v = 0;
                if (on) {
                    const uint32_t x0 = a[i], x1 = a[i + 1 == CTF_MT_N ? 0 : i + 1];
                    const uint32_t m = a[i + 397 >= CTF_MT_N ? i + 397 - CTF_MT_N : i + 397];
                    const uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
                    v = m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
                __syncthreads();
                if (on) a[i] = v;
                __syncthreads();
            }
        }
        uint32_t* out = dst[k] + (size_t)e * (CTF_MT_N + 1);
        for (int i = t; i < CTF_MT_N; i += blockDim.x) out[i] = a[i];
        if (t == 0) out[CTF_MT_N] = (uint32_t)pos;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic actions: Philox4x32-10 (Salmon et al. 2011), one lane per (env, block of 8 agents)
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void k_random_actions(DevCfg cfg, int8_t* actions, uint64_t seed, uint32_t step, uint32_t env_offset) {
    const int nblk = (cfg.N + 7) / 8;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cfg.n_envs * nblk) return;
    const int e = idx / nblk, blk = idx - e * nblk;
    uint32_t c0 = env_offset + (uint32_t)e, c1 = step, c2 = (uint32_t)blk, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (blk * 8 + j < cfg.N) {
            const uint32_t h = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            actions[(size_t)e * cfg.N + blk * 8 + j] = (int8_t)((h * 9u) >> 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (called from ctf_abi.hip)
// ------------------------------------------------------------------------------------------------
extern "C" hipError_t ctf_launch_seed(const DevCfg& cfg, const DevPtrs& p, const uint64_t* py, const uint64_t* np_, hipStream_t st) {
    hipLaunchKernelGGL(k_seed, dim3((cfg.n_envs + 63) / 64), dim3(64), 0, st, cfg, p, py, np_);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_reset(const DevCfg& cfg, const DevPtrs& p, const uint8_t* mask, int init_perm, hipStream_t st) {
    hipLaunchKernelGGL(k_reset, dim3(cfg.n_envs), dim3(WAVE), 0, st, cfg, p, mask, init_perm);
    return hipGetLastError();
}
template <bool METRICS, int W>
static void launch_step_w(const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64, uint8_t* done,
                          uint32_t flags, hipStream_t st) {
    constexpr int EPW = WAVE / W;
    const dim3 grid((cfg.n_envs + EPW - 1) / EPW), block(WAVE);
    const size_t sh = (size_t)EPW * step_slot_bytes(cfg.GS, cfg.RS, cfg.N, METRICS);
    hipLaunchKernelGGL((k_step<METRICS, W>), grid, block, sh, st, cfg, p, actions, rw32, rw64, done, flags);
}
template <bool METRICS>
static void launch_step_m(int w, const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64,
                          uint8_t* done, uint32_t flags, hipStream_t st) {
    if (w <= 1) launch_step_w<METRICS, 1>(cfg, p, actions, rw32, rw64, done, flags, st);
    else if (w == 2) launch_step_w<METRICS, 2>(cfg, p, actions, rw32, rw64, done, flags, st);
    else if (w == 4) launch_step_w<METRICS, 4>(cfg, p, actions, rw32, rw64, done, flags, st);
    else launch_step_w<METRICS, 8>(cfg, p, actions, rw32, rw64, done, flags, st);
}
// lanes per env: the power of two that covers the larger opponents list (<= 8), so one tag pass per agent turn
static int step_lanes(const DevCfg& cfg) {
    const int mo = cfg.n_opp[0] > cfg.n_opp[1] ? cfg.n_opp[0] : cfg.n_opp[1];
    int w = mo <= 1 ? 1 : (mo <= 2 ? 2 : (mo <= 4 ? 4 : 8));
    if (cfg.step_lanes_override) w = cfg.step_lanes_override;  // profiling knob (CTF_STEP_W), results are identical
    return w;
}
extern "C" hipError_t ctf_launch_step(const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64,
                                      uint8_t* done, uint32_t flags, hipStream_t st) {
    const int w = step_lanes(cfg);
    if (cfg.log_metrics) launch_step_m<true>(w, cfg, p, actions, rw32, rw64, done, flags, st);
    else launch_step_m<false>(w, cfg, p, actions, rw32, rw64, done, flags, st);
    return hipGetLastError();
}
#if STEP_TRACE
// what the runtime thinks fits: blocks of k_step<true, 4> per CU at `lds` bytes of dynamic LDS, and the device's LDS per CU
extern "C" int ctf_debug_step_occupancy(int lds, int* blocks_per_cu, int* lds_per_cu, int* lds_per_block) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    *lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    *lds_per_block = (int)prop.sharedMemPerBlock;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_step<true, 4>, WAVE, (size_t)lds);
}
#endif
static int obs_reserve_blocks() {
    static const int v = [] { const char* e = getenv("CTF_OBS_RESERVE_BLOCKS"); return e ? atoi(e) : 0; }();
    return v < 0 ? 0 : v;
}
extern "C" hipError_t ctf_launch_observe(const DevCfg& cfg, const DevPtrs& p, uint8_t* obs, uint16_t* meta, uint32_t reverse_mask,
                                         int n_cus, hipStream_t st) {
    const uintptr_t a = (uintptr_t)obs;
    const int align = ((cfg.obs_bytes % 16) == 0 && (a % 16) == 0) ? 16 : (((cfg.obs_bytes % 4) == 0 && (a % 4) == 0) ? 4 : 1);
    const char* tenv = getenv("CTF_OBS_TILES");  // 0 / 1: never / whenever possible (tests, profiling)
    const bool tiles = obs && align == 16 && cfg.tile_k > 0 && (tenv ? atoi(tenv) != 0 : OBS_TILES_DEFAULT);
    if (tiles) {
        // one wave per tile, 4 independent waves per block; tile_bx blocks per group of tile_k envs (whose blocks fill tile_tpg tiles)
        const int wpb = CTF_OBS_TILE_WPB;
        const size_t sh = (size_t)wpb * tiles_wave_bytes(cfg.RS, cfg.N, cfg.M);
        const char* xenv = getenv("CTF_OBS_XCD");  // 0: launch-order tiles (profiling); default: XCD-contiguous
        const uint32_t xcd_map = xenv ? (atoi(xenv) != 0) : 1u;
        hipLaunchKernelGGL(k_observe_tiles, dim3((unsigned)cfg.tile_nb), dim3(wpb * WAVE), sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
        return hipGetLastError();
    }
    // waves per block: 4 unless one env's bitmap is so large that 4 of them would crowd the CU's LDS
    const int per_wave = obs_wave_bytes(cfg.RS, cfg.N, cfg.M, cfg.obs_bytes);
    int wpb = 4;
    while (wpb > 1 && wpb * per_wave > 40 * 1024) wpb >>= 1;
    const size_t sh = (size_t)wpb * per_wave;
    int blocks = (cfg.n_envs + wpb - 1) / wpb;
    // the CU's 32-wave limit, grid-stride beyond that; a few block slots stay free so that a concurrent small kernel
    // (the RCCL all-gather of the rollout tensors) can start beside this launch instead of behind it
    int cap = n_cus * (32 / wpb);
    if (const char* ov = getenv("CTF_OBS_BLOCKS_PER_CU")) {  // profiling only: occupancy scaling of the render
        const int v = atoi(ov);
        if (v >= 1 && v < 32 / wpb) cap = n_cus * v;
    }
    if (cap > 64) cap -= obs_reserve_blocks();
    if (blocks > cap) blocks = cap;
    const dim3 grid(blocks), block(wpb * WAVE);
    const char* xenv = getenv("CTF_OBS_XCD");  // 0: plain grid-stride split of the envs (profiling)
    const uint32_t xcd_map = (blocks % 8 == 0 && (xenv ? atoi(xenv) != 0 : true)) ? 1u : 0u;
    if (align == 16) hipLaunchKernelGGL(k_observe<16>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
    else if (align == 4) hipLaunchKernelGGL(k_observe<4>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
    else hipLaunchKernelGGL(k_observe<1>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_observe_codes(const DevCfg& cfg, const DevPtrs& p, uint8_t* codes, uint16_t* meta, uint16_t* selfcells,
                                               uint32_t reverse_mask, int n_cus, hipStream_t st) {
    const int per_wave = codes_wave_bytes(cfg.GS, cfg.RS, cfg.GG, cfg.N, cfg.M);
    int wpb = 4;
    while (wpb > 1 && wpb * per_wave > 40 * 1024) wpb >>= 1;
    const size_t sh = (size_t)wpb * per_wave;
    int blocks = (cfg.n_envs + wpb - 1) / wpb;
    if (blocks > n_cus * 8) blocks = n_cus * 8;
    const bool dwords = ((cfg.N * cfg.GG) % 4) == 0 && ((uintptr_t)codes % 4) == 0;
    if (dwords) hipLaunchKernelGGL(k_observe_codes<true>, dim3(blocks), dim3(wpb * WAVE), sh, st, cfg, p, codes, meta, selfcells, reverse_mask);
    else hipLaunchKernelGGL(k_observe_codes<false>, dim3(blocks), dim3(wpb * WAVE), sh, st, cfg, p, codes, meta, selfcells, reverse_mask);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_export_counters(const DevCfg& cfg, const DevPtrs& p, int32_t* metrics, int32_t* captures, int32_t* steps,
                                                 hipStream_t st) {
    const size_t total = (size_t)cfg.n_envs * CTF_N_METRICS * cfg.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_export_counters, dim3(blocks), dim3(256), 0, st, cfg, p, metrics, captures, steps);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_import_rng(const DevCfg& cfg, const DevPtrs& p, const uint32_t* py, const uint32_t* np_, hipStream_t st) {
    hipLaunchKernelGGL(k_import_rng, dim3(cfg.n_envs), dim3(256), 0, st, cfg, p, py, np_);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_export_rng(const DevCfg& cfg, const DevPtrs& p, uint32_t* py, uint32_t* np_, hipStream_t st) {
    hipLaunchKernelGGL(k_export_rng, dim3(cfg.n_envs), dim3(256), 0, st, cfg, p, py, np_);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_random_actions(const DevCfg& cfg, int8_t* actions, uint64_t seed, uint32_t step,
                                                uint32_t env_offset, hipStream_t st) {
    const int n = cfg.n_envs * ((cfg.N + 7) / 8);
    hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, st, cfg, actions, seed, step, env_offset);
    return hipGetLastError();
}
