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
// Profiling-only phase trace of the step kernel (tools/trace_step.py): lane 0 of every block stamps the 100 MHz wall clock at
// fixed points (0 start, 1 staged, 7 hit bits, 5 ring, 6 shuffle 1, 8/9/10 + 3 k act / tagging / metrics of turn k, 32 turns
// done, 33 shuffle 2, 34 stepped, 2 barrier, 3 written back, 4 end); never defined in the shipped build.
#ifndef STEP_TRACE
#define STEP_TRACE 0
#endif
#if STEP_TRACE
__device__ unsigned long long g_step_trace[8192][40];
#ifndef STEP_TRACE_MASK
#define STEP_TRACE_MASK 0xFFFFFFFFFFull
#endif
#define CTF_STAMP(k) do { if (((STEP_TRACE_MASK >> (k)) & 1ull) && threadIdx.x == 0 && blockIdx.x < 8192) g_step_trace[blockIdx.x][(k)] = wall_clock64(); } while (0)
#endif
#include "ctf_step_core.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// small helpers (fdiv, cheb, the SGPR-pinned config lookups, MT19937: ctf_step_core.h / ctf_mt.h)
// ------------------------------------------------------------------------------------------------
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

// py_seeds / np_seeds: device arrays [E].  After this (and the k_rng_refill(init) launch that follows it), env e ==
// random.seed(py) ; np.random.seed(np) (MT19937 mode: ring 0 = the seeded state, position 624, as CPython / NumPy hold it), or
// its two streams are the counter streams of these seeds at word 0 (counter mode).
extern "C" __global__ void k_seed(DevCfg cfg, DevPtrs p, const uint64_t* py_seeds, const uint64_t* np_seeds) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.n_envs) return;
    uint32_t* a_py = p.mt_py + (size_t)e * 2 * CTF_MT_N;
    uint32_t* a_np = p.mt_np + (size_t)e * 2 * CTF_MT_N;
    const uint64_t ps = py_seeds[e], ns = np_seeds[e];
    if (cfg.rng_mode == CTF_RNG_COUNTER) {
        for (unsigned long long blk = 0; blk < CTF_MT_N / 4; blk++) {
            ctr_block(ps, blk, 0u, a_py + 4 * blk);
            ctr_block(ns, blk, 1u, a_np + 4 * blk);
        }
        unsigned long long* ctr = p.rngctr + 6 * (size_t)e;
        ctr[0] = 0; ctr[2] = 0; ctr[4] = ps; ctr[5] = ns;  // ring 0 of either stream starts at word 0
        p.rngpos[2 * e + 0] = CTF_RP_MAKE(0, 0);
        p.rngpos[2 * e + 1] = CTF_RP_MAKE(0, 0);
    } else {
        uint32_t key[2] = {(uint32_t)ps, (uint32_t)(ps >> 32)};
        mt_init_by_array(a_py, key, key[1] ? 2 : 1);
        mt_init_genrand(a_np, (uint32_t)ns);
        p.rngpos[2 * e + 0] = CTF_RP_MAKE(CTF_MT_N, 0);  // both generators start exhausted: the first draw comes from the next block
        p.rngpos[2 * e + 1] = CTF_RP_MAKE(CTF_MT_N, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// reset
// ------------------------------------------------------------------------------------------------
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
// ring regeneration by one wave (ctf_mt.h): used by the tail blocks of k_step and by k_rng_refill
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define RNG_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F); \
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// ---- the digests of ctf_mt.h as ONE WAVE makes them, from a block's OUTPUT words (tempered, or as they are in counter mode) in
// LDS: the hit bits of 64 positions are one ballot, a nibble / top-byte dword is a handful of LDS reads.  (ring_digest /
// ring_link in ctf_mt.h are the same arithmetic one word at a time: the step kernel's safety net and the host simulator use
// those, and tests run both against each other through CTF_RNG_REFILL_EVERY.)
__device__ __forceinline__ void wave_digest(int lane, const uint32_t* T, const RingPtrs& p, int r, const RingParams& q) {
    if (q.stream == 1) {
        uint32_t* hit = p.hit + r * CTF_HB_DW;
        uint32_t t0[10], t1[10];
#pragma unroll
        for (int c = 0; c < 10; c++) {  // 624 positions = 10 x 64 (the last 16 lanes of the last pass idle): the reads first
            const int i = 64 * c + lane;
            t0[c] = T[i < CTF_MT_N ? i : 0];
            t1[c] = T[i + 1 < CTF_MT_N ? i + 1 : 0];
        }
#pragma unroll
        for (int c = 0; c < 10; c++) {
            const int i = 64 * c + lane;
            const unsigned long long m = __ballot(i < CTF_MT_N - 1 && mt_lt53(t0[c] >> 5, t1[c] >> 6, q.th, q.tl));
            if (lane < 2 && 2 * c + lane < (CTF_MT_N + 31) / 32) hit[2 * c + lane] = (uint32_t)(m >> (32 * lane));
        }
        uint32_t* nib = p.nib + r * CTF_NB_DW;
        const u32x4_t* T4 = (const u32x4_t*)T;
#pragma unroll
        for (int it = 0; it < 2; it++) {  // 78 dwords of 8 nibbles
            const int d = lane + 64 * it;
            if (d < CTF_MT_N / 8) {
                const u32x4_t a = T4[2 * d], b = T4[2 * d + 1];
                nib[d] = (a.x & 15u) | ((a.y & 15u) << 4) | ((a.z & 15u) << 8) | ((a.w & 15u) << 12) | ((b.x & 15u) << 16) | ((b.y & 15u) << 20) |
                         ((b.z & 15u) << 24) | ((b.w & 15u) << 28);
            }
        }
    } else {
        uint32_t* top = p.top + r * CTF_P8_DW;
        const u32x4_t* T4 = (const u32x4_t*)T;
#pragma unroll
        for (int it = 0; it < 3; it++) {  // 156 dwords of 4 top bytes
            const int d = lane + 64 * it;
            if (d < CTF_MT_N / 4) {
                const u32x4_t a = T4[d];
                top[d] = (a.x >> 24) | ((a.y >> 24) << 8) | ((a.z >> 24) << 16) | ((a.w >> 24) << 24);
            }
        }
    }
}
// Tc: outputs of ring c (only words 608 .. 623 are read), To: outputs of its successor ring
__device__ __forceinline__ void wave_link(int lane, const uint32_t* Tc, const uint32_t* To, const RingPtrs& p, int c, const RingParams& q) {
    if (q.stream == 1) {
        uint32_t* hc = p.hit + c * CTF_HB_DW;
        constexpr int P0 = (CTF_MT_N >> 5) * 32;  // 608: the first position of dword 19
        uint32_t w0[4], w1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // positions 608 .. 863, counted from ring c's start (dwords 19 .. 26: the array ends at 25)
            const int pc = P0 + 64 * k + lane;
            w0[k] = pc < CTF_MT_N ? Tc[pc] : To[pc - CTF_MT_N];
            w1[k] = pc + 1 < CTF_MT_N ? Tc[pc + 1] : To[pc + 1 - CTF_MT_N];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(mt_lt53(w0[k] >> 5, w1[k] >> 6, q.th, q.tl));
            const int d = (CTF_MT_N >> 5) + 2 * k + lane;
            if (lane < 2 && d < CTF_HB_DW) hc[d] = (uint32_t)(m >> (32 * lane));
        }
        uint32_t* nc = p.nib + c * CTF_NB_DW;
        if (lane < CTF_NB_MIRROR / 8) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) v |= (To[8 * lane + k] & 15u) << (4 * k);
            nc[CTF_MT_N / 8 + lane] = v;
        }
    } else {
        uint32_t* tc = p.top + c * CTF_P8_DW;
        if (lane < CTF_P8_MIRROR / 4) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (To[4 * lane + k] >> 24) << (8 * k);
            tc[CTF_MT_N / 4 + lane] = v;
        }
    }
}
// the block after `src` into `dst` (both LDS) by one wave: ring_next_block of ctf_mt.h with the three dependent chunks unrolled, every
// chunk's LDS reads issued before its arithmetic
__device__ __forceinline__ void wave_next_block(int lane, const uint32_t* src, uint32_t* dst, const RingParams& q) {
    if (q.counter_mode) {
#pragma unroll 1
        for (int b = lane; b < CTF_MT_N / 4; b += WAVE) {
            uint32_t o[4];
            ctr_block(q.seed, (q.nbase + CTF_MT_N) / 4 + (unsigned long long)b, (uint32_t)q.stream, o);
            ((u32x4_t*)dst)[b] = u32x4_t{o[0], o[1], o[2], o[3]};
        }
        RNG_WAVE_SYNC();
        return;
    }
    constexpr int M = CTF_MT_N - 397;  // 227
#pragma unroll
    for (int chunk = 0; chunk < 3; chunk++) {
        const int lo = chunk * M, hi = chunk == 2 ? CTF_MT_N - 1 : lo + M;  // [0, 227), [227, 454), [454, 623)
        uint32_t x0[4], x1[4], m[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int i = lo + lane + 64 * it, ic = i < hi ? i : lo;
            x0[it] = src[ic];
            x1[it] = src[ic + 1];
            m[it] = chunk == 0 ? src[ic + 397] : dst[ic - M];
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int i = lo + lane + 64 * it;
            if (i < hi) dst[i] = mt_twist(x0[it], x1[it], m[it]);
        }
        if (chunk == 2 && lane == 0) dst[CTF_MT_N - 1] = mt_twist(src[CTF_MT_N - 1], dst[0], dst[396]);
        RNG_WAVE_SYNC();
    }
}

// One ring, one wave: the ring the consumer has left becomes the block after the current one, with its digests, and the current
// ring is linked to it (mirror, hit bit of its last position).  src / dst: 2 x 624 words of the wave's LDS.  `init`: the CURRENT
// ring's digests are made too (after a seed or a state import).  2.5 KB read, 2.5 KB + the digests written, every access of the
// wave contiguous.
#if STEP_TRACE
#define RING_STAMP(k) CTF_STAMP(k)  // of a tail block's LAST ring (tools/trace_step.py)
#else
#define RING_STAMP(k) do { } while (0)
#endif
// the ring's words on their way into the wave (issued early: the previous ring of the same tail block is still being worked on)
struct RingIn {
    StreamFull st;
    u32x4_t a, b, c;
};
__device__ __forceinline__ RingIn ring_fetch(const DevCfg& cfg, const DevPtrs& p, int e, int stream, uint32_t flag, int lane) {
    RingIn in;
    in.st.r = ring_ptrs(p, e, stream);
    in.st.q = ring_params(cfg, p, e, stream);
    // The flag says which ring is stale (the position word may be moving); only an init pass (flag 0) has to read the position word —
    // the branch is uniform, and without it every regeneration would wait for that load before it can even address its ring: one
    // more dependent memory round trip on a 5.65 us job.
    const uint32_t uflag = (uint32_t)__builtin_amdgcn_readfirstlane((int)flag);  // (the same in every lane: one ring per wave)
    if (uflag >= 2u) in.st.cur = 1u - (uflag - 2u);
    else in.st.cur = ring_source(uflag, p.rngpos[2 * (size_t)e + stream]);
    ring_counter_params(in.st.q, p, e, stream, in.st.cur);
    const u32x4_t* gsrc = (const u32x4_t*)(in.st.r.raw + in.st.cur * CTF_MT_N);
    constexpr int NQ = CTF_MT_N / 4;  // 156 quads: two full passes of the wave and 28 lanes of a third
    in.a = gsrc[lane];
    in.b = gsrc[lane + WAVE];
    in.c = gsrc[lane + 2 * WAVE < NQ ? lane + 2 * WAVE : 0];
    return in;
}
__device__ __forceinline__ void refill_ring(const DevCfg& cfg, const DevPtrs& p, int e, int stream, const RingIn& in, int lane, uint32_t* src,
                                            uint32_t* dst, bool init) {
    StreamFull st = in.st;
    u32x4_t* gdst = (u32x4_t*)(st.r.raw + (1 - st.cur) * CTF_MT_N);
    constexpr int NQ = CTF_MT_N / 4;
    ((u32x4_t*)src)[lane] = in.a;
    ((u32x4_t*)src)[lane + WAVE] = in.b;
    if (lane + 2 * WAVE < NQ) ((u32x4_t*)src)[lane + 2 * WAVE] = in.c;
    RING_STAMP(10);
    RNG_WAVE_SYNC();
    wave_next_block(lane, src, dst, st.q);
    RING_STAMP(11);
    {   // the new block's raw words leave; both LDS copies then become OUTPUT words (of ring c only what is looked at)
        u32x4_t v[3];
#pragma unroll
        for (int it = 0; it < 3; it++) v[it] = ((const u32x4_t*)dst)[lane + WAVE * it < NQ ? lane + WAVE * it : 0];
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int d = lane + WAVE * it;
            if (d < NQ) {
                gdst[d] = v[it];
                ((u32x4_t*)dst)[d] = u32x4_t{ring_out(st.q, v[it].x), ring_out(st.q, v[it].y), ring_out(st.q, v[it].z), ring_out(st.q, v[it].w)};
            }
        }
        if (init) {
#pragma unroll
            for (int it = 0; it < 3; it++) {
                const int d = lane + WAVE * it;
                if (d < NQ) {
                    const u32x4_t w = ((const u32x4_t*)src)[d];
                    ((u32x4_t*)src)[d] = u32x4_t{ring_out(st.q, w.x), ring_out(st.q, w.y), ring_out(st.q, w.z), ring_out(st.q, w.w)};
                }
            }
        } else if (lane < 16) {
            src[(CTF_MT_N >> 5) * 32 + lane] = ring_out(st.q, src[(CTF_MT_N >> 5) * 32 + lane]);
        }
    }
    RNG_WAVE_SYNC();
    RING_STAMP(12);
    if (init) wave_digest(lane, src, st.r, (int)st.cur, st.q);
    wave_digest(lane, dst, st.r, 1 - (int)st.cur, st.q);
    RING_STAMP(13);
    wave_link(lane, src, dst, st.r, (int)st.cur, st.q);
    RING_STAMP(14);
    if (lane == 0) {
        ring_counter_store(st.q, p, e, stream, 1u - st.cur);
        p.rngready[2 * (size_t)e + stream] = 1;
    }
    RNG_WAVE_SYNC();  // the LDS copies are reused by the wave's next ring
}

// ------------------------------------------------------------------------------------------------
// step — W lanes per env (W = 1, 2, 4 or 8 >= opponents per team), 64 / W envs per 64-thread block
// ------------------------------------------------------------------------------------------------
// The per-agent loop of GridworldCtf.step is inherently sequential (later agents see earlier agents' moves, tags and
// respawns), so parallelism is across envs — but one lane per env leaves one wave per SIMD and a kernel bound by the
// dependent LDS / VALU chain.  Here the W lanes of a group run the env's control flow redundantly (state reads are LDS
// broadcasts; state writes are the same store from every lane) and split the work that is parallel inside an agent's turn:
//   - tagging: every lane holds the rand() < TAG_PROBABILITY bits of its share of the step's np.random window and evaluates
//     one opponent; __ballot finds the first hit, which is applied (possibly respawning, which consumes extra words) before
//     the remaining opponents are re-evaluated from the shifted stream position — exactly the reference's draw order;
//   - adjacency / zone metrics, healing, rewards, visitation: one agent per sub-lane;
//   - the digests of the step's random words (ctf_mt.h) arrive with the staging loads; the MT19937 blocks themselves are
//     regenerated by the launch's tail blocks.
// 64 / W envs per wave means W times more waves (4 per SIMD for the arena) to hide the latency chain.
// The logic itself — env_step, the two streams, group_step — is ctf_step_core.h.
//
#if STEP_TRACE
#define STEP_STAMP(k) CTF_STAMP(k)
extern "C" int ctf_debug_step_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_trace), sizeof(g_step_trace));
}
#else
#define STEP_STAMP(k) do { } while (0)
#endif

#define STEP_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F); \
        __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

#ifndef STEP_TAIL_PAIRS
#define STEP_TAIL_PAIRS 16  // (env, stream) pairs a tail block looks after (ring regeneration)
#endif

// A TAIL block of k_step: regenerates rings whose consumers have moved on (rngready says
// which).  It looks after STEP_TAIL_PAIRS (env, stream) pairs; a stale ring of env e is taken by the launch that finds it at
// the age (launches waited so far) with (e + age) % rng_spread == 0, because the envs' stream positions move in step (every env
// draws the same words per step, give or take a respawn): most of them leave their block in the same step, and that burst is
// spread over rng_spread launches — always before the ring is needed.  The ages live in device memory (rngage, written by the
// pair's tail block only), not in a launch argument: a launch is the same whichever step it is, so a caller may capture
// ctf_step / ctf_step_observe into a hipGraph and replay it.  A tail block touches nothing a step block reads: an env whose ring
// is stale stands at the head of its new block and looks at neither the other ring nor its mirror.
__device__ __forceinline__ void tail_block(const DevCfg& cfg, const DevPtrs& p, int tb, int lane, uint32_t* lds) {
    const int first = tb * STEP_TAIL_PAIRS;
    const bool mine = lane < STEP_TAIL_PAIRS && first + lane < 2 * cfg.n_envs;
    uint32_t flag = 1, age = 0;
    if (mine) {
        flag = p.rngready[first + lane];
        age = p.rngage[first + lane];
    }
    const uint32_t spread = (uint32_t)cfg.rng_spread;
    const bool take = flag >= 2u && (((uint32_t)((first + lane) >> 1) + age) % spread == 0u || age >= spread);
    {   // a ring that goes on waiting is a launch older; everything else is new again
        const uint32_t older = (flag >= 2u && !take) ? age + 1u : 0u;
        if (mine && older != age) p.rngage[first + lane] = (uint8_t)older;
    }
    unsigned long long work = __ballot(take);
    STEP_STAMP(0);
    int n_done = 0;
    if (work) {  // uniform
        int k = __ffsll((long long)work) - 1;
        work &= work - 1;
        RingIn cur = ring_fetch(cfg, p, (first + k) >> 1, (first + k) & 1, (uint32_t)__shfl((int)flag, k, WAVE), lane);
        for (;;) {
            // the next ring's words set out before this one is worked on (its loads land during the ~3 us of twisting and digesting)
            const int kn = work ? __ffsll((long long)work) - 1 : k;
            const bool more = work != 0;
            work &= work - 1;
            RingIn nxt = cur;
            if (more) nxt = ring_fetch(cfg, p, (first + kn) >> 1, (first + kn) & 1, (uint32_t)__shfl((int)flag, kn, WAVE), lane);
            refill_ring(cfg, p, (first + k) >> 1, (first + k) & 1, cur, lane, lds, lds + CTF_MT_N, false);
            n_done++;
            if (!more) break;
            cur = nxt;
            k = kn;
        }
    }
    STEP_STAMP(4);
#if STEP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_step_trace[blockIdx.x][5] = (unsigned long long)n_done;
#else
    (void)n_done;
#endif
}
// 16 blocks (= waves) per CU fit by LDS: the register budget is held to the matching 4 waves per SIMD (128 VGPRs)
template <bool METRICS, int W>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_step(DevCfg cfg, DevPtrs p, const int8_t* __restrict__ actions, float* __restrict__ rw32, double* __restrict__ rw64,
       uint8_t* __restrict__ done_out, uint32_t flags, int n_step_blocks) {
    constexpr int EPW = WAVE / W;  // envs per wave
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
#ifndef STEP_TAIL_FIRST
#define STEP_TAIL_FIRST 0  // 1 (measured, not shipped): the ring-regenerating blocks at the HEAD of the grid instead of its tail
#endif
    const int n_tail_blocks = (int)gridDim.x - n_step_blocks;
    const int sb = STEP_TAIL_FIRST ? (int)blockIdx.x - n_tail_blocks : (int)blockIdx.x;  // this step block's index
    if (STEP_TAIL_FIRST ? sb < 0 : sb >= n_step_blocks) {
        // ---- a TAIL block (see tail_block): these start as step blocks retire — the LDS is full until then
        tail_block(cfg, p, STEP_TAIL_FIRST ? (int)blockIdx.x : sb - n_step_blocks, lane, lds);
        return;
    }
    STEP_STAMP(0);
#if STEP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192)
        g_step_trace[blockIdx.x][39] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
    const int g = lane / W, j = lane % W;
    const int env0 = sb * EPW;
    const int nvalid = min(EPW, cfg.n_envs - env0);
    const int SLB = step_slot_bytes(cfg.GS, cfg.RS, cfg.N, METRICS);
    const int SLW = SLB / 4, GW = cfg.GS / 4, RW = cfg.RS / 4;
    const int AW = 4, WW = STEP_RNG_WORDS;  // action words, digest-window words per slot
    const int N = cfg.N;
    const int e = env0 + g;
    const bool live = g < nvalid;
    const int e_ld = live ? e : cfg.n_envs - 1;  // the idle groups of a ragged last block shadow a valid env: every load below is
                                                 // unconditional, so that the compiler can count them (a load under a branch makes
                                                 // every later wait a full drain: two serialised round trips instead of one)
    // the two stream positions: the addresses of the step's random digests depend on them, so they go first
    const uint32_t rp_py = p.rngpos[2 * e_ld], rp_np = p.rngpos[2 * e_ld + 1];
    const uint32_t ready2 = *(const uint16_t*)(p.rngready + 2 * (size_t)e_ld);

    // ---- stage the wave's envs' grids, records and actions into LDS.  Flat, coalesced 16-byte loads, the first 4 + 2 per lane
    // issued before anything waits; the digest windows' loads follow them as soon as the stream positions are there.
    const u32x4* gsrc = (const u32x4*)(p.grid + (size_t)env0 * cfg.GS);
    const u32x4* rsrc = (const u32x4*)(p.rec + (size_t)env0 * cfg.RS);
    const int GQ = GW / 4, RQ = RW / 4;  // 16-byte quads per env (GS and RS are multiples of 16)
    const int ng = nvalid * GQ, nr = nvalid * RQ;
    u32x4 gv[4], rv[2];
#pragma unroll
    for (int u = 0; u < 4; u++) gv[u] = gsrc[min(lane + WAVE * u, ng - 1)];
#pragma unroll
    for (int u = 0; u < 2; u++) rv[u] = rsrc[min(lane + WAVE * u, nr - 1)];
    const int na = nvalid * N;
    const int8_t* asrc = actions + (size_t)env0 * N;
    int8_t av[2];
#pragma unroll
    for (int u = 0; u < 2; u++) av[u] = asrc[min(lane + WAVE * u, na - 1)];
    GroupRng<W> R;
    group_issue_loads<W>(R, cfg, p, e_ld, j, live, rp_py, rp_np, ready2);
    {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = lane + WAVE * u;
            if (q < ng) {
                const int el = (int)fdiv((uint32_t)q, cfg.div_gq), w = (q - el * GQ) * 4;
                uint32_t* slot = lds + el * SLW + w;
                slot[0] = gv[u].x; slot[1] = gv[u].y; slot[2] = gv[u].z; slot[3] = gv[u].w;
            }
        }
        for (int q = lane + WAVE * 4; q < ng; q += WAVE) {  // G > 16 only
            const u32x4 v = gsrc[q];
            const int el = (int)fdiv((uint32_t)q, cfg.div_gq), w = (q - el * GQ) * 4;
            uint32_t* slot = lds + el * SLW + w;
            slot[0] = v.x; slot[1] = v.y; slot[2] = v.z; slot[3] = v.w;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = lane + WAVE * u;
            if (q < nr) {
                const int el = (int)fdiv((uint32_t)q, cfg.div_rq), w = (q - el * RQ) * 4;
                uint32_t* slot = lds + el * SLW + GW + w;
                slot[0] = rv[u].x; slot[1] = rv[u].y; slot[2] = rv[u].z; slot[3] = rv[u].w;
            }
        }
        for (int q = lane + WAVE * 2; q < nr; q += WAVE) {  // N > 8 only
            const u32x4 v = rsrc[q];
            const int el = (int)fdiv((uint32_t)q, cfg.div_rq), w = (q - el * RQ) * 4;
            uint32_t* slot = lds + el * SLW + GW + w;
            slot[0] = v.x; slot[1] = v.y; slot[2] = v.z; slot[3] = v.w;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int idx = lane + WAVE * u;
            if (idx < na) {
                const int el = (int)fdiv((uint32_t)idx, cfg.div_n), i = idx - el * N;
                ((int8_t*)(lds + el * SLW + GW + RW))[i] = av[u];
            }
        }
        for (int idx = lane + WAVE * 2; idx < na; idx += WAVE) {  // W = 1 with N > 2 only
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
    // a block is ONE wave: LDS operations of a wave complete in order, so the staged bytes only need the wave's own LDS
    // counter — a __syncthreads() would also drain every outstanding global load (the random words') and store
    STEP_LDS_SYNC();
    STEP_STAMP(1);

    uint32_t left_ring = 0;
    if (live) group_step<METRICS, W>(R, cfg, p, (uint8_t*)(lds + g * SLW), e, j, g * W, flags, rw32, rw64, done_out, left_ring);
    (void)left_ring;
    STEP_LDS_SYNC();
    STEP_STAMP(2);

    // ---- write the envs back (flat, coalesced 16-byte stores)
    {
        u32x4* gdst = (u32x4*)(p.grid + (size_t)env0 * cfg.GS);
        u32x4* rdst = (u32x4*)(p.rec + (size_t)env0 * cfg.RS);
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
    STEP_STAMP(4);
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
// one-shot tiles do not (profiles/r02_store_bw8_tiles.txt).  The price: the cell scan is repeated by
// every tile of an env (3.1 tiles per arena env) for the views the tile holds, and every wave pays the prologue — so both are
// kept lean: envs come in groups whose blocks fill a whole number of tiles (no 64-bit division), a cell's flipped index and
// its channel code for either viewing team are worked out once per grid dword, a view costs ~7 instructions per cell, and
// the metadata LUT comes from a table the host built.
// The 1-D grid is walked XCD-contiguously (block b -> logical block (b % 8) * (blocks / 8) + b / 8): each of the 8 XCDs writes
// its own eighth of the buffer front to back, 0.306 -> 0.259 ms on the arena (profiles/r02_store_bw9_xcd.txt for the bare pattern).
// Metadata rows: written by the wave whose tile holds an env's first byte.
// Used when an env's block is a multiple of 16 bytes and >= one tile and the buffer is 16-byte aligned (ctf_launch_observe).
#define OBS_TILE CTF_OBS_TILE
__host__ __device__ inline int tiles_wave_bytes(int RS, int N, int M) {
    return OBS_TILE / 8 + RS + OBS_MV_BYTES + obs_meta_stage_bytes(N, M);
}

template <int STORE_NT>  // 1 (batches whose observations exceed the memory-side cache, ctf_derive.h): the tile's stores carry the nontemporal hint
__global__ void __launch_bounds__(CTF_OBS_TILE_WPB * 64) k_observe_tiles(DevCfg cfg, DevPtrs p, uint8_t* __restrict__ obs, uint16_t* __restrict__ meta,
                                                       uint32_t reverse_mask, uint32_t xcd_map) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int N = cfg.N, M = cfg.M, G = cfg.G, GG = cfg.GG, CGG = cfg.CGG, OB = cfg.obs_bytes;
    // ---- which tile of which group of envs (uniform, 32-bit)
    // The launch is 1-D and consecutive workgroups go to consecutive XCDs (8 of them).  Block b therefore takes logical
    // block (b % 8) * (blocks / 8) + b / 8: every XCD writes ITS OWN contiguous eighth of the buffer front to back instead of
    // every XCD touching every page — a constant 8 KiB-tile fill measures 6.2 instead of 5.6 TB/s that way (profiles/r02_store_bw9_xcd.txt).
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
            if (k < nchunk && !(OBS_ABLATE & 2)) {
                if (STORE_NT) __builtin_nontemporal_store(v, (u32x4_t*)(out + ((size_t)k << 4)));
                else *(u32x4_t*)(out + ((size_t)k << 4)) = v;
            }
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
// the bulk ring refill (ctf_mt.h) and the hand-over of the generator states
// ------------------------------------------------------------------------------------------------
#define RNG_PAIRS_PER_WAVE 16  // (env, stream) pairs a wave looks after: their flags arrive in one load

// One wave per ring to regenerate: the ring the consumer has left becomes the block after the current one, with its digests, and
// the current ring is linked to it (mirror, hit bit of its last position).  Envs [e0, e0 + count).  `init`: every stream of
// the range is treated as not ready and the CURRENT ring's digests are made first (after a seed or a state import).
// Whole blocks, staged in LDS: 2.5 KB read, 2.5 KB + the digests written per ring, every access of a wave contiguous.
extern "C" __global__ void __launch_bounds__(256) k_rng_refill(DevCfg cfg, DevPtrs p, int e0, int count, int init) {
    __shared__ uint32_t sh[4][2 * CTF_MT_N];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    uint32_t* src = sh[wave];
    uint32_t* dst = src + CTF_MT_N;
    const int first = (blockIdx.x * 4 + wave) * RNG_PAIRS_PER_WAVE;  // pair t = env e0 + t / 2, stream t % 2
    if (first >= 2 * count) return;
    uint32_t flag = 1;
    if (lane < RNG_PAIRS_PER_WAVE && first + lane < 2 * count) flag = init ? 0u : p.rngready[2 * (size_t)e0 + first + lane];
    const bool todo = flag != 1u;
    unsigned long long work = __ballot(todo);
    while (work) {  // uniform
        const int k = __ffsll((long long)work) - 1;
        work &= work - 1;
        const int t = first + k, e = e0 + (t >> 1), stream = t & 1;
        refill_ring(cfg, p, e, stream, ring_fetch(cfg, p, e, stream, (uint32_t)__shfl((int)flag, k, WAVE), lane), lane, src, dst, init != 0);
    }
}

// Standard form per env and generator: 624 state words + the position (0..624), as random.getstate()[1] /
// np.random.get_state()[1:3] give them — which is what ring `cur` and the stream position ARE.  One block per env: block b
// handles env e0 + b and record b of the arrays.  An import is followed by k_rng_refill(init) over the same envs.
extern "C" __global__ void __launch_bounds__(256) k_import_rng(DevCfg cfg, DevPtrs p, const uint32_t* __restrict__ py,
                                                               const uint32_t* __restrict__ np_, int e0) {
    const int e = e0 + (int)blockIdx.x, t = threadIdx.x;
    const uint32_t* src[2] = {py, np_};
    uint32_t* dst[2] = {p.mt_py, p.mt_np};
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;  // uniform
        const uint32_t* in = src[k] + (size_t)blockIdx.x * (CTF_MT_N + 1);
        uint32_t* out = dst[k] + (size_t)e * 2 * CTF_MT_N;
        for (int i = t; i < CTF_MT_N; i += blockDim.x) out[i] = in[i];
        if (t == 0) p.rngpos[2 * e + k] = CTF_RP_MAKE(in[CTF_MT_N] > CTF_MT_N ? CTF_MT_N : in[CTF_MT_N], 0);
    }
}
extern "C" __global__ void __launch_bounds__(256) k_export_rng(DevCfg cfg, DevPtrs p, uint32_t* __restrict__ py, uint32_t* __restrict__ np_, int e0) {
    const int e = e0 + (int)blockIdx.x, t = threadIdx.x;
    uint32_t* dst[2] = {py, np_};
    const uint32_t* src[2] = {p.mt_py, p.mt_np};
    for (int k = 0; k < 2; k++) {
        if (!dst[k]) continue;  // uniform
        const uint32_t rp = p.rngpos[2 * e + k];
        const uint32_t* in = src[k] + ((size_t)e * 2 + CTF_RP_CUR(rp)) * CTF_MT_N;
        uint32_t* out = dst[k] + (size_t)blockIdx.x * (CTF_MT_N + 1);
        for (int i = t; i < CTF_MT_N; i += blockDim.x) out[i] = in[i];
        if (t == 0) out[CTF_MT_N] = CTF_RP_POS(rp);
    }
}
// counter mode: (words consumed from the `random` stream, ... from the np.random stream) of every env
extern "C" __global__ void k_get_counters(DevCfg cfg, DevPtrs p, unsigned long long* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.n_envs) return;
    for (int k = 0; k < 2; k++) {
        const uint32_t rp = p.rngpos[2 * e + k];
        out[2 * (size_t)e + k] = p.rngctr[6 * (size_t)e + 2 * k + CTF_RP_CUR(rp)] + CTF_RP_POS(rp);
    }
}
// ... and the way back (a checkpoint restore; followed by k_rng_refill(init)): ring 0 = the block that holds word n
extern "C" __global__ void k_set_counters(DevCfg cfg, DevPtrs p, const unsigned long long* in) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cfg.n_envs) return;
    unsigned long long* ctr = p.rngctr + 6 * (size_t)e;
    for (int k = 0; k < 2; k++) {
        const unsigned long long n = in[2 * (size_t)e + k], blk = n / CTF_MT_N;
        ctr[2 * k] = blk * CTF_MT_N;
        uint32_t* a = (k ? p.mt_np : p.mt_py) + (size_t)e * 2 * CTF_MT_N;
        for (unsigned long long b = 0; b < CTF_MT_N / 4; b++) ctr_block(ctr[4 + k], blk * (CTF_MT_N / 4) + b, (uint32_t)k, a + 4 * b);
        p.rngpos[2 * e + k] = CTF_RP_MAKE((uint32_t)(n - blk * CTF_MT_N), 0);
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
                          uint32_t flags, bool with_tail, hipStream_t st) {
    constexpr int EPW = WAVE / W;
    const int nstep = (cfg.n_envs + EPW - 1) / EPW;
    const int ntail = with_tail ? (2 * cfg.n_envs + STEP_TAIL_PAIRS - 1) / STEP_TAIL_PAIRS : 0;
    const dim3 grid(nstep + ntail), block(WAVE);
    size_t sh = (size_t)EPW * step_slot_bytes(cfg.GS, cfg.RS, cfg.N, METRICS);
    if (sh < 2 * CTF_MT_N * 4) sh = 2 * CTF_MT_N * 4;  // a tail block stages two rings
    hipLaunchKernelGGL((k_step<METRICS, W>), grid, block, sh, st, cfg, p, actions, rw32, rw64, done, flags, nstep);
}
template <bool METRICS>
static void launch_step_m(int w, const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64,
                          uint8_t* done, uint32_t flags, bool with_tail, hipStream_t st) {
    if (w <= 1) launch_step_w<METRICS, 1>(cfg, p, actions, rw32, rw64, done, flags, with_tail, st);
    else if (w == 2) launch_step_w<METRICS, 2>(cfg, p, actions, rw32, rw64, done, flags, with_tail, st);
    else if (w == 4) launch_step_w<METRICS, 4>(cfg, p, actions, rw32, rw64, done, flags, with_tail, st);
    else launch_step_w<METRICS, 8>(cfg, p, actions, rw32, rw64, done, flags, with_tail, st);
}
// lanes per env: the power of two that covers the larger opponents list (<= 8), so one tag pass per agent turn — and more
// (up to 8) as long as the waves that makes stay within 8 per CU (half of what is resident at once): fewer envs per wave then,
// i.e. a shorter divergent chain per wave (0_the_split at 4 096 envs: 2 -> 8 lanes, 0.0312 -> 0.0280 ms per env-step; the whole
// table, every width at 4 096 .. 131 072 envs of both workloads: profiles/r05_step_lanes_sweep.txt).  Results do not depend on it.
static int step_lanes(const DevCfg& cfg) {
    const int mo = cfg.n_opp[0] > cfg.n_opp[1] ? cfg.n_opp[0] : cfg.n_opp[1];
    int w = mo <= 1 ? 1 : (mo <= 2 ? 2 : (mo <= 4 ? 4 : 8));
    while (w < 8 && (long long)cfg.n_envs * (2 * w) / WAVE <= 8LL * cfg.n_cus) w *= 2;
    if (cfg.step_lanes_override) w = cfg.step_lanes_override;  // profiling / test knob (CTF_STEP_W)
    return w;
}
// with_tail: the ring regeneration rides at the tail of this launch.  Nothing in the launch depends on how many went before it
// (see tail_block), so a captured launch can be replayed.
extern "C" hipError_t ctf_launch_step(const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64,
                                      uint8_t* done, uint32_t flags, int with_tail, hipStream_t st) {
    const int w = step_lanes(cfg);
    const bool tail = with_tail && cfg.rng_refill_every;
    if (cfg.log_metrics) launch_step_m<true>(w, cfg, p, actions, rw32, rw64, done, flags, tail, st);
    else launch_step_m<false>(w, cfg, p, actions, rw32, rw64, done, flags, tail, st);
    return hipGetLastError();
}
extern "C" int ctf_step_blocks(const DevCfg& cfg) { return (cfg.n_envs + WAVE / step_lanes(cfg) - 1) / (WAVE / step_lanes(cfg)); }
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
static int observe_align(const DevCfg& cfg, const uint8_t* obs) {
    const uintptr_t a = (uintptr_t)obs;
    return ((cfg.obs_bytes % 16) == 0 && (a % 16) == 0) ? 16 : (((cfg.obs_bytes % 4) == 0 && (a % 4) == 0) ? 4 : 1);
}
// the one rule by which a render into `obs` is the tile kernel (1) or the wave-per-env kernel (0)
extern "C" int ctf_observe_uses_tiles(const DevCfg& cfg, const uint8_t* obs) {
    const char* tenv = getenv("CTF_OBS_TILES");  // 0 / 1: never / whenever possible (tests, profiling)
    return obs && observe_align(cfg, obs) == 16 && cfg.tile_k > 0 && (tenv ? atoi(tenv) != 0 : OBS_TILES_DEFAULT);
}
extern "C" hipError_t ctf_launch_observe(const DevCfg& cfg, const DevPtrs& p, uint8_t* obs, uint16_t* meta, uint32_t reverse_mask,
                                         int n_cus, hipStream_t st) {
    const int align = observe_align(cfg, obs);
    const bool tiles = ctf_observe_uses_tiles(cfg, obs) != 0;
    if (tiles) {
        // one wave per tile, 4 independent waves per block; tile_bx blocks per group of tile_k envs (whose blocks fill tile_tpg tiles)
        const int wpb = CTF_OBS_TILE_WPB;
        const size_t sh = (size_t)wpb * tiles_wave_bytes(cfg.RS, cfg.N, cfg.M);
        const char* xenv = getenv("CTF_OBS_XCD");  // 0: launch-order tiles (profiling); default: XCD-contiguous
        const uint32_t xcd_map = xenv ? (atoi(xenv) != 0) : 1u;
        if (cfg.obs_store_nt) hipLaunchKernelGGL(k_observe_tiles<1>, dim3((unsigned)cfg.tile_nb), dim3(wpb * WAVE), sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
        else hipLaunchKernelGGL(k_observe_tiles<0>, dim3((unsigned)cfg.tile_nb), dim3(wpb * WAVE), sh, st, cfg, p, obs, meta, reverse_mask, xcd_map);
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
// envs [e0, e0 + count): record b of the arrays belongs to env e0 + b
extern "C" hipError_t ctf_launch_import_rng(const DevCfg& cfg, const DevPtrs& p, const uint32_t* py, const uint32_t* np_, int e0, int count,
                                            hipStream_t st) {
    hipLaunchKernelGGL(k_import_rng, dim3(count), dim3(256), 0, st, cfg, p, py, np_, e0);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_export_rng(const DevCfg& cfg, const DevPtrs& p, uint32_t* py, uint32_t* np_, int e0, int count, hipStream_t st) {
    hipLaunchKernelGGL(k_export_rng, dim3(count), dim3(256), 0, st, cfg, p, py, np_, e0);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_rng_refill(const DevCfg& cfg, const DevPtrs& p, int e0, int count, int init, hipStream_t st) {
    const int waves = (2 * count + RNG_PAIRS_PER_WAVE - 1) / RNG_PAIRS_PER_WAVE;
    hipLaunchKernelGGL(k_rng_refill, dim3((waves + 3) / 4), dim3(256), 0, st, cfg, p, e0, count, init);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_get_counters(const DevCfg& cfg, const DevPtrs& p, unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL(k_get_counters, dim3((cfg.n_envs + 63) / 64), dim3(64), 0, st, cfg, p, out);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_set_counters(const DevCfg& cfg, const DevPtrs& p, const unsigned long long* in, hipStream_t st) {
    hipLaunchKernelGGL(k_set_counters, dim3((cfg.n_envs + 63) / 64), dim3(64), 0, st, cfg, p, in);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_random_actions(const DevCfg& cfg, int8_t* actions, uint64_t seed, uint32_t step,
                                                uint32_t env_offset, hipStream_t st) {
    const int n = cfg.n_envs * ((cfg.N + 7) / 8);
    hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, st, cfg, actions, seed, step, env_offset);
    return hipGetLastError();
}
