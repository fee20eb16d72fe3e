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

// MT19937 with LAZY in-place regeneration: instead of rewriting all 624 words when the block is
// exhausted (a 624-iteration burst that would serialise a divergent wave), word i of the next block
// is produced from a[i], a[i+1], a[i+397] at the moment it is consumed.  Words [0,pos) then belong
// to the new block and [pos,624) to the old one, which is exactly the order the standard in-place
// algorithm visits them in, so the output stream is identical.  `lazy` is 0 only between a state
// import (all words already tempered-ready) and the first wrap.
struct Mt {
    uint32_t* a;
    uint32_t pos, lazy;
};
__device__ __forceinline__ Mt mt_open(uint32_t* base, uint32_t packed) {
    Mt g;
    g.a = base;
    g.pos = packed & CTF_POS_MASK;
    g.lazy = (packed & CTF_LAZY_BIT) ? 1u : 0u;
    return g;
}
__device__ __forceinline__ uint32_t mt_close(const Mt& g) { return g.pos | (g.lazy ? CTF_LAZY_BIT : 0u); }
__device__ __forceinline__ uint32_t mt_next(Mt& g) {
    uint32_t i = g.pos;
    if (i >= CTF_MT_N) { i = 0; g.lazy = 1; }
    uint32_t v;
    if (g.lazy) {
        uint32_t i1 = (i + 1 == CTF_MT_N) ? 0u : i + 1;
        uint32_t im = (i + 397 >= CTF_MT_N) ? i + 397 - CTF_MT_N : i + 397;
        uint32_t y = (g.a[i] & 0x80000000u) | (g.a[i1] & 0x7fffffffu);
        v = g.a[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g.a[i] = v;
    } else {
        v = g.a[i];
    }
    g.pos = i + 1;
    return mt_temper(v);
}
// CPython random._randbelow_with_getrandbits(n): k = n.bit_length(); draw k bits until < n
__device__ __forceinline__ uint32_t py_randbelow(Mt& g, uint32_t n) {
    uint32_t sh = (uint32_t)__clz((int)n);  // 32 - bit_length
    uint32_t r = mt_next(g) >> sh;
    while (r >= n) r = mt_next(g) >> sh;
    return r;
}
// NumPy legacy random_sample()
__device__ __forceinline__ double np_rand(Mt& g) {
    uint32_t a = mt_next(g) >> 5, b = mt_next(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
// NumPy legacy randint(k), k >= 1: masked rejection on one 32-bit word; k == 1 draws nothing
__device__ __forceinline__ uint32_t np_randint(Mt& g, uint32_t k) {
    uint32_t rng = k - 1;
    if (rng == 0) return 0;
    uint32_t mask = 0xFFFFFFFFu >> __clz((int)rng);
    uint32_t v;
    do { v = mt_next(g) & mask; } while (v > rng);
    return v;
}

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
    misc[3] = 0;  // done
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
        uint32_t* v = p.vis + (size_t)e * cfg.N * cfg.GS;
        for (int w = lane; w < cfg.N * cfg.GS; w += WAVE) v[w] = 0;
        __syncthreads();
        if (lane < cfg.N) v[lane * cfg.GS + cfg.start_pos[lane][0] * cfg.G + cfg.start_pos[lane][1]] = 1;  // :473
    }
}

// ------------------------------------------------------------------------------------------------
// step — one lane per env, 64 envs per 64-thread block, state staged through LDS
// ------------------------------------------------------------------------------------------------
// LDS slot of one env (bytes): [grid GS][rec RS][actions 16][metric deltas u16 13*N (METRICS)] ; the slot
// stride in dwords is odd so that the 64 lanes' same-offset accesses fall in distinct banks.
__host__ __device__ inline int step_slot_bytes(int GS, int RS, int N, bool metrics) {
    int b = GS + RS + 16 + (metrics ? ((CTF_N_METRICS * N * 2 + 3) & ~3) : 0);
    if (((b / 4) & 1) == 0) b += 4;
    return b;
}

struct StepCtx {
    uint8_t* sg;   // grid  (LDS)
    uint8_t* sr;   // record (LDS)
    uint16_t* sm;  // metric deltas (LDS) or nullptr
};

__device__ __forceinline__ double ld_hp(const StepCtx& s, int a) {
    const uint32_t* q = (const uint32_t*)(s.sr + 8 * a);
    return __hiloint2double((int)q[1], (int)q[0]);
}
__device__ __forceinline__ void st_hp(const StepCtx& s, int a, double v) {
    uint32_t* q = (uint32_t*)(s.sr + 8 * a);
    q[0] = (uint32_t)__double2loint(v);
    q[1] = (uint32_t)__double2hiint(v);
}

template <bool METRICS>
__device__ __forceinline__ void metric_add(const DevCfg& cfg, const StepCtx& s, int m, int a, int v) {
    if (METRICS) s.sm[m * cfg.N + a] += (uint16_t)v;
}

// respawn, gridworld_ctf.py:761-794
__device__ __forceinline__ void respawn(const DevCfg& cfg, const StepCtx& s, Mt& np_, int o, uint32_t& status) {
    const int G = cfg.G, team = cfg.team[o];
    const int x = cfg.spawn_pos[team][0], y = cfg.spawn_pos[team][1];
    const int r0 = x - 1 > 0 ? x - 1 : 0, c0 = y - 1 > 0 ? y - 1 : 0;
    const int r1 = x + 2 < G ? x + 2 : G, c1 = y + 2 < G ? y + 2 : G;
    // open cells of the (clipped) 3x3 window as a bitmask in row-major candidate order
    uint32_t open = 0;
    int k = 0;
    for (int r = r0; r < r1; r++)
        for (int c = c0; c < c1; c++) {
            if (s.sg[r * G + c] == 0) { open |= 1u << ((r - r0) * 3 + (c - c0)); k++; }
        }
    if (k == 0) { status |= CTF_ST_NO_RESPAWN; return; }
    uint32_t rnd = np_randint(np_, (uint32_t)k);
    // rnd-th set bit
    uint32_t bits = open;
    for (uint32_t t = 0; t < rnd; t++) bits &= bits - 1;
    int sel = __ffs((int)bits) - 1;
    int nr = x + sel / 3 - 1, nc = y + sel % 3 - 1;  // "-1" even when the window was clipped (:775)
    if (nr < 0 || nc < 0) { status |= CTF_ST_SPAWN_EDGE; nr = nr < 0 ? nr + G : nr; nc = nc < 0 ? nc + G : nc; }
    int8_t* ps = (int8_t*)(s.sr + cfg.off_pos);
    const int orow = ps[2 * o], ocol = ps[2 * o + 1];
    s.sg[orow * G + ocol] = 0;
    s.sg[nr * G + nc] = (uint8_t)(4 + cfg.type[o] + 4 * team);
    ps[2 * o] = (int8_t)nr;
    ps[2 * o + 1] = (int8_t)nc;
    st_hp(s, o, cfg.type_hp[cfg.type[o]]);
    if (s.sr[cfg.off_flag + o]) {
        s.sr[cfg.off_flag + o] = 0;
        if (cfg.drop_flag) s.sg[orow * G + ocol] = (uint8_t)(12 + (1 - team));
        else s.sg[cfg.flag_pos[1 - team][0] * G + cfg.flag_pos[1 - team][1]] = (uint8_t)(12 + (1 - team));
    }
}

// The body of GridworldCtf.step for ONE env whose state sits in LDS.  Returns rewards through `rw`.
template <bool METRICS>
__device__ __forceinline__ void env_step(const DevCfg& cfg, const StepCtx& s, const int8_t* act, Mt& py, Mt& np_,
                                         uint32_t& status, double* rw /*[CTF_MAX_AGENTS], statically indexed*/) {
    const int N = cfg.N, G = cfg.G;
    int32_t* misc = (int32_t*)(s.sr + cfg.off_misc);
    int8_t* ps = (int8_t*)(s.sr + cfg.off_pos);
    uint8_t* flag = s.sr + cfg.off_flag;
    uint8_t* perm = s.sr + cfg.off_perm;
    int16_t* inv = (int16_t*)(s.sr + cfg.off_inv);

    misc[0] += 1;  // env_step_count
    uint32_t cap_mask = 0, resp_mask = 0, cap_team = 0;

    // dice_roll (:734-742): random.shuffle(self._arr)
    for (int i = N - 1; i >= 1; i--) {
        uint32_t j = py_randbelow(py, (uint32_t)i + 1u);
        uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }

    for (int k = 0; k < N; k++) {
        const int a = perm[k];
        const int type = cfg.type[a], team = cfg.team[a];
        int action = act[a];
        if (action < 0 || action >= CTF_N_ACTIONS) { status |= CTF_ST_BAD_ACTION; action = 4; }

        // ---- act (:700-732)
        int dr = 0, dc = 0;
        {
            const int base = action <= 4 ? action : action - 5;
            const int scale = action <= 4 ? 1 : (type == 2 ? 2 : (type == 3 ? 1 : 0));
            dr = (base == 0 ? -1 : (base == 1 ? 1 : 0)) * scale;
            dc = (base == 2 ? 1 : (base == 3 ? -1 : 0)) * scale;
        }
        int pr = ps[2 * a], pc = ps[2 * a + 1];
        const int nr = pr + dr, nc = pc + dc;
        if (nr >= 0 && nr < G && nc >= 0 && nc < G) {
            const int cell = s.sg[nr * G + nc];
            if (cell == 0 && (action <= 3 || (action >= 5 && type == 2 && (ld_hp(s, a) - cfg.vault_cost) > cfg.vault_min))) {
                // movement_handler (:569-612)
                s.sg[pr * G + pc] = 0;
                s.sg[nr * G + nc] = (uint8_t)(4 + type + 4 * team);
                pr = nr; pc = nc;
                ps[2 * a] = (int8_t)nr;
                ps[2 * a + 1] = (int8_t)nc;
                const int ofr = cfg.flag_pos[1 - team][0], ofc = cfg.flag_pos[1 - team][1];
                const int hfr = cfg.flag_pos[team][0], hfc = cfg.flag_pos[team][1];
                if (cheb(nr, nc, ofr, ofc) <= 1 && s.sg[ofr * G + ofc] == 12 + (1 - team)) {  // pickup: flag cell -> BLOCK
                    flag[a] = 1;
                    s.sg[ofr * G + ofc] = 1;
                    metric_add<METRICS>(cfg, s, CTF_M_FLAG_PICKUPS, a, 1);
                }
                if (cheb(nr, nc, hfr, hfc) <= 1 && flag[a] == 1) {  // capture
                    if (!cfg.home_flag_capture || s.sg[hfr * G + hfc] == 12 + team) {
                        flag[a] = 0;
                        s.sg[ofr * G + ofc] = (uint8_t)(12 + (1 - team));
                        misc[1 + team] += 1;
                        metric_add<METRICS>(cfg, s, CTF_M_FLAG_CAPTURES, a, 1);
                        cap_mask |= 1u << a;
                        cap_team |= 1u << team;
                    }
                }
                if (action >= 5 && type == 2) st_hp(s, a, ld_hp(s, a) - cfg.vault_cost);  // update_vaulter_hp
            } else if (action >= 5 && type == 3 && inv[a] > 0 && cell == 0 &&
                       cheb(nr, nc, cfg.spawn_pos[team][0], cfg.spawn_pos[team][1]) > 1 &&
                       cheb(nr, nc, cfg.spawn_pos[1 - team][0], cfg.spawn_pos[1 - team][1]) > 1) {
                s.sg[nr * G + nc] = 2;  // add_block (:614-634)
                inv[a] -= 1;
                if (METRICS) {
                    metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID, a, 1);
                    metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID_DIST_OWN_FLAG, a,
                                        cheb(pr, pc, cfg.capture_pos[team][0], cfg.capture_pos[team][1]));
                    metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_LAID_DIST_OPP_FLAG, a,
                                        cheb(pr, pc, cfg.capture_pos[1 - team][0], cfg.capture_pos[1 - team][1]));
                }
            } else if (action < 5 && type == 3 && (cell == 2 || cell == 3)) {
                if (cell == 2) {
                    s.sg[nr * G + nc] = 3;  // mine_block (:677-690)
                } else {
                    s.sg[nr * G + nc] = 0;
                    if (inv[a] < 1000) inv[a] += 1;
                    metric_add<METRICS>(cfg, s, CTF_M_BLOCKS_MINED, a, 1);
                }
            }
        }

        // ---- tagging_logic (:796-837)
        const double dmg = cfg.type_damage[type];
        if (dmg > 0) {
            double mult = 1.0;
            if (type == 1 && cheb(pr, pc, cfg.flag_pos[team][0], cfg.flag_pos[team][1]) <= 3) mult = cfg.guard_mult;
            const double hit = dmg * mult;
            const int no = cfg.n_opp[team];
            for (int q = 0; q < no; q++) {
                const int o = cfg.opp[team][q];
                const double u = np_rand(np_);  // drawn first, unconditionally
                if (u < cfg.tag_p && cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1) {
                    const double h = ld_hp(s, o) - hit;
                    st_hp(s, o, h);
                    metric_add<METRICS>(cfg, s, CTF_M_TAG_COUNT, a, 1);
                    if (h <= 0) {
                        if (flag[o] == 1) metric_add<METRICS>(cfg, s, CTF_M_FLAG_DISPOSSESSIONS, a, 1);
                        respawn(cfg, s, np_, o, status);
                        resp_mask |= 1u << a;
                        metric_add<METRICS>(cfg, s, CTF_M_RESPAWN_TAG_COUNT, a, 1);
                    }
                }
            }
        }

        // ---- metric-only section (:879-902)
        if (METRICS) {
            if (cheb(pr, pc, cfg.capture_pos[team][0], cfg.capture_pos[team][1]) <= 3)
                metric_add<METRICS>(cfg, s, CTF_M_STEPS_DEFENDING_ZONE, a, 1);
            if (cheb(pr, pc, cfg.capture_pos[1 - team][0], cfg.capture_pos[1 - team][1]) <= 3)
                metric_add<METRICS>(cfg, s, CTF_M_STEPS_ATTACKING_ZONE, a, 1);
            int adj = 0;
            for (int q = 0; q < cfg.n_opp[1 - team]; q++) {  // OPPONENTS[1-team]: own team, self included
                const int m = cfg.opp[1 - team][q];
                adj += cheb(pr, pc, ps[2 * m], ps[2 * m + 1]) <= 1;
            }
            metric_add<METRICS>(cfg, s, CTF_M_STEPS_ADJ_TEAMMATE, a, adj);
            adj = 0;
            for (int q = 0; q < cfg.n_opp[team]; q++) {
                const int o = cfg.opp[team][q];
                adj += cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1;
            }
            metric_add<METRICS>(cfg, s, CTF_M_STEPS_ADJ_OPPONENT, a, adj);
        }
    }

    // heal_agents (:839-847): a second shuffle (consumes RNG, mutates _arr), then heal everyone
    for (int i = N - 1; i >= 1; i--) {
        uint32_t j = py_randbelow(py, (uint32_t)i + 1u);
        uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    for (int a = 0; a < N; a++) {
        const double mx = cfg.type_hp[cfg.type[a]];
        double h = ld_hp(s, a);
        if (h < mx) {
            h += cfg.heal;
            st_hp(s, a, h > mx ? mx : h);
        }
    }

    // rewards: act() reward, + tagging reward, adjusted (:957-966), terminal (:920-940) — same op order
    const bool terminal = (misc[0] == cfg.game_steps);
    int winner = -1, margin = 0;
    if (terminal) {
        misc[3] = 1;
        const int c0 = misc[1], c1 = misc[2];
        margin = iabs_(c0 - c1);
        winner = c0 > c1 ? 0 : (c0 < c1 ? 1 : -1);
    }
#pragma unroll
    for (int i = 0; i < CTF_MAX_AGENTS; i++) {
        if (i < N) {
            const int team = cfg.team[i];
            double r = 0.0 + cfg.r_step;
            if ((cap_mask >> i) & 1u) r += cfg.r_capture;
            r += ((resp_mask >> i) & 1u) ? cfg.r_tag : 0.0;
            if (cfg.use_adjusted) r -= (((cap_team >> (1 - team)) & 1u) ? 1.0 : 0.0) * cfg.r_capture * cfg.punish;
            if (winner >= 0) {
                if (team == winner) r += margin * cfg.win_scalar;
                else r -= margin * cfg.loss_scalar;
            }
            rw[i] = r;
        }
    }
}

template <bool METRICS>
__global__ void __launch_bounds__(WAVE) k_step(DevCfg cfg, DevPtrs p, const int8_t* __restrict__ actions,
                                                float* __restrict__ rw32, double* __restrict__ rw64,
                                                uint8_t* __restrict__ done_out, uint32_t flags) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int env0 = blockIdx.x * WAVE;
    const int nvalid = min(WAVE, cfg.n_envs - env0);
    const int SLB = step_slot_bytes(cfg.GS, cfg.RS, cfg.N, METRICS);
    const int SLW = SLB / 4, GW = cfg.GS / 4, RW = cfg.RS / 4;
    const int AW = 4;  // action words per slot
    const int N = cfg.N;

    // ---- stage 64 envs' grids, records and actions into LDS (coalesced: one env per iteration)
    {
        const uint32_t* gsrc = (const uint32_t*)(p.grid + (size_t)env0 * cfg.GS);
        const uint32_t* rsrc = (const uint32_t*)(p.rec + (size_t)env0 * cfg.RS);
        for (int el = 0; el < nvalid; el++) {
            uint32_t* slot = lds + el * SLW;
            for (int w = lane; w < GW; w += WAVE) slot[w] = gsrc[el * GW + w];
            for (int w = lane; w < RW; w += WAVE) slot[GW + w] = rsrc[el * RW + w];
        }
        const int8_t* asrc = actions + (size_t)env0 * N;
        for (int idx = lane; idx < nvalid * N; idx += WAVE) {
            int el = idx / N, i = idx - el * N;
            ((int8_t*)(lds + el * SLW + GW + RW))[i] = asrc[idx];
        }
        if (METRICS) {
            const int MW = (CTF_N_METRICS * N * 2 + 3) / 4;
            for (int el = 0; el < nvalid; el++)
                for (int w = lane; w < MW; w += WAVE) lds[el * SLW + GW + RW + AW + w] = 0;
        }
    }
    __syncthreads();

    const int e = env0 + lane;
    if (lane < nvalid) {
        StepCtx s;
        s.sg = (uint8_t*)(lds + lane * SLW);
        s.sr = s.sg + cfg.GS;
        s.sm = METRICS ? (uint16_t*)(s.sr + cfg.RS + 16) : nullptr;
        const int8_t* act = (const int8_t*)(s.sr + cfg.RS);
        int32_t* misc = (int32_t*)(s.sr + cfg.off_misc);

        if ((flags & CTF_STEP_AUTO_RESET) && misc[3]) {
            // reset() of this env inside the step launch (not in the reference: opt-in flag)
            const uint32_t* src = (const uint32_t*)p.init_grid;
            for (int w = 0; w < GW; w++) ((uint32_t*)s.sg)[w] = src[w];
            reset_record(cfg, s.sr);
            if (METRICS) {
                int32_t* m = p.metrics + (size_t)e * CTF_N_METRICS * N;
                for (int w = 0; w < CTF_N_METRICS * N; w++) m[w] = 0;
                uint32_t* v = p.vis + (size_t)e * N * cfg.GS;
                for (int w = 0; w < N * cfg.GS; w++) v[w] = 0;
                for (int i = 0; i < N; i++) v[i * cfg.GS + cfg.start_pos[i][0] * cfg.G + cfg.start_pos[i][1]] = 1;
            }
        }

        Mt py = mt_open(p.mt_py + (size_t)e * CTF_MT_N, p.rngpos[2 * e]);
        Mt npg = mt_open(p.mt_np + (size_t)e * CTF_MT_N, p.rngpos[2 * e + 1]);
        uint32_t status = 0;
        double rw[CTF_MAX_AGENTS];
        env_step<METRICS>(cfg, s, act, py, npg, status, rw);
        p.rngpos[2 * e] = mt_close(py);
        p.rngpos[2 * e + 1] = mt_close(npg);
        if (status) atomicOr(p.status, status);

#pragma unroll
        for (int i = 0; i < CTF_MAX_AGENTS; i++) {
            if (i < N) {
                if (rw32) rw32[(size_t)e * N + i] = (float)rw[i];
                if (rw64) rw64[(size_t)e * N + i] = rw[i];
            }
        }
        if (done_out) done_out[e] = (uint8_t)misc[3];

        if (METRICS) {  // update_visitation_map (:479-486); u32 counters, exported modulo 256 (the reference's u8 wraps)
            const int8_t* ps = (const int8_t*)(s.sr + cfg.off_pos);
            uint32_t* v = p.vis + (size_t)e * N * cfg.GS;
            for (int i = 0; i < N; i++) atomicAdd(v + i * cfg.GS + ps[2 * i] * cfg.G + ps[2 * i + 1], 1u);
        }
    }
    __syncthreads();

    // ---- write the 64 envs back (coalesced)
    {
        uint32_t* gdst = (uint32_t*)(p.grid + (size_t)env0 * cfg.GS);
        uint32_t* rdst = (uint32_t*)(p.rec + (size_t)env0 * cfg.RS);
        for (int el = 0; el < nvalid; el++) {
            const uint32_t* slot = lds + el * SLW;
            for (int w = lane; w < GW; w += WAVE) gdst[el * GW + w] = slot[w];
            for (int w = lane; w < RW; w += WAVE) rdst[el * RW + w] = slot[GW + w];
        }
        if (METRICS) {
            const int MN = CTF_N_METRICS * N;
            int32_t* mdst = p.metrics + (size_t)env0 * MN;
            for (int el = 0; el < nvalid; el++) {
                const uint16_t* d = (const uint16_t*)(lds + el * SLW + GW + RW + AW);
                for (int w = lane; w < MN; w += WAVE) {
                    uint16_t inc = d[w];
                    if (inc) mdst[el * MN + w] += inc;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// observe — one wave per env at a time; 16 output bytes per lane per iteration
// ------------------------------------------------------------------------------------------------
// Per-wave LDS (bytes): [rec RS][aginfo 16 x u32][4 view slots x VS] ; a view slot holds, for one
// (viewer team, reversed?) combination, the channel code of every cell in the view's orientation,
// twice: codes[0..GG) followed by codes[0..GG) - 1.  A 16-byte output chunk that starts in plane c at
// cell0 and runs into plane c+1 is then simply 16 consecutive bytes of the slot compared with c.
__host__ __device__ inline int obs_view_bytes(int GG) { return (2 * GG + 8 + 3) & ~3; }
__host__ __device__ inline int obs_wave_bytes(int RS, int GG) { return RS + 64 + 4 * obs_view_bytes(GG); }

#define OBS_WAVES 4

template <int ALIGN>
struct OutVec;
template <>
struct OutVec<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <>
struct OutVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };

template <int ALIGN>
__global__ void __launch_bounds__(OBS_WAVES* WAVE) k_observe(DevCfg cfg, DevPtrs p, uint8_t* __restrict__ obs,
                                                              uint16_t* __restrict__ meta, uint32_t reverse_mask) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int N = cfg.N, G = cfg.G, GG = cfg.GG, C = cfg.C;
    const int VS = obs_view_bytes(GG);
    uint8_t* wl = (uint8_t*)lds + wave * obs_wave_bytes(cfg.RS, GG);
    uint8_t* srec = wl;
    uint32_t* aginfo = (uint32_t*)(wl + cfg.RS);
    uint8_t* views = wl + cfg.RS + 64;
    // which (team, reversed) view slots are needed — uniform over the launch
    uint32_t need = 0;
    for (int i = 0; i < N; i++) need |= 1u << (cfg.team[i] * 2 + ((reverse_mask >> i) & 1u));

    for (int e = blockIdx.x * OBS_WAVES + wave; e < cfg.n_envs; e += gridDim.x * OBS_WAVES) {
        // ---- record -> LDS
        {
            const uint32_t* rsrc = (const uint32_t*)(p.rec + (size_t)e * cfg.RS);
            for (int w = lane; w < cfg.RS / 4; w += WAVE) ((uint32_t*)srec)[w] = rsrc[w];
        }
        // ---- build the view slots: every lane relabels + scatters 4 cells per pass
        if (obs) {
            const uint32_t* gsrc = (const uint32_t*)(p.grid + (size_t)e * cfg.GS);
            for (int w = lane; w < cfg.GS / 4; w += WAVE) {
                uint32_t cells = gsrc[w];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int cell = w * 4 + b;
                    if (cell < GG) {
                        const uint32_t v = (cells >> (8 * b)) & 0xFFu;
                        const int r = (int)fdiv((uint32_t)cell, cfg.div_g), c = cell - r * G;
                        int fl;  // destination cell under the reversal (gridworld_ctf.py:1003-1007)
                        if (cfg.flip_axis == -1) fl = GG - 1 - cell;
                        else if (cfg.flip_axis == 0) fl = (G - 1 - r) * G + c;
                        else if (cfg.flip_axis == 1) fl = r * G + (G - 1 - c);
                        else fl = (G - 1 - c) * G + (G - 1 - r);
#pragma unroll
                        for (int slot = 0; slot < 4; slot++) {
                            if (need & (1u << slot)) {
                                const uint32_t code = (uint32_t)(cfg.chan_lut[slot >> 1] >> (4 * v)) & 15u;
                                const int dst = (slot & 1) ? fl : cell;
                                uint8_t* vs = views + slot * VS;
                                vs[dst] = (uint8_t)code;
                                vs[GG + dst] = (uint8_t)(code - 1u);
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // the wave's own LDS writes have landed before its reads below
        __builtin_amdgcn_wave_barrier();
        if (obs && lane < N) {
            const int8_t* ps = (const int8_t*)(srec + cfg.off_pos);
            const int r = ps[2 * lane], c = ps[2 * lane + 1];
            const uint32_t rev = (reverse_mask >> lane) & 1u;
            int cell = r * G + c;
            if (rev) {
                if (cfg.flip_axis == -1) cell = GG - 1 - cell;
                else if (cfg.flip_axis == 0) cell = (G - 1 - r) * G + c;
                else if (cfg.flip_axis == 1) cell = r * G + (G - 1 - c);
                else cell = (G - 1 - c) * G + (G - 1 - r);
            }
            const uint32_t slot = (uint32_t)cfg.team[lane] * 2 + rev;
            aginfo[lane] = ((uint32_t)(cfg.RS + 64 + slot * VS) << 16) | (uint32_t)cell;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();

        // ---- metadata (gridworld_ctf.py:1027-1069): one lane per f16 element
        if (meta) {
            const int32_t* misc = (const int32_t*)(srec + cfg.off_misc);
            const uint8_t* flag = srec + cfg.off_flag;
            uint16_t* mdst = meta + (size_t)e * N * cfg.M;
            for (int idx = lane; idx < N * cfg.M; idx += WAVE) {
                const int i = (int)fdiv((uint32_t)idx, cfg.div_m), k = idx - i * cfg.M;
                const int team = cfg.team[i];
                double val = 0.0;
                if (k == 0) val = (double)misc[0] / (double)cfg.game_steps;
                else if (k == 1) val = (double)(misc[1 + team] + 1) / (double)(misc[1 + (1 - team)] + 1);
                else if (k < 6) val = (k - 2 == cfg.type[i]) ? 1.0 : 0.0;
                else {
                    int who, which;
                    if (k < 8) { who = i; which = k - 6; }
                    else { who = cfg.meta_order[i][(k - 8) >> 1]; which = (k - 8) & 1; }
                    if (who >= 0) {
                        if (which) val = (double)flag[who];
                        else {
                            // the quirk at :1039-1041: hp of the agent whose INDEX is type(who), over max hp of type(who), as uint8
                            const int tv = cfg.type[who];
                            double q = 0.0;
                            if (tv < N) {
                                const uint32_t* hq = (const uint32_t*)(srec + 8 * tv);
                                q = __hiloint2double((int)hq[1], (int)hq[0]) / cfg.type_hp[tv];
                            }
                            val = (double)(uint8_t)(long long)q;
                        }
                    }
                }
                mdst[idx] = f64_to_f16(val);
            }
        }

        // ---- stream the observation block: u8 [N][C][G][G], 16 bytes per lane per iteration
        if (obs) {
            uint8_t* out = obs + (size_t)e * cfg.obs_bytes;
            const int nfull = cfg.obs_bytes >> 4;
            const int tail = cfg.obs_bytes & 15;
            const int nchunks = nfull + (tail ? 1 : 0);
            for (int k = lane; k < nchunks; k += WAVE) {
                const uint32_t o = (uint32_t)k << 4;
                const uint32_t agent = fdiv(o, cfg.div_cgg);
                const uint32_t rem = o - agent * (uint32_t)cfg.CGG;
                const uint32_t c = fdiv(rem, cfg.div_gg);
                const uint32_t cell0 = rem - c * (uint32_t)GG;
                const uint32_t info = aginfo[agent];
                const uint32_t addr = (info >> 16) + cell0;
                const uint32_t* src = (const uint32_t*)(wl + (addr & ~3u));
                const uint32_t sh = addr & 3u;
                const uint32_t w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3], w4 = src[4];
                const uint32_t cv = c * 0x01010101u;
                uint32_t x[4];
                x[0] = __builtin_amdgcn_alignbyte(w1, w0, sh) ^ cv;
                x[1] = __builtin_amdgcn_alignbyte(w2, w1, sh) ^ cv;
                x[2] = __builtin_amdgcn_alignbyte(w3, w2, sh) ^ cv;
                x[3] = __builtin_amdgcn_alignbyte(w4, w3, sh) ^ cv;
                // byte == 0  ->  1 ; all bytes < 0x80 so the subtraction never borrows across bytes
#pragma unroll
                for (int j = 0; j < 4; j++) x[j] = ((0x80808080u - x[j]) >> 7) & 0x01010101u;
                // plane 0 (own position) of this agent, or of the next agent when the chunk runs past plane C-1
                int h = -1;
                if (c == 0) h = (int)(info & 0xFFFFu) - (int)cell0;
                else if (c == (uint32_t)(C - 1) && agent + 1 < (uint32_t)N) h = GG + (int)(aginfo[agent + 1] & 0xFFFFu) - (int)cell0;
                if (h >= 0 && h < 16) {
                    const uint32_t bit = 1u << ((h & 3) * 8);
                    x[0] |= (h >> 2) == 0 ? bit : 0u;
                    x[1] |= (h >> 2) == 1 ? bit : 0u;
                    x[2] |= (h >> 2) == 2 ? bit : 0u;
                    x[3] |= (h >> 2) == 3 ? bit : 0u;
                }
                if (k < nfull) {
                    if (ALIGN >= 4) {
                        typedef typename OutVec<(ALIGN >= 16 ? 16 : 4)>::type V;
                        V v = {x[0], x[1], x[2], x[3]};
                        __builtin_nontemporal_store(v, (V*)(out + o));
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; j++) out[o + j] = (uint8_t)(x[j >> 2] >> ((j & 3) * 8));
                    }
                } else {
                    for (int j = 0; j < tail; j++) out[o + j] = (uint8_t)(x[j >> 2] >> ((j & 3) * 8));
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // this env's LDS reads are done before the next env overwrites the slots
        __builtin_amdgcn_wave_barrier();
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
extern "C" hipError_t ctf_launch_step(const DevCfg& cfg, const DevPtrs& p, const int8_t* actions, float* rw32, double* rw64,
                                      uint8_t* done, uint32_t flags, hipStream_t st) {
    const dim3 grid((cfg.n_envs + WAVE - 1) / WAVE), block(WAVE);
    if (cfg.log_metrics) {
        const size_t sh = (size_t)WAVE * step_slot_bytes(cfg.GS, cfg.RS, cfg.N, true);
        hipLaunchKernelGGL(k_step<true>, grid, block, sh, st, cfg, p, actions, rw32, rw64, done, flags);
    } else {
        const size_t sh = (size_t)WAVE * step_slot_bytes(cfg.GS, cfg.RS, cfg.N, false);
        hipLaunchKernelGGL(k_step<false>, grid, block, sh, st, cfg, p, actions, rw32, rw64, done, flags);
    }
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_observe(const DevCfg& cfg, const DevPtrs& p, uint8_t* obs, uint16_t* meta, uint32_t reverse_mask,
                                         int n_cus, hipStream_t st) {
    const size_t sh = (size_t)OBS_WAVES * obs_wave_bytes(cfg.RS, cfg.GG);
    int blocks = (cfg.n_envs + OBS_WAVES - 1) / OBS_WAVES;
    const int cap = n_cus * 8;  // 8 blocks of 4 waves per CU = the 32-wave limit; grid-stride beyond that
    if (blocks > cap) blocks = cap;
    const dim3 grid(blocks), block(OBS_WAVES * WAVE);
    const uintptr_t a = (uintptr_t)obs;
    if ((cfg.obs_bytes % 16) == 0 && (a % 16) == 0)
        hipLaunchKernelGGL(k_observe<16>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask);
    else if ((cfg.obs_bytes % 4) == 0 && (a % 4) == 0)
        hipLaunchKernelGGL(k_observe<4>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask);
    else
        hipLaunchKernelGGL(k_observe<1>, grid, block, sh, st, cfg, p, obs, meta, reverse_mask);
    return hipGetLastError();
}
extern "C" hipError_t ctf_launch_random_actions(const DevCfg& cfg, int8_t* actions, uint64_t seed, uint32_t step,
                                                uint32_t env_offset, hipStream_t st) {
    const int n = cfg.n_envs * ((cfg.N + 7) / 8);
    hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, st, cfg, actions, seed, step, env_offset);
    return hipGetLastError();
}
