// ctf_step_core.h — GridworldCtf.step() for ONE env, executed by the W lanes of its group (reference gridworld_ctf.py:849-918).
//
// Included by ctf_kernels.hip (device code, W = 1, 2, 4 or 8) and — with CTF_HOSTSIM defined and W = 1, where a "group" is a
// single thread and every cross-lane primitive is the identity — by tests/hostsim/, which runs exactly this logic on the
// CPU against the oracle (a unit test of the kernel's logic; the product has no CPU path).
//
// Random numbers: both streams are MT19937 in run-ahead form (ctf_mt.h): every word a step can consume is final in memory
// when the kernel starts, so the step's random words arrive with its staging loads:
//   np.random (tagging_logic's rand() per opponent :815, respawn's randint :771): the group loads the next 2 * P + 16
//     words at once (P = rand() draws of a step without respawns).  Lane j owns the word pairs ("slots") s with
//     s % W == j and works out, for BOTH alignments of a rand() on its slots, whether rand() < TAG_PROBABILITY — one bit per
//     stream position — plus the low 4 bits of every tempered word (all randint() of a <= 9-cell spawn window needs).
//     An agent's turn then only tests bits: the rand() of opponent q sits at slot (X >> 1) + q, owned by a different lane for
//     every q, whatever the respawns so far have shifted the position X by.  Positions beyond the covered window (> 16
//     words of respawn draws in one step, or a team-size the window does not cover) fall back to direct loads.
//   random (the two shuffles :734-742, :844): 32 tempered words in an LDS ring + 16 more in registers that top the
//     ring up after the first shuffle; a reload from memory if that ever runs out.
// The words consumed are replaced by their successors (the twist) at the END of the step: mt_produce_*.
#pragma once
#include "ctf_device.h"
#include "ctf_mt.h"

#ifdef CTF_HOSTSIM
#include <string.h>
#define CTF_DEV static inline
#define CTF_MEMBER inline
#define CTF_WAVE 1
static inline unsigned long long ctf_ballot(bool p) { return p ? 1ull : 0ull; }
static inline int ctf_shfl(int v, int) { return v; }
static inline int ctf_clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int ctf_ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int ctf_popc(uint32_t x) { return __builtin_popcount(x); }
static inline double ctf_hilo2double(uint32_t hi, uint32_t lo) {
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    double d;
    memcpy(&d, &b, 8);
    return d;
}
static inline uint64_t ctf_double_bits(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    return b;
}
static inline int pin(int v) { return v; }
static inline uint64_t pin64(uint64_t v) { return v; }
static inline double pind(double v) { return v; }
static inline void ctf_atomic_add_u32(uint32_t* p, uint32_t v) { *p += v; }
static inline void ctf_atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
static inline void ctf_fence_agent() {}
#else
#define CTF_DEV __device__ __forceinline__
#define CTF_MEMBER __device__ __forceinline__
#define CTF_WAVE 64
CTF_DEV unsigned long long ctf_ballot(bool p) { return __ballot(p); }
CTF_DEV int ctf_shfl(int v, int src) { return __shfl(v, src, CTF_WAVE); }
CTF_DEV int ctf_clz(uint32_t x) { return __clz((int)x); }
CTF_DEV int ctf_ffs(uint32_t x) { return __ffs((int)x); }
CTF_DEV int ctf_popc(uint32_t x) { return __popc(x); }
CTF_DEV double ctf_hilo2double(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
CTF_DEV uint64_t ctf_double_bits(double d) { return (uint64_t)__double_as_longlong(d); }
// pin*(): pass a kernel-argument value through an empty asm with an SGPR constraint so that it is an opaque SGPR value.  Without
// this the compiler rewrites "team ? cfg.x[1] : cfg.x[0]" into ONE load from a lane-selected kernarg
// address, i.e. a dependent vector-memory access in the middle of the per-agent loop.
CTF_DEV int pin(int v) {
    asm("" : "+s"(v));  // zero instructions: just makes the value an opaque SGPR operand
    return v;
}
CTF_DEV uint64_t pin64(uint64_t v) {
    asm("" : "+s"(v));
    return v;
}
CTF_DEV double pind(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    asm("" : "+s"(b));
    return __longlong_as_double((long long)b);
}
CTF_DEV void ctf_atomic_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
CTF_DEV void ctf_atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }
CTF_DEV void ctf_fence_agent() { __threadfence(); }
#endif

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
CTF_DEV uint32_t fdiv(uint32_t n, FastDiv d) { return (uint32_t)(((uint64_t)n * d.m) >> d.s); }
CTF_DEV int iabs_(int x) { return x < 0 ? -x : x; }
CTF_DEV int cheb(int r0, int c0, int r1, int c1) {
    int a = iabs_(r0 - r1), b = iabs_(c0 - c1);
    return a > b ? a : b;
}

// lane-divergent config lookups: bit-field extracts / selects on SGPR-resident values (no memory traffic)
CTF_DEV int cfg_team(const DevCfg& c, int a) { return (int)(((uint32_t)pin((int)c.team_mask) >> a) & 1u); }
CTF_DEV int cfg_type(const DevCfg& c, int a) { return (int)(((uint32_t)pin((int)c.type_pack) >> (2 * a)) & 3u); }
CTF_DEV int cfg_opp(const DevCfg& c, int team, int q) {
    const uint64_t p0 = pin64(c.opp_pack[0]), p1 = pin64(c.opp_pack[1]);
    return (int)(((team ? p1 : p0) >> (4 * q)) & 15u);
}
CTF_DEV int cfg_nopp(const DevCfg& c, int team) { return team ? pin(c.n_opp[1]) : pin(c.n_opp[0]); }
CTF_DEV double sel4(const double* t, int k) {
    const double t0 = pind(t[0]), t1 = pind(t[1]), t2 = pind(t[2]), t3 = pind(t[3]);
    return k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3));
}
#define TSEL(arr, team, k) ((team) ? pin((int)(arr)[1][k]) : pin((int)(arr)[0][k]))

// Writes the reset record of one env at `sr` (any address space) — everything except `perm`.
template <typename BytePtr>
CTF_DEV void reset_record(const DevCfg& cfg, BytePtr sr) {
    for (int i = 0; i < cfg.N; i++) {
        const uint64_t hb = ctf_double_bits(cfg.type_hp[cfg.type[i]]);
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

// Profiling-only ablations of the step kernel (results become wrong; never defined in the shipped build):
//   bit0 no tagging, bit1 no metric section, bit2 no shuffles, bit3 no act, bit4 no visitation log, bit5 no metric flush,
//   bit6 no state write-back, bit7 no hit-bit precompute, bit8 no production (the consumed words are not replaced)
#ifndef STEP_ABLATE
#define STEP_ABLATE 0
#endif

template <int W>
struct Log2;
template <> struct Log2<1> { static constexpr int v = 0; };
template <> struct Log2<2> { static constexpr int v = 1; };
template <> struct Log2<4> { static constexpr int v = 2; };
template <> struct Log2<8> { static constexpr int v = 3; };

typedef uint32_t mt_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t mt_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t mt_u32x2 __attribute__((ext_vector_type(2), aligned(4)));

// n contiguous words (n = 1, 2 or a multiple of 4) from a 4-byte aligned address
template <int NW>
CTF_DEV void mt_load_words(const uint32_t* src, uint32_t* out) {
    if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NW / 4; q++) {
            const mt_u32x4 v = *(const mt_u32x4*)(src + 4 * q);
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
    } else if constexpr (NW == 2) {
        const mt_u32x2 v = *(const mt_u32x2*)src;
        out[0] = v.x; out[1] = v.y;
    } else {
#pragma unroll
        for (int k = 0; k < NW; k++) out[k] = src[k];
    }
}
CTF_DEV uint32_t mt_wrap(uint32_t i) { return i >= CTF_MT_N ? i - CTF_MT_N : i; }  // i < 2 * 624

// ------------------------------------------------------------------------------------------------
// np.random: hit bits and randint nibbles of the step's window
// ------------------------------------------------------------------------------------------------
#ifndef NP_CH_MAX
#define NP_CH_MAX 12  // slots (word pairs) per lane covered at most; 24 W words
#endif
#ifndef NP_SLACK
#define NP_SLACK 16   // words covered beyond the 2 P a step without respawns consumes
#endif

template <int W>
struct NpRegs {  // the window as loaded: slot c * W + j = words 2 s, 2 s + 1 and the first word of the next slot
    uint32_t w[NP_CH_MAX][3];
};

struct NpStream {
    const uint32_t* a;     // this env's ring (global memory)
    uint32_t pos;          // ring position at the start of the step
    uint32_t X;            // words consumed so far in this step
    uint32_t slots;        // slots covered: W * CH (window = words [0, 2 * slots])
    uint32_t mine;         // bit 2 c + par: rand() at word 2 (c W + j) + par is < TAG_PROBABILITY
    uint32_t nib[(2 * NP_CH_MAX + 7) / 8];  // nibble 2 c + par: low 4 bits of the tempered word 2 (c W + j) + par
    uint32_t th, tl;       // TAG_PROBABILITY * 2^53, rounded up, split at bit 26
};

CTF_DEV int np_chunks(const DevCfg& cfg, int W) {  // uniform over the launch
    const int need = 2 * pin(cfg.np_pairs) + NP_SLACK;
    const int ch = (need + 2 * W - 1) / (2 * W);
    return ch < NP_CH_MAX ? ch : NP_CH_MAX;
}
CTF_DEV bool np_lt53(const NpStream& g, uint32_t hi27, uint32_t lo26) { return hi27 < g.th || (hi27 == g.th && lo26 < g.tl); }

template <int W>
CTF_DEV void np_issue_loads(NpRegs<W>& r, const uint32_t* a, uint32_t pos, int j, int CH) {
#pragma unroll
    for (int c = 0; c < NP_CH_MAX; c++) {
        if (c < CH) {  // uniform
            const mt_u32x3 v = *(const mt_u32x3*)(a + pos + 2 * (c * W + j));  // contiguous through the mirror: < 624 + 24 W + 1
            r.w[c][0] = v.x; r.w[c][1] = v.y; r.w[c][2] = v.z;
        }
    }
}
template <int W>
CTF_DEV void np_setup(NpStream& g, const NpRegs<W>& r, const DevCfg& cfg, const uint32_t* a, uint32_t pos, int CH) {
    g.a = a; g.pos = pos; g.X = 0;
    g.slots = (uint32_t)(W * CH);
    g.th = (uint32_t)pin((int)cfg.tag_th); g.tl = (uint32_t)pin((int)cfg.tag_tl);
    g.mine = 0;
#pragma unroll
    for (int k = 0; k < (2 * NP_CH_MAX + 7) / 8; k++) g.nib[k] = 0;
    if (STEP_ABLATE & 128) return;
#pragma unroll
    for (int c = 0; c < NP_CH_MAX; c++) {
        if (c < CH) {  // uniform
            const uint32_t t0 = mt_temper(r.w[c][0]), t1 = mt_temper(r.w[c][1]), t2 = mt_temper(r.w[c][2]);
            const bool even = np_lt53(g, t0 >> 5, t1 >> 6), odd = np_lt53(g, t1 >> 5, t2 >> 6);
            g.mine |= (even ? (1u << (2 * c)) : 0u) | (odd ? (2u << (2 * c)) : 0u);
            g.nib[c >> 2] |= ((t0 & 15u) | ((t1 & 15u) << 4)) << (8 * (c & 3));
        }
    }
}
// tempered word at offset x of the step's stream, straight from memory (beyond the covered window: rare)
CTF_DEV uint32_t np_word_slow(const NpStream& g, uint32_t x) {
    uint32_t i = g.pos + x;
    while (i >= CTF_MT_N) i -= CTF_MT_N;
    return mt_temper(g.a[i]);
}
// NumPy legacy randint(k), k >= 1: masked rejection on one 32-bit word; k == 1 draws nothing.  k <= 9 here: 4 bits suffice.
template <int W>
CTF_DEV uint32_t np_randint(NpStream& g, int j, int gshift, uint32_t k) {
    const uint32_t rng = k - 1;
    if (rng == 0) return 0;
    const uint32_t mask = 0xFFFFFFFFu >> ctf_clz(rng);
    uint32_t v;
    do {
        const uint32_t sl = g.X >> 1, par = g.X & 1u;
        if (sl < g.slots) {
            const uint32_t idx = 2 * (sl >> Log2<W>::v) + par;  // the owner lane's nibble index
            uint32_t word = g.nib[0];
#pragma unroll
            for (int q = 1; q < (2 * NP_CH_MAX + 7) / 8; q++) word = ((idx >> 3) == (uint32_t)q) ? g.nib[q] : word;
            const uint32_t mine = (word >> (4 * (idx & 7u))) & 15u;
            v = (uint32_t)ctf_shfl((int)mine, gshift + (int)(sl & (W - 1))) & mask;
        } else {
            v = np_word_slow(g, g.X) & mask;
        }
        g.X++;
    } while (v > rng);
    return v;
}

// ------------------------------------------------------------------------------------------------
// random: the ring of the two shuffles
// ------------------------------------------------------------------------------------------------
#ifndef PY_RING
#define PY_RING 32  // tempered words in LDS (a power of two >= 8)
#endif
#ifndef PY_EXT
#define PY_EXT 16   // further words in registers until the first shuffle is done (<= PY_RING, a multiple of 8)
#endif
template <int W>
struct PyRegs {
    uint32_t w[PY_RING / W];
    uint32_t e[PY_EXT / W];
};
struct PyStream {
    const uint32_t* a;
    uint32_t* ring;  // LDS: slot x % 32 holds the tempered word at offset x of the step's stream, for x in [hi - 32, hi) at most
    uint32_t pos;
    uint32_t cur;    // words consumed so far in this step
    uint32_t hi;     // first offset the ring does not hold
};
template <int W>
CTF_DEV void py_issue_loads(PyRegs<W>& r, const uint32_t* a, uint32_t pos, int j) {
    mt_load_words<PY_RING / W>(a + pos + (PY_RING / W) * j, r.w);
    mt_load_words<PY_EXT / W>(a + pos + PY_RING + (PY_EXT / W) * j, r.e);
}
template <int W>
CTF_DEV void py_setup(PyStream& g, PyRegs<W>& r, const uint32_t* a, uint32_t* ring, uint32_t pos, int j) {
    g.a = a; g.ring = ring; g.pos = pos; g.cur = 0; g.hi = PY_RING;
#pragma unroll
    for (int k = 0; k < PY_RING / W; k++) ring[(PY_RING / W) * j + k] = mt_temper(r.w[k]);
#pragma unroll
    for (int k = 0; k < PY_EXT / W; k++) r.e[k] = mt_temper(r.e[k]);
}
// after the first shuffle: the slots it consumed take the words held back in registers
template <int W>
CTF_DEV void py_topup(PyStream& g, const PyRegs<W>& r, int j) {
    const uint32_t n_add = g.cur < PY_EXT ? g.cur : PY_EXT;
    if (g.hi != PY_RING) return;  // the ring was reloaded meanwhile (never in practice): it is ahead already
#pragma unroll
    for (int k = 0; k < PY_EXT / W; k++) {
        const uint32_t x = (uint32_t)((PY_EXT / W) * j + k);
        if (x < n_add) g.ring[x] = r.e[k];
    }
    g.hi = PY_RING + n_add;
}
// the ring ran dry (a shuffle of a large team, or very many rejections): refill it from memory at the current position
template <int W>
CTF_DEV void py_reload(PyStream& g, int j) {
    uint32_t i = g.pos + g.cur;
    while (i >= CTF_MT_N) i -= CTF_MT_N;
    uint32_t w[PY_RING / W];
    mt_load_words<PY_RING / W>(g.a + i + (PY_RING / W) * j, w);
#pragma unroll
    for (int k = 0; k < PY_RING / W; k++) g.ring[(g.cur + (uint32_t)((PY_RING / W) * j + k)) & (PY_RING - 1)] = mt_temper(w[k]);
    g.hi = g.cur + PY_RING;
}
// random.shuffle(self._arr) (:740): Fisher-Yates, j = _randbelow(i + 1) = the top bit_length(i + 1) bits of one word, drawn
// again while >= i + 1.  One loop over the WORDS: a rejected word costs one iteration, as an accepted one does.
template <int W>
CTF_DEV void py_shuffle(PyStream& g, int j, int N, uint64_t& perm) {
    int i = N - 1;
    while (i >= 1) {
        if (g.cur >= g.hi) py_reload<W>(g, j);
        const uint32_t t = g.ring[g.cur & (PY_RING - 1)];
        g.cur++;
        const uint32_t r = t >> ctf_clz((uint32_t)i + 1u);
        if (r <= (uint32_t)i) {
            const uint64_t d = ((perm >> (4 * i)) ^ (perm >> (4 * r))) & 15u;
            perm ^= (d << (4 * i)) | (d << (4 * r));
            i--;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// production: the words a step consumed are replaced by their successors one block later
// ------------------------------------------------------------------------------------------------
#ifndef PROD_TQ
#define PROD_TQ 4  // quads of 4 words per lane and batch
#endif
template <int W>
struct ProdRegs {
    uint32_t x[PROD_TQ][5], m[PROD_TQ][4];
    uint32_t rx[2], rm;
};
// One batch replaces the words [off, off + cov) of the w consumed ones: whole quads of 4 words, dealt to the lanes in turn (at
// most PROD_TQ each), and then one single word per lane.  The words of a batch never feed each other (that takes a distance of
// 227 words; a batch holds at most 4 * PROD_TQ * W + W <= 136).
template <int W>
CTF_DEV uint32_t prod_quads(uint32_t off, uint32_t w) {
    const uint32_t nq = (w - off) >> 2;
    return nq > (uint32_t)(PROD_TQ * W) ? (uint32_t)(PROD_TQ * W) : nq;
}
template <int W>
CTF_DEV uint32_t prod_cover(uint32_t off, uint32_t w) {
    const uint32_t left = w - off, c = 4 * prod_quads<W>(off, w) + (uint32_t)W;
    return left < c ? left : c;
}
template <int W>
CTF_DEV void mt_produce_load(ProdRegs<W>& r, const uint32_t* a, uint32_t pos, uint32_t off, uint32_t w, int j) {
    if (STEP_ABLATE & 256) return;
    const uint32_t nq = prod_quads<W>(off, w), cnt = prod_cover<W>(off, w);
    const uint32_t base = mt_wrap(mt_wrap(pos + off));  // pos < 624, off < 624
#pragma unroll
    for (int t = 0; t < PROD_TQ; t++) {
        const uint32_t qd = (uint32_t)(t * W + j);
        if (qd < nq) {
            const uint32_t i0 = mt_wrap(base + 4 * qd);  // 4 qd <= 124
            mt_load_words<4>(a + i0, r.x[t]);            // contiguous through the mirror
            r.x[t][4] = a[i0 + 4];
            mt_load_words<4>(a + mt_wrap(i0 + 397), r.m[t]);
        }
    }
    const uint32_t rr = 4 * nq + (uint32_t)j;  // the words that do not fill a quad
    if (rr < cnt) {
        const uint32_t ir = mt_wrap(base + rr);
        r.rx[0] = a[ir]; r.rx[1] = a[ir + 1];
        r.rm = a[mt_wrap(ir + 397)];
    }
}
CTF_DEV void mt_store_word(uint32_t* a, uint32_t i /* < 624 */, uint32_t v, uint32_t old) {
    a[i] = v;
    if (i < CTF_MT_MIRROR) a[CTF_MT_N + i] = v;
    if (i == 0) a[CTF_MT_SAVE] = old;
}
template <int W>
CTF_DEV void mt_produce_store(const ProdRegs<W>& r, uint32_t* a, uint32_t pos, uint32_t off, uint32_t w, int j) {
    if (STEP_ABLATE & 256) return;
    const uint32_t nq = prod_quads<W>(off, w), cnt = prod_cover<W>(off, w);
    const uint32_t base = mt_wrap(mt_wrap(pos + off));
#pragma unroll
    for (int t = 0; t < PROD_TQ; t++) {
        const uint32_t qd = (uint32_t)(t * W + j);
        if (qd < nq) {
            const uint32_t i0 = mt_wrap(base + 4 * qd);
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = mt_twist(r.x[t][k], r.x[t][k + 1], r.m[t][k]);
            if (i0 + 3 < CTF_MT_N) {
                const mt_u32x4 vv = {v[0], v[1], v[2], v[3]};
                *(mt_u32x4*)(a + i0) = vv;
                if (i0 < CTF_MT_MIRROR) *(mt_u32x4*)(a + CTF_MT_N + i0) = vv;  // may run up to 3 words past the mirror: slack
                if (i0 == 0) a[CTF_MT_SAVE] = r.x[t][0];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) mt_store_word(a, mt_wrap(i0 + k), v[k], r.x[t][k]);
            }
        }
    }
    const uint32_t rr = 4 * nq + (uint32_t)j;
    if (rr < cnt) {
        const uint32_t ir = mt_wrap(base + rr);
        mt_store_word(a, ir, mt_twist(r.rx[0], r.rx[1], r.rm), r.rx[0]);
    }
}
// everything after the first batch (a step that consumed more than one batch holds: rare at W >= 4)
template <int W>
CTF_DEV void mt_produce_rest(ProdRegs<W>& r, uint32_t* a, uint32_t pos, uint32_t w, int j) {
    for (uint32_t off = prod_cover<W>(0u, w); off < w; off += prod_cover<W>(off, w)) {
        // a word 227 or more behind this batch's last one is an INPUT of this batch: make the earlier stores visible first
        if (off + prod_cover<W>(off, w) > 227u) ctf_fence_agent();
        mt_produce_load<W>(r, a, pos, off, w, j);
        mt_produce_store<W>(r, a, pos, off, w, j);
    }
}

// Counter mode (cfg.rng_mode == CTF_RNG_COUNTER): the streams are a pure function of (stream seed, word index) —
// word n = Philox4x32-10(key = seed, counter = (n / 4, stream, 0x43544631))[n % 4] — and the ring is only a window of it:
// a consumed word n is replaced by word n + 624.  Same consumption code, no twist, no partner loads.
// (rounds rolled: inside the step kernel the ten round keys must not be hoisted into twenty registers)
CTF_DEV void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
#pragma unroll 1
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define CTF_CTR_TAG 0x43544631u
CTF_DEV void ctr_block(unsigned long long seed, unsigned long long blk, uint32_t stream, uint32_t* out) {
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)blk, (uint32_t)(blk >> 32), stream, CTF_CTR_TAG, out);
}
// Replaces the w words consumed from ring position pos (= word index n0 of the stream) by words n0 + 624 ... : the lanes of
// the group take the 4-word Philox blocks that overlap [n0 + 624, n0 + 624 + w) in turn.
template <int W>
CTF_DEV void ctr_produce(uint32_t* a, uint32_t pos, unsigned long long n0, uint32_t w, unsigned long long seed, uint32_t stream, int j) {
    if (STEP_ABLATE & 256) return;
    const unsigned long long first = n0 + CTF_MT_N, last = first + w;  // absolute indices of the new words
    for (unsigned long long blk = (first >> 2) + (unsigned long long)j; (blk << 2) < last; blk += W) {
        uint32_t o[4];
        ctr_block(seed, blk, stream, o);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long n = (blk << 2) + (unsigned long long)k;
            if (n >= first && n < last) mt_store_word(a, mt_wrap(pos + (uint32_t)(n - first)), o[k], 0u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the step
// ------------------------------------------------------------------------------------------------
template <int W>
struct StepCtx {
    uint8_t* sg;   // grid  (LDS)
    uint8_t* sr;   // record (LDS)
    uint8_t* sm;   // metric deltas of this step (LDS) or nullptr
    int j;         // sub-lane within the env's group
    int gshift;    // first lane of the group
    bool lead;     // j == 0: performs the state writes
    CTF_MEMBER uint32_t ballot(bool p) const { return (uint32_t)(ctf_ballot(p) >> gshift) & ((1u << W) - 1u); }
};

template <int W>
CTF_DEV double ld_hp(const StepCtx<W>& s, int a) {
    const uint32_t* q = (const uint32_t*)(s.sr + 8 * a);
    return ctf_hilo2double(q[1], q[0]);
}
template <int W>
CTF_DEV void st_hp(const StepCtx<W>& s, int a, double v) {
    uint32_t* q = (uint32_t*)(s.sr + 8 * a);
    const uint64_t b = ctf_double_bits(v);
    q[0] = (uint32_t)b;
    q[1] = (uint32_t)(b >> 32);
}
template <bool METRICS, int W>
CTF_DEV void metric_add(const DevCfg& cfg, const StepCtx<W>& s, int m, int a, int v) {
    if (METRICS && s.lead) s.sm[m * cfg.N + a] += (uint8_t)v;  // per-step deltas stay far below 256
}
// a counter that agent a's turn touches exactly once per step: a plain store instead of an LDS read-modify-write (the deltas
// start the step at zero), so that consecutive updates do not wait for each other's reads
template <bool METRICS, int W>
CTF_DEV void metric_set(const DevCfg& cfg, const StepCtx<W>& s, int m, int a, int v) {
    if (METRICS && s.lead) s.sm[m * cfg.N + a] = (uint8_t)v;
}

// respawn, gridworld_ctf.py:761-794 (all lanes of the group compute; sub-lane 0 writes)
template <int W>
CTF_DEV void respawn(const DevCfg& cfg, const StepCtx<W>& s, NpStream& np_, int o, uint32_t& flagm, uint32_t& status) {
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
    const int sel = ctf_ffs(bits) - 1;
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

// GridworldCtf.step for ONE env (state in LDS), executed by the W lanes of its group.  On return np_.X / py.cur are the
// words the step consumed and `npp` holds the loads of the np stream's first production batch (issued behind the last turn,
// so that the second shuffle covers their latency).
template <bool METRICS, int W>
CTF_DEV void env_step(const DevCfg& cfg, const DevPtrs& p, const StepCtx<W>& s, const int8_t* act, PyStream& py, const PyRegs<W>& pyr,
                      NpStream& np_, ProdRegs<W>& npp, uint32_t& status, int e, float* __restrict__ rw32, double* __restrict__ rw64,
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
            ctf_fence_agent();
            for (int i = j; i < N; i += W) ctf_atomic_add_u32(base + i * cfg.GS + cfg.start_pos[i][0] * G + cfg.start_pos[i][1], 1u);
        }
        for (int st = (vis_flags >> CTF_F_FOLDED_SHIFT) + 1; st <= step - 1; st++)
            for (int i = j; i < N; i += W)
                ctf_atomic_add_u32(base + i * cfg.GS + p.vislog[((size_t)(st & (CTF_VIS_LOG - 1)) * cfg.n_envs + e) * N + i], 1u);
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

    // dice_roll (:734-742): the first of the step's two shuffles
    if (!(STEP_ABLATE & 4)) {
        py_shuffle<W>(py, j, N, perm);
        py_topup<W>(py, pyr, j);
    }

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

        // ---- tagging_logic (:796-837): one opponent per lane; hits are applied in opponent order
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
                // opponent q0 + q draws its rand() from words X + 2 q, X + 2 q + 1 = the slot (X >> 1) + q at parity X & 1
                const uint32_t sl = np_.X >> 1, par = np_.X & 1u;
                uint32_t hits;
                if (sl + (uint32_t)cnt <= np_.slots) {
                    // lane j owns the slots == j (mod W): it evaluates the opponent whose slot that is
                    const int qj = (int)(((uint32_t)j - sl) & (uint32_t)(W - 1));
                    bool is_hit = false;
                    if (qj < cnt) {
                        const uint32_t c = (sl + (uint32_t)qj) >> Log2<W>::v;
                        const int o = cfg_opp(cfg, team, q0 + qj);
                        is_hit = ((np_.mine >> (2 * c + par)) & 1u) && cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1;
                    }
                    const uint32_t by_lane = s.ballot(is_hit);
                    const uint32_t rot = sl & (uint32_t)(W - 1);  // opponent q <-> lane (sl + q) % W
                    hits = ((by_lane >> rot) | (by_lane << (W - rot))) & ((1u << W) - 1u);
                } else {  // beyond the covered window: the words come straight from memory, opponent q0 + j on lane j
                    bool is_hit = false;
                    if (j < cnt) {
                        const int o = cfg_opp(cfg, team, q0 + j);
                        const uint32_t wa = np_word_slow(np_, np_.X + 2u * (uint32_t)j) >> 5, wb = np_word_slow(np_, np_.X + 2u * (uint32_t)j + 1u) >> 6;
                        is_hit = np_lt53(np_, wa, wb) && cheb(pr, pc, ps[2 * o], ps[2 * o + 1]) <= 1;
                    }
                    hits = s.ballot(is_hit);
                }
                if (hits == 0) {  // nobody tagged: all cnt doubles consumed
                    np_.X += 2u * (uint32_t)cnt;
                    q0 += cnt;
                    continue;
                }
                const int first = ctf_ffs(hits) - 1;
                np_.X += 2u * (uint32_t)(first + 1);  // doubles up to and including the tagged opponent's
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
                adj_own += ctf_popc(s.ballot(near_own));
                adj_opp += ctf_popc(s.ballot(near_opp));
                if (q - j + W >= (n_own > n_opp ? n_own : n_opp)) break;  // group-uniform exit
            }
            metric_set<METRICS>(cfg, s, CTF_M_STEPS_ADJ_TEAMMATE, a, adj_own);
            metric_set<METRICS>(cfg, s, CTF_M_STEPS_ADJ_OPPONENT, a, adj_opp);
        }
    }

    // the np stream is done for this step: its first production batch's loads go out before the second shuffle
    if (pin(cfg.rng_mode) == CTF_RNG_MT19937) mt_produce_load<W>(npp, np_.a, np_.pos, 0u, np_.X, j);

    // heal_agents' shuffle (:844)
    if (!(STEP_ABLATE & 4)) py_shuffle<W>(py, j, N, perm);

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

// ------------------------------------------------------------------------------------------------
// one group's step between the staging and the write-back of its wave
// ------------------------------------------------------------------------------------------------
// LDS slot of one env (bytes): [grid GS][rec RS][actions 16][py ring 4 * PY_RING][metric deltas u8 13 * N (METRICS)]
// The slot stride in dwords is odd so that different groups' same-offset accesses fall in distinct banks.
CTF_HD int step_slot_bytes(int GS, int RS, int N, bool metrics) {
    int b = GS + RS + 16 + 4 * PY_RING + (metrics ? ((CTF_N_METRICS * N + 3) & ~3) : 0);
    if (((b / 4) & 1) == 0) b += 4;
    return b;
}

template <int W>
struct GroupRng {  // what a group keeps from the prologue to the end of the step
    NpRegs<W> npr;
    PyRegs<W> pyr;
    NpStream np_;
    PyStream py;
    ProdRegs<W> npp, pyp;
};

// the random words of the step: issued as early as the ring positions are known
template <int W>
CTF_DEV void group_issue_loads(GroupRng<W>& R, const DevCfg& cfg, const DevPtrs& p, int e, int j, uint32_t rp_py, uint32_t rp_np) {
    np_issue_loads<W>(R.npr, p.mt_np + (size_t)e * CTF_MT_STRIDE, rp_np, j, np_chunks(cfg, W));
    py_issue_loads<W>(R.pyr, p.mt_py + (size_t)e * CTF_MT_STRIDE, rp_py, j);
}

template <bool METRICS, int W>
CTF_DEV void group_step(GroupRng<W>& R, const DevCfg& cfg, const DevPtrs& p, uint8_t* slot, int e, int j, int gshift, uint32_t rp_py,
                        uint32_t rp_np, uint32_t flags, float* __restrict__ rw32, double* __restrict__ rw64, uint8_t* __restrict__ done_out) {
    const int N = cfg.N;
    StepCtx<W> s;
    s.sg = slot;
    s.sr = slot + cfg.GS;
    uint32_t* ring = (uint32_t*)(s.sr + cfg.RS + 16);
    s.sm = METRICS ? (uint8_t*)(ring + PY_RING) : nullptr;
    s.j = j;
    s.gshift = gshift;
    s.lead = (j == 0);
    const int8_t* act = (const int8_t*)(s.sr + cfg.RS);
    int32_t* misc = (int32_t*)(s.sr + cfg.off_misc);

    np_setup<W>(R.np_, R.npr, cfg, p.mt_np + (size_t)e * CTF_MT_STRIDE, rp_np, np_chunks(cfg, W));
    py_setup<W>(R.py, R.pyr, p.mt_py + (size_t)e * CTF_MT_STRIDE, ring, rp_py, j);

    if ((flags & CTF_STEP_AUTO_RESET) && (misc[3] & CTF_F_DONE)) {
        // reset() of this env inside the step launch (not in the reference: opt-in flag); the group's lanes share the copies
        const uint32_t* src = (const uint32_t*)p.init_grid;
        for (int w = j; w < cfg.GS / 4; w += W) ((uint32_t*)s.sg)[w] = src[w];
        if (s.lead) reset_record(cfg, s.sr);
        if (METRICS) {
            int32_t* m = p.metrics + (size_t)e * CTF_N_METRICS * N;
            for (int w = j; w < CTF_N_METRICS * N; w += W) m[w] = 0;
            // (visitation: reset_record flagged the base maps as zero and emptied the log)
        }
    }

    uint32_t status = 0;
    env_step<METRICS, W>(cfg, p, s, act, R.py, R.pyr, R.np_, R.npp, status, e, rw32, rw64, done_out);
    if (s.lead && status) ctf_atomic_or_u32(p.status, status);
    // the loads of the `random` stream's production: their latency is covered by the wave's state write-back
    if (pin(cfg.rng_mode) == CTF_RNG_MT19937) mt_produce_load<W>(R.pyp, p.mt_py + (size_t)e * CTF_MT_STRIDE, R.py.pos, 0u, R.py.cur, j);
}

// Replaces the words the step consumed and stores the new ring positions (the end of the env's step).
template <int W>
CTF_DEV void group_finish(GroupRng<W>& R, const DevCfg& cfg, const DevPtrs& p, int e, int j) {
    uint32_t* a_py = p.mt_py + (size_t)e * CTF_MT_STRIDE;
    uint32_t* a_np = p.mt_np + (size_t)e * CTF_MT_STRIDE;
    const PyStream& py = R.py;
    const NpStream& np_ = R.np_;
    if (pin(cfg.rng_mode) == CTF_RNG_MT19937) {
        mt_produce_store<W>(R.npp, a_np, np_.pos, 0u, np_.X, j);
        mt_produce_store<W>(R.pyp, a_py, py.pos, 0u, py.cur, j);
        if (np_.X > prod_cover<W>(0u, np_.X)) mt_produce_rest<W>(R.npp, a_np, np_.pos, np_.X, j);
        if (py.cur > prod_cover<W>(0u, py.cur)) mt_produce_rest<W>(R.pyp, a_py, py.pos, py.cur, j);
    } else {
        unsigned long long* ctr = p.rngctr + 4 * (size_t)e;
        const unsigned long long n_py = ctr[0], n_np = ctr[1];
        ctr_produce<W>(a_py, py.pos, n_py, py.cur, ctr[2], 0u, j);
        ctr_produce<W>(a_np, np_.pos, n_np, np_.X, ctr[3], 1u, j);
        if (j == 0) { ctr[0] = n_py + py.cur; ctr[1] = n_np + np_.X; }
    }
    if (j == 0) {
        uint32_t a = py.pos + py.cur, b = np_.pos + np_.X;
        while (a >= CTF_MT_N) a -= CTF_MT_N;
        while (b >= CTF_MT_N) b -= CTF_MT_N;
        p.rngpos[2 * e] = a;
        p.rngpos[2 * e + 1] = b;
    }
}
