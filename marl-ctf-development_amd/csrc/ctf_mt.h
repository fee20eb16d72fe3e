// ctf_mt.h — the two per-env random streams as the step kernel sees them: RINGS of raw words + DIGESTS of what a step needs.
//
// The reference draws from two process-global MT19937 generators (CPython `random`: the shuffles, reference
// gridworld_ctf.py:740; NumPy legacy `np.random`: rand() per opponent :815 and randint() :771).  An MT19937 block of 624
// words is a function of the block before it (a[i] <- twist(a[i], a[i+1], a[i+397])), so somebody has to do that
// regeneration — three loads, a store and a dozen operations per word — and with it inside the step kernel every step
// paid a dependent memory round trip per 16 words (round 2) or ~0.5 KB of scattered loads and stores per env (the first
// version of round 3).  Now the step kernel regenerates nothing:
//
//   per env and stream, TWO rings of 624 raw words: ring `cur` holds block B (the consumer stands at `pos` in it, exactly
//   the standard (state, position) form of random.getstate() / np.random.get_state()), the other ring holds block B + 1;
//   per ring, DIGESTS of its words — everything the step kernel ever looks at:
//     np.random:  hit bit per position p (rand() drawn from words p, p + 1 is < TAG_PROBABILITY) and the low 4 bits of the
//                 tempered word (all a randint() over a <= 9-cell spawn window needs);
//     random:     the top byte of the tempered word (_randbelow(n) for n <= 16 takes its top <= 5 bits);
//   the digest arrays of a ring end with a MIRROR of the head of the other ring's digests, so that a window that runs over the
//   end of the block is still one contiguous load.
//
// A step loads ~128 digest bytes per env with three load instructions and stores two positions.  When the consumer leaves a
// ring it marks it stale (rngready), and one wave of a later launch — a TAIL BLOCK of k_step, or k_rng_refill after seeding /
// import — regenerates it from the ring that is current now: whole blocks, fully coalesced, 8 bytes of traffic per word, and
// writes its digests.  ring_next_block / ring_digest / ring_link below are that regeneration, written once for 64 lanes (staging
// in LDS), for one lane (the step kernel's safety net, should a ring ever be needed before a tail block got to it) and for the
// host (tests/hostsim).
//
// Counter mode (CTF_RNG_COUNTER): the same rings and digests, but block k of a stream is words [624 k, 624 k + 624) of
// Philox4x32-10(key = the stream's seed, counter = (word / 4, stream, "CTF1")) and the words are used as they are (no tempering).
#pragma once
#include <stdint.h>

#include "../../include/ctf_env.h"

#if defined(__HIPCC__)
#define CTF_HD __host__ __device__ __forceinline__
#else
#define CTF_HD static inline
#endif

// digest array sizes per ring, in dwords (ring + mirror of the other ring's head)
#define CTF_HB_DW 26    // np hit bits: 624 + 208 bits (a 128-bit window from dword pos >> 5 <= 19 ends at dword 22)
#define CTF_NB_DW 92    // np nibbles: 624 + 112 nibbles (a 12-dword window from dword pos >> 3 <= 78 ends at dword 89)
#define CTF_P8_DW 176   // py top bytes: 624 + 80 bytes (a 16-dword window from dword pos >> 2 <= 156 ends at dword 171)
#define CTF_HB_MIRROR 208
#define CTF_NB_MIRROR 112
#define CTF_P8_MIRROR 80
// rngpos word of a stream: position 0..624 | current ring << 16.  Whether the OTHER ring is in place (regenerated and linked) is
// a byte of its own, rngready: 1 = in place; 2 + r = ring r is stale (its consumer has moved on to ring 1 - r) and waits to be
// regenerated; 0 = nothing valid yet (before the first k_rng_refill(init)).  The consumer's launch writes the position, the
// regenerating wave — possibly of the same launch — the flag, and the flag alone says which ring is to be rebuilt from which.
#define CTF_RP_POS(x) ((x) & 0xFFFFu)
#define CTF_RP_CUR(x) (((x) >> 16) & 1u)
#define CTF_RP_MAKE(pos, cur) ((uint32_t)(pos) | ((uint32_t)(cur) << 16))

CTF_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
// word i of the next block from words i, i + 1 and i + 397 (indices modulo 624) of the state as the standard in-place loop sees it
CTF_HD uint32_t mt_twist(uint32_t x0, uint32_t x1, uint32_t m) {
    const uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

CTF_HD void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
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
// the four words 4 blk .. 4 blk + 3 of counter stream `stream` (0 random, 1 np.random) of seed `seed`
CTF_HD void ctr_block(unsigned long long seed, unsigned long long blk, uint32_t stream, uint32_t* out) {
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)blk, (uint32_t)(blk >> 32), stream, CTF_CTR_TAG, out);
}

// np.random.rand() < p on the 53-bit integer of the draw: (a >> 5) * 2^26 + (b >> 6) < thr, thr = ceil(p * 2^53) split at bit 26
CTF_HD bool mt_lt53(uint32_t hi27, uint32_t lo26, uint32_t th, uint32_t tl) { return hi27 < th || (hi27 == th && lo26 < tl); }

struct RingPtrs {       // one env, one stream
    uint32_t* raw;      // [2][624]
    uint32_t* hit;      // [2][CTF_HB_DW]   (np stream only)
    uint32_t* nib;      // [2][CTF_NB_DW]   (np stream only)
    uint32_t* top;      // [2][CTF_P8_DW]   (py stream only)
};
struct RingParams {
    int stream;                 // 0 random (py), 1 np.random
    int counter_mode;
    uint32_t th, tl;            // np: the tag threshold
    unsigned long long seed;    // counter mode
    unsigned long long nbase;   // counter mode: stream index of word 0 of the SOURCE ring
};
CTF_HD uint32_t ring_out(const RingParams& q, uint32_t w) { return q.counter_mode ? w : mt_temper(w); }

// Digests of the 624 words `w` (ring r of the env): everything but the hit bit of position 623, which needs the next block.
// Lane `lane` of LANES takes every LANES-th digest dword.
template <int LANES>
CTF_HD void ring_digest(int lane, const uint32_t* w, const RingPtrs& p, int r, const RingParams& q) {
    if (q.stream == 1) {
        uint32_t* hit = p.hit + r * CTF_HB_DW;
        for (int d = lane; d < (CTF_MT_N + 31) / 32; d += LANES) {
            uint32_t bits = 0;
            uint32_t t0 = ring_out(q, w[32 * d]);
            for (int k = 0; k < 32 && 32 * d + k < CTF_MT_N - 1; k++) {
                const uint32_t t1 = ring_out(q, w[32 * d + k + 1]);
                bits |= (mt_lt53(t0 >> 5, t1 >> 6, q.th, q.tl) ? 1u : 0u) << k;
                t0 = t1;
            }
            hit[d] = bits;  // (dword 19: position 623 and the mirror above it are written by ring_link, once the next block exists)
        }
        uint32_t* nib = p.nib + r * CTF_NB_DW;
        for (int d = lane; d < CTF_MT_N / 8; d += LANES) {
            uint32_t v = 0;
            for (int k = 0; k < 8; k++) v |= (ring_out(q, w[8 * d + k]) & 15u) << (4 * k);
            nib[d] = v;
        }
    } else {
        uint32_t* top = p.top + r * CTF_P8_DW;
        for (int d = lane; d < CTF_MT_N / 4; d += LANES) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) v |= (ring_out(q, w[4 * d + k]) >> 24) << (8 * k);
            top[d] = v;
        }
    }
}
// Links ring `c` to its successor ring `o`: the mirror behind ring c's digests = the digests of the head of ring o, and the hit
// bit of ring c's last position (its rand() takes ring o's first word).  wc / wo: the two rings' words.  Everything is worked out
// from the words (nothing another lane has just stored is read back).
template <int LANES>
CTF_HD void ring_link(int lane, const uint32_t* wc, const uint32_t* wo, const RingPtrs& p, int c, const RingParams& q) {
    if (q.stream == 1) {
        uint32_t* hc = p.hit + c * CTF_HB_DW;
        // dwords 19 .. 25 of ring c's hit array: positions 608 .. 831, i.e. ring c's own last 16 and ring o's first 208
        for (int d = lane; d < CTF_HB_DW - (CTF_MT_N >> 5); d += LANES) {
            uint32_t bits = 0;
            for (int k = 0; k < 32; k++) {
                const int pc = 32 * ((CTF_MT_N >> 5) + d) + k;  // position counted from ring c's start
                const uint32_t w0 = pc < CTF_MT_N ? wc[pc] : wo[pc - CTF_MT_N];
                const uint32_t w1 = pc + 1 < CTF_MT_N ? wc[pc + 1] : wo[pc + 1 - CTF_MT_N];
                bits |= (mt_lt53(ring_out(q, w0) >> 5, ring_out(q, w1) >> 6, q.th, q.tl) ? 1u : 0u) << k;
            }
            hc[(CTF_MT_N >> 5) + d] = bits;
        }
        uint32_t* nc = p.nib + c * CTF_NB_DW;
        for (int d = lane; d < CTF_NB_MIRROR / 8; d += LANES) {
            uint32_t v = 0;
            for (int k = 0; k < 8; k++) v |= (ring_out(q, wo[8 * d + k]) & 15u) << (4 * k);
            nc[CTF_MT_N / 8 + d] = v;
        }
    } else {
        uint32_t* tc = p.top + c * CTF_P8_DW;
        for (int d = lane; d < CTF_P8_MIRROR / 4; d += LANES) {
            uint32_t v = 0;
            for (int k = 0; k < 4; k++) v |= (ring_out(q, wo[4 * d + k]) >> 24) << (8 * k);
            tc[CTF_MT_N / 4 + d] = v;
        }
    }
}
// The block after `src` (624 words) into `dst` (624 words; must not alias src).  Three dependent chunks (word i >= 227 takes the
// NEW word i - 227): `sync` separates them.
template <int LANES, typename Sync>
CTF_HD void ring_next_block(int lane, const uint32_t* src, uint32_t* dst, const RingParams& q, Sync sync) {
    if (q.counter_mode) {
        for (int b = lane; b < CTF_MT_N / 4; b += LANES) {
            uint32_t o[4];
            ctr_block(q.seed, (q.nbase + CTF_MT_N) / 4 + (unsigned long long)b, (uint32_t)q.stream, o);  // nbase is a multiple of 624 = 4 * 156
            dst[4 * b] = o[0]; dst[4 * b + 1] = o[1]; dst[4 * b + 2] = o[2]; dst[4 * b + 3] = o[3];
        }
        sync();
        return;
    }
    for (int i = lane; i < 227; i += LANES) dst[i] = mt_twist(src[i], src[i + 1], src[i + 397]);
    sync();
    for (int i = 227 + lane; i < 454; i += LANES) dst[i] = mt_twist(src[i], src[i + 1], dst[i - 227]);
    sync();
    for (int i = 454 + lane; i < CTF_MT_N - 1; i += LANES) dst[i] = mt_twist(src[i], src[i + 1], dst[i - 227]);
    if (lane == 0) dst[CTF_MT_N - 1] = mt_twist(src[CTF_MT_N - 1], dst[0], dst[396]);
    sync();
}
