// ctf_mt.h — MT19937 in RUN-AHEAD form: the per-env state array always holds the NEXT 624 raw (untempered) outputs.
//
// The reference draws from two process-global MT19937 generators (CPython `random`, NumPy legacy `np.random`;
// reference gridworld_ctf.py:740,771,815).  The standard representation (624 words of block B + a position p) needs
// a[i], a[i+1], a[i+397] to be combined ("twisted") before word i of block B+1 can be output, i.e. a dependent memory
// round trip in the middle of whatever consumes the numbers.  Here the array is kept one whole block AHEAD of the
// consumer instead:
//
//     a[i], i in [pos, 624)  = raw word i of block B        (not consumed yet)
//     a[i], i in [0, pos)    = raw word i of block B + 1    (not consumed yet either)
//
// so the next outputs are temper(a[pos]), temper(a[pos+1]), ... (wrapping to a[0]) with no arithmetic in front of them,
// and a consumer that took w words replaces exactly those w words by their successors one block later
// (a[i] <- twist(a[i], a[i+1], a[i+397]), in stream order: the standard in-place regeneration, merely delayed by 624
// words).  That replacement has no consumer inside the step, so it runs at the tail of the step kernel.
//
// Array layout per env and stream (u32 words, CTF_MT_STRIDE of them):
//     [0, 624)                      the ring above
//     [624, 624 + CTF_MT_MIRROR)    copy of words [0, CTF_MT_MIRROR): every span a kernel reads is contiguous
//     [CTF_MT_SAVE]                 the raw word 0 of the block BEFORE the one a[0] belongs to (see below)
//
// Conversion to / from the standard form (random.getstate() / np.random.get_state()):
//   std -> run-ahead: the first p iterations of the standard in-place regeneration (all 624 when p == 624).
//   run-ahead -> std: those iterations are undone.  The regeneration step is invertible: word i of the new block fixes the
//   top bit of old word i and the low 31 bits of old word i + 1; only the low 31 bits of old word 0 never enter any later
//   output — they are kept in a[CTF_MT_SAVE] so that the round trip is exact to the last bit.  pos == 0 is reported as the
//   equivalent standard state (previous block, 624), which is what CPython / NumPy hold after a block's last word.
#pragma once
#include <stdint.h>

#include "../../include/ctf_env.h"

#if defined(__HIPCC__) || defined(__CUDACC__)
#define CTF_HD __host__ __device__ __forceinline__
#else
#define CTF_HD static inline
#endif

#define CTF_MT_MIRROR 200  // >= the longest span any kernel reads past a start index < 624 (193 words: W = 8, 12 chunks)
#define CTF_MT_SAVE 828    // word index of the saved "old word 0"
#define CTF_MT_STRIDE 832  // words per env and stream (a multiple of 32: arrays start on 128-byte lines)

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
// the inverse: from a new word and its partner m -> (top bit of old word i) | (low 31 bits of old word i + 1)
CTF_HD uint32_t mt_untwist(uint32_t v, uint32_t m) {
    uint32_t y = v ^ m;
    const uint32_t odd = y >> 31;  // (y >> 1) has a clear top bit, the magic constant a set one
    if (odd) y ^= 0x9908b0dfu;
    return (y << 1) | odd;
}

// Standard form (a[0..624) = block B, position p in 0..624) -> run-ahead form, in place (sequential; a may be LDS or host memory).
// Returns the run-ahead position (0..623) and leaves the word for a[CTF_MT_SAVE] in *save0.
CTF_HD uint32_t mt_std_to_runahead(uint32_t* a, uint32_t p, uint32_t* save0) {
    *save0 = a[0];
    const uint32_t n = p > CTF_MT_N ? CTF_MT_N : p;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t i1 = (i + 1 == CTF_MT_N) ? 0u : i + 1;
        const uint32_t im = (i + 397 >= CTF_MT_N) ? i + 397 - CTF_MT_N : i + 397;
        a[i] = mt_twist(a[i], a[i1], a[im]);
    }
    return n == CTF_MT_N ? 0u : n;
}
// Run-ahead form -> standard form, in place (sequential).  Returns the standard position (1..624).
CTF_HD uint32_t mt_runahead_to_std(uint32_t* a, uint32_t pos, uint32_t save0) {
    const uint32_t n = pos == 0 ? (uint32_t)CTF_MT_N : pos;
    for (uint32_t i = n; i-- > 0;) {
        // partner: old word i + 397 (untouched if >= n, else already recovered: its two halves came from steps i + 397 and
        // i + 396, both behind us) or new word i - 227 (not reached yet)
        const uint32_t m = (i < CTF_MT_N - 397) ? a[i + 397] : a[i - (CTF_MT_N - 397)];
        const uint32_t y = mt_untwist(a[i], m);
        if (i + 1 < n) a[i + 1] = (a[i + 1] & 0x80000000u) | (y & 0x7fffffffu);
        a[i] = y & 0x80000000u;  // the low bits follow from step i - 1
    }
    a[0] = (a[0] & 0x80000000u) | (save0 & 0x7fffffffu);
    return n;
}
