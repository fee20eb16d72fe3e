// ctf_device.h — device-side description of one GridworldCtf configuration and the HBM layout of the
// per-env state.  Shared by the kernels (ctf_kernels.hip) and the C-ABI host code (ctf_abi.hip).
//
// HBM layout (E = n_envs), everything per-env-contiguous so that one wave moves one env's (or 64
// envs') bytes with full-width coalesced accesses:
//   grid   u8  [E][GS]        GS = G*G rounded up to 16; row-major tile codes (reference self.grid)
//   rec    u8  [E][RS]        one record per env, RS multiple of 16:
//                               f64 hp[N] | i8 pos[N][2] | u8 has_flag[N] | u8 perm[N] | i16 inv[N]
//                               | i32 step | i32 caps[2] | i32 flags (done, visitation log state)
//   mt_py  u32 [E][2][624]    CPython `random` stream: two rings of raw MT19937 words, the current block and the next (ctf_mt.h)
//   mt_np  u32 [E][2][624]    NumPy legacy `np.random` stream, same form
//   py_top u32 [E][2][176]    digest of a `random` ring: the top byte of every tempered word + mirror of the other ring's head
//   np_hit u32 [E][2][26]     digests of an np.random ring: per position, is the rand() that starts there < TAG_PROBABILITY ...
//   np_nib u32 [E][2][92]     ... and the low 4 bits of the tempered word (randint over the spawn window); + mirrors
//   rngpos u32 [E][2]         per stream: position 0..624 in the current ring | current ring << 16
//   rngready u8 [E][2]        per stream: 1 = the other ring is in place; 2 + r = ring r is stale and waits for its regeneration
//   rngage u8 [E][2]          per stream: step launches a stale ring has waited so far (k_step's tail blocks: theirs alone)
//   rngctr u64 [E][6]         counter mode only (cfg.rng_mode == 1): stream index of word 0 of ring 0 / ring 1 (py), of ring 0 / 1 (np), seeds (py, np)
//   metric i32 [E][13][N]     agent-level counters (only when log_metrics)
//   vislog u16 [512][E][N]    visitation LOG: entry (step % 512) = the cell of every agent after that step; the
//                             maps are rebuilt from it on export, so a step writes 2N coalesced bytes per env
//                             instead of N scattered read-modify-writes (only when log_metrics)
//   vis    u32 [E][N][GS]     visitation BASE maps: what has been folded out of the log (an env that is not reset
//                             for 511 steps folds its own log in-kernel) or handed in by ctf_set_state
#pragma once
#include <stdint.h>

#include "../../include/ctf_env.h"
#include "ctf_mt.h"

#define CTF_TILE_NONE 15u  // channel code of a tile that has no observation plane
#ifndef CTF_OBS_TILE
#define CTF_OBS_TILE 8192  // bytes of the flat observation buffer one wave of k_observe_tiles renders (-D: profiling)
#endif
#ifndef CTF_OBS_TILE_WPB
#define CTF_OBS_TILE_WPB 4  // independent one-wave tiles per block (-D: profiling)
#endif

struct FastDiv {  // q = (n * m) >> s, exact for every n the kernels use (verified on the host at create)
    uint32_t m, s;
};

struct DevCfg {
    int32_t n_envs, N, G, GG, C, M;
    int32_t game_steps, flip_axis;
    int32_t home_flag_capture, use_adjusted, drop_flag, log_metrics;
    int32_t n_opp[2];
    int32_t GS, RS;                 // strides of grid / rec in bytes
    int32_t CGG, obs_bytes;         // C*G*G, N*C*G*G
    int32_t off_pos, off_flag, off_perm, off_inv, off_misc;  // record offsets (hp is at 0)
    int32_t default_reverse;        // bit i = (team(i) == 1)
    FastDiv div_cgg, div_gg, div_g, div_m, div_n, div_gq, div_rq, div_mn, div_mw;
    FastDiv div_gg_row;             // / GG over 0 .. N*GG (the compact observation's rows)
    int32_t step_lanes_override;    // 0 = automatic; set from CTF_STEP_W for profiling
    int32_t rng_refill_every;       // 1: rings are regenerated at the tail of the next step launch (default); 0: never (tests: the
                                    // step kernel's safety net does all the work)
    int32_t rng_safe_ahead;         // words: a step that starts this far (or further) before the end of its block needs no other ring
    int32_t rng_spread;             // a stale ring is regenerated within this many launches of being seen: a burst of them (the
                                    // envs' positions move in step) is spread over as many launches by the rings' ages (rngage)
    int32_t n_cus;                  // compute units of the device (launch shapes)
    int32_t obs_store_nt;           // k_observe_tiles stores with the nontemporal hint: when the batch's observations exceed 320 MB (CTF_OBS_NT=0 / 1 forces)
    // np.random.rand() < TAG_PROBABILITY on the 53-bit integer x = (a >> 5) * 2^26 + (b >> 6): x < tag_thr, split at bit 26
    uint32_t tag_th, tag_tl;
    int32_t np_pairs;               // rand() draws of one step without respawns: sum over the agents that deal damage of their opponents
    uint32_t dmg_mask;              // bit i = AGENT_TYPE_DAMAGE[type(i)] > 0
    uint32_t flag_pack, capture_pack, spawn_pack;  // the two teams' cells: row 0 | col 0 << 8 | row 1 << 16 | col 1 << 24
    int32_t rng_mode;               // CTF_RNG_MT19937 / CTF_RNG_COUNTER
    // the tile render (k_observe_tiles): envs are taken in groups of tile_k, the smallest count whose blocks fill a whole
    // number (tile_tpg) of tiles; div_ob_tile divides a tile's byte offset inside its group (a multiple of the tile size) by obs_bytes
    int32_t tile_k, tile_tpg;
    FastDiv div_ob_tile;
    // its launch is 1-D: tile_nb blocks (a multiple of 8) of CTF_OBS_TILE_WPB tiles, tile_bx blocks per env group; div_tile_bx
    // splits a block index into (group, block in group)
    int32_t tile_bx, tile_nb;
    FastDiv div_tile_bx;
    double heal, tag_p, guard_mult, vault_cost, vault_min;
    double r_capture, r_step, r_tag, win_scalar, loss_scalar, punish;
    double type_hp[4], type_damage[4];
    uint64_t chan_lut[2];           // [viewer team]: nibble v = channel of tile v after relabelling, 15 = none
    // lane-divergent lookups are bit-field extracts from these SGPR-resident packs (no memory access):
    uint32_t team_mask;             // bit i = team(i)
    uint32_t type_pack;             // 2 bits per agent
    uint64_t opp_pack[2];           // nibble q of [t] = OPPONENTS[t][q]
    uint64_t self_idx_pack;         // nibble i = index of agent i in its own team's list, 15 = not in it
    int8_t team[CTF_MAX_AGENTS], type[CTF_MAX_AGENTS];
    int8_t opp[2][CTF_MAX_AGENTS];
    int8_t flag_pos[2][2], capture_pos[2][2], spawn_pos[2][2];
    int8_t start_pos[CTF_MAX_AGENTS][2];
    int8_t meta_order[CTF_MAX_AGENTS][CTF_MAX_AGENTS];  // viewer i: teammates (ascending, not i) then opponents; -1 = none
};

struct DevPtrs {
    uint8_t* grid;
    uint8_t* rec;
    uint32_t* mt_py;
    uint32_t* mt_np;
    uint32_t* rngpos;
    uint8_t* rngready;
    uint8_t* rngage;
    unsigned long long* rngctr;  // counter mode: u64 [E][6] = stream index of word 0 of each ring (py 0, py 1, np 0, np 1), stream seeds (py, np)
    uint32_t* py_top;
    uint32_t* np_hit;
    uint32_t* np_nib;
    int32_t* metrics;
    uint32_t* vis;             // base maps u32 [E][N][GS]; valid only when the env's CTF_F_BASE_ZERO flag is clear
    uint16_t* vislog;          // u16 [CTF_VIS_LOG][E][N]
    const uint8_t* init_grid;  // GS bytes
    const uint8_t* meta_lut;   // N * M bytes: which of the env's few distinct metadata values each element of the N x M block shows
    uint32_t* status;
};

// rec.misc[3]: bit 0 done, bit 1 base maps are implicitly zero (+1 at the start cells), bits 2.. = last step whose
// log entry has been folded into the base maps
#define CTF_F_DONE 1
#define CTF_F_BASE_ZERO 2
#define CTF_F_FOLDED_SHIFT 2
#define CTF_VIS_LOG 512
#define CTF_POS_MASK 0xFFFFu
