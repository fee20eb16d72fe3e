// ctf_derive.h — ctf_config (the C ABI's flat description of one GridworldCtf configuration) -> DevCfg (what the kernels
// take as their constant block).  Pure host C++: shared by ctf_abi.hip and by the host simulator of the step logic
// (tests/hostsim/).  The includer provides `static int fail(int code, const char* fmt, ...)`.
#pragma once
#include <cstdlib>
#include <cstring>

#include "ctf_device.h"

static int ceil_log2(uint32_t d) {
    int l = 0;
    while ((1u << l) < d) l++;
    return l;
}
// Granlund-Montgomery round-up reciprocal, exact for every n < n_max (checked exhaustively)
static bool make_fastdiv(uint32_t d, uint32_t n_max, FastDiv* out) {
    int nbits = ceil_log2(n_max + 1);
    if (nbits < 1) nbits = 1;
    const int s = nbits + ceil_log2(d);
    if (s > 62) return false;
    const uint64_t m = ((1ull << s) + d - 1) / d;
    if (m > 0xFFFFFFFFull) return false;
    out->m = (uint32_t)m;
    out->s = (uint32_t)s;
    for (uint32_t n = 0; n < n_max; n++)
        if ((uint32_t)(((uint64_t)n * out->m) >> out->s) != n / d) return false;
    return true;
}

// the same reciprocal, needed (and checked) only for the multiples of `stride` below n_max
static bool make_fastdiv_strided(uint32_t d, uint64_t n_max, uint32_t stride, FastDiv* out) {
    int nbits = 1;
    while ((1ull << nbits) < n_max + 1) nbits++;
    const int s = nbits + ceil_log2(d);
    if (s > 62) return false;
    const uint64_t m = ((1ull << s) + d - 1) / d;
    if (m > 0xFFFFFFFFull) return false;
    out->m = (uint32_t)m;
    out->s = (uint32_t)s;
    for (uint64_t n = 0; n < n_max; n += stride)
        if ((uint32_t)((n * out->m) >> out->s) != (uint32_t)(n / d)) return false;
    return true;
}
static int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

static int round_up(int x, int a) { return (x + a - 1) / a * a; }

// Which of the env's distinct metadata values (mv[] of obs_build_env: 0 step fraction, 1 + t capture ratio for a viewer of
// team t, 4 + j uint8-truncated hp of agent j, 20 + j has_flag[j], 40 = 1.0, 41 = 0.0) element k of viewer i's row shows
// (gridworld_ctf.py:1044-1067) — the host twin of obs_meta_lut in ctf_kernels.hip, for the kernels that cannot afford to
// rebuild the table per wave.
static void host_meta_lut(const DevCfg& d, uint8_t* out) {
    const int N = d.N, M = d.M;
    for (int i = 0; i < N; i++)
        for (int k = 0; k < M; k++) {
            const int team = d.team[i];
            int src = 41;
            if (k == 0) src = 0;
            else if (k == 1) src = 1 + team;
            else if (k < 6) src = (k - 2 == d.type[i]) ? 40 : 41;
            else {
                int who = i;
                if (k >= 8) {
                    const int pidx = (k - 8) >> 1;
                    const int n_own = d.n_opp[1 - team], n_op = d.n_opp[team];
                    int self_idx = 15;
                    for (int q = n_own - 1; q >= 0; q--)
                        if (d.opp[1 - team][q] == i) self_idx = q;
                    const int n_mates = n_own - (self_idx < n_own ? 1 : 0);
                    if (pidx < n_mates) who = d.opp[1 - team][pidx + (pidx >= self_idx ? 1 : 0)];
                    else if (pidx - n_mates < n_op) who = d.opp[team][pidx - n_mates];
                    else who = -1;
                }
                if (who >= 0) src = ((k & 1) ? 20 : 4) + who;
            }
            out[i * M + k] = (uint8_t)src;
        }
}

static int derive(const ctf_config* c, int32_t n_envs, DevCfg* d) {
    memset(d, 0, sizeof(*d));
    if (c->abi_version != CTF_ABI_VERSION) return fail(CTF_E_INVALID, "ctf_config.abi_version %d != %d", c->abi_version, CTF_ABI_VERSION);
    if (n_envs < 1) return fail(CTF_E_INVALID, "n_envs must be >= 1");
    const int N = c->n_agents, G = c->grid_size, C = c->n_channels;
    if (N < 2 || N > CTF_MAX_AGENTS) return fail(CTF_E_INVALID, "n_agents %d outside 2..%d", N, CTF_MAX_AGENTS);
    if (G < 4 || G > CTF_MAX_GRID) return fail(CTF_E_INVALID, "grid_size %d outside 4..%d", G, CTF_MAX_GRID);
    if (C < 2 || C > 15) return fail(CTF_E_INVALID, "n_channels %d outside 2..15", C);
    if (c->game_steps < 1) return fail(CTF_E_INVALID, "game_steps must be >= 1");
    if (c->flip_axis < -1 || c->flip_axis > 2) return fail(CTF_E_INVALID, "flip_axis %d", c->flip_axis);
    if (c->rng_mode != CTF_RNG_MT19937 && c->rng_mode != CTF_RNG_COUNTER) return fail(CTF_E_INVALID, "rng_mode %d", c->rng_mode);
    for (int t = 0; t < 2; t++) {
        if (c->n_opponents[t] < 0 || c->n_opponents[t] > N) return fail(CTF_E_INVALID, "n_opponents[%d]", t);
        for (int k = 0; k < c->n_opponents[t]; k++)
            if (c->opponents[t][k] < 0 || c->opponents[t][k] >= N) return fail(CTF_E_INVALID, "opponents[%d][%d]", t, k);
        const int8_t* pos[3] = {c->flag_pos[t], c->capture_pos[t], c->spawn_pos[t]};
        for (int k = 0; k < 3; k++)
            if (pos[k][0] < 0 || pos[k][0] >= G || pos[k][1] < 0 || pos[k][1] >= G) return fail(CTF_E_INVALID, "team %d position outside the grid", t);
    }
    for (int i = 0; i < N; i++) {
        if (c->agent_team[i] < 0 || c->agent_team[i] > 1) return fail(CTF_E_INVALID, "agent_team[%d]", i);
        if (c->agent_type[i] < 0 || c->agent_type[i] > 3) return fail(CTF_E_INVALID, "agent_type[%d]", i);
        if (c->start_pos[i][0] < 0 || c->start_pos[i][0] >= G || c->start_pos[i][1] < 0 || c->start_pos[i][1] >= G)
            return fail(CTF_E_INVALID, "start_pos[%d] outside the grid", i);
        if (!(c->type_hp[c->agent_type[i]] > 0)) return fail(CTF_E_INVALID, "type_hp of agent %d must be > 0", i);
    }
    for (int k = 1; k < C; k++)
        if (c->tile_of_channel[k] < 1 || c->tile_of_channel[k] > 13) return fail(CTF_E_INVALID, "tile_of_channel[%d]", k);
    for (int k = 0; k < G * G; k++)
        if (c->init_grid[k] > 13) return fail(CTF_E_INVALID, "init_grid[%d] = %d", k, c->init_grid[k]);

    d->n_cus = 256;  // (ctf_create overwrites it with the device's count)
    d->n_envs = n_envs; d->N = N; d->G = G; d->GG = G * G; d->C = C; d->M = 2 * N + 6;
    d->game_steps = c->game_steps; d->flip_axis = c->flip_axis;
    d->home_flag_capture = c->home_flag_capture; d->use_adjusted = c->use_adjusted_rewards;
    d->drop_flag = c->drop_flag_when_no_hp; d->log_metrics = c->log_metrics ? 1 : 0;
    d->rng_mode = c->rng_mode;
    {   // np.random.rand() < p  <=>  the 53-bit integer of the draw < ceil(p * 2^53) (x / 2^53 and p * 2^53 are exact)
        const double y = c->tag_probability * 9007199254740992.0;
        uint64_t thr = 0;
        if (y > 0) thr = y >= 9007199254740992.0 ? (1ull << 53) : (uint64_t)y + (((double)(uint64_t)y < y) ? 1u : 0u);
        d->tag_th = (uint32_t)(thr >> 26);
        d->tag_tl = (uint32_t)(thr & ((1u << 26) - 1u));
    }
    for (int i = 0; i < N; i++) {
        if (c->type_damage[c->agent_type[i]] > 0) {
            d->dmg_mask |= 1u << i;
            d->np_pairs += c->n_opponents[c->agent_team[i]];
        }
    }
    d->n_opp[0] = c->n_opponents[0]; d->n_opp[1] = c->n_opponents[1];
    d->GS = round_up(G * G, 16);
    d->off_pos = 8 * N; d->off_flag = 10 * N; d->off_perm = 11 * N; d->off_inv = 12 * N;
    d->off_misc = round_up(14 * N, 4);
    d->RS = round_up(d->off_misc + 16, 16);
    d->CGG = C * G * G; d->obs_bytes = N * d->CGG;
    if (!make_fastdiv((uint32_t)d->CGG, (uint32_t)d->obs_bytes + 16, &d->div_cgg) ||
        !make_fastdiv((uint32_t)d->GG, (uint32_t)d->CGG + 16, &d->div_gg) ||
        !make_fastdiv((uint32_t)d->GG, (uint32_t)(N * d->GG) + 16, &d->div_gg_row) ||
        !make_fastdiv((uint32_t)G, (uint32_t)d->GS + 4, &d->div_g) ||
        !make_fastdiv((uint32_t)d->M, (uint32_t)(N * d->M) + 64, &d->div_m) ||
        !make_fastdiv((uint32_t)N, (uint32_t)(256 * N) + 64, &d->div_n) ||
        !make_fastdiv((uint32_t)(d->GS / 16), (uint32_t)(64 * d->GS / 16) + 64, &d->div_gq) ||
        !make_fastdiv((uint32_t)(d->RS / 16), (uint32_t)(64 * d->RS / 16) + 64, &d->div_rq) ||
        !make_fastdiv((uint32_t)(CTF_N_METRICS * N), (uint32_t)(64 * CTF_N_METRICS * N) + 64, &d->div_mn) ||
        !make_fastdiv((uint32_t)((CTF_N_METRICS * N + 3) / 4), (uint32_t)(64 * ((CTF_N_METRICS * N + 3) / 4)) + 64, &d->div_mw))
        return fail(CTF_E_INVALID, "internal: reciprocal division not exact for these dimensions");
    d->tile_k = d->tile_tpg = 0;  // 0: no tile render for this configuration
    if (d->obs_bytes % 16 == 0 && d->obs_bytes >= CTF_OBS_TILE) {
        d->tile_k = CTF_OBS_TILE / gcd_int(d->obs_bytes, CTF_OBS_TILE);
        d->tile_tpg = (int)((int64_t)d->tile_k * d->obs_bytes / CTF_OBS_TILE);
        d->tile_bx = (d->tile_tpg + CTF_OBS_TILE_WPB - 1) / CTF_OBS_TILE_WPB;
        const int64_t nb = ((int64_t)d->tile_bx * ((n_envs + d->tile_k - 1) / d->tile_k) + 7) / 8 * 8;
        d->tile_nb = (int32_t)nb;
        if (nb > 0x3FFFFFFF || !make_fastdiv((uint32_t)d->tile_bx, (uint32_t)nb + 8, &d->div_tile_bx) ||
            !make_fastdiv_strided((uint32_t)d->obs_bytes, (uint64_t)d->tile_k * d->obs_bytes + CTF_OBS_TILE, CTF_OBS_TILE, &d->div_ob_tile))
            d->tile_k = d->tile_tpg = d->tile_bx = d->tile_nb = 0;
    }
    // the rings a step's consumers leave are regenerated at the tail of the next step launch; tests set 0 = never, which leaves all
    // regeneration to the step kernel's safety net
    d->rng_refill_every = 1;
    if (const char* ov = getenv("CTF_RNG_REFILL_EVERY")) d->rng_refill_every = atoi(ov) != 0;
    {   // A consumer that has just moved to a new block stands in its first `worst` words and needs the block after it once it is
        // closer than rng_safe_ahead to the end: (624 - ahead - worst) / worst steps later at the earliest.  A stale ring is
        // regenerated within rng_spread launches, the first of them the launch after (or of) the move.
        const int worst_np = 2 * d->np_pairs + 32, worst_py = 8 * N + 32;  // words per step, generously
        const int worst = worst_np > worst_py ? worst_np : worst_py;
        d->rng_safe_ahead = 192 > worst + 64 ? 192 : worst + 64;  // >= the digest windows' reach (128 + 31) and any step's draws
        int slack = (CTF_MT_N - d->rng_safe_ahead - worst) / worst;   // whole steps between the move and the first need
        d->rng_spread = slack < 1 ? 1 : (slack > 4 ? 4 : slack);
        if (const char* ov = getenv("CTF_RNG_SPREAD")) {
            const int v = atoi(ov);
            if (v >= 1 && v <= d->rng_spread) d->rng_spread = v;
        }
    }
    // The tile render's stores carry the nontemporal hint when the batch's observations are larger than the caches can absorb
    // (the memory-side cache holds 256 MB): 1.65 GB per launch then pass by instead of sweeping out the 88 MB of grids, records and
    // digest windows that the next k_step (and the render itself) reads — k_step 0.067 -> 0.059 ms, the render 0.251 -> 0.248 ms on
    // the same buffer.  A smaller batch's observations stay in that cache and the hint only forces them out to HBM.  Measured
    // crossover, plain -> hint in M env-steps/s (profiles/r05_render_nontemporal.md): arena 10 240 envs (258 MB) 149 -> 131, 11 264
    // (284 MB) 145 -> 137, 12 288 (310 MB) 140 -> 141, 14 336 (361 MB) 137 -> 147; 20x20: 6 144 envs (275 MB) 82 -> 74, 8 192
    // (367 MB) 80 -> 82.  Hence 320 MB.  CTF_OBS_NT=0 / 1 forces it off / on.  (This is one handle's own share; the library applies the
    // rule to the sum over all live handles on the device — ctf_abi.hip: store_hint.)
    d->obs_store_nt = (int64_t)n_envs * d->obs_bytes > ((int64_t)320 << 20);
    if (const char* ov = getenv("CTF_OBS_NT")) d->obs_store_nt = atoi(ov) != 0;
    if (const char* ov = getenv("CTF_STEP_W")) {
        const int w = atoi(ov);
        if (w == 1 || w == 2 || w == 4 || w == 8) d->step_lanes_override = w;
    }
    d->heal = c->heal_per_step; d->tag_p = c->tag_probability; d->guard_mult = c->guardian_damage_multiplier;
    d->vault_cost = c->vault_hp_cost; d->vault_min = c->vault_min_hp;
    d->r_capture = c->reward_capture; d->r_step = c->reward_step; d->r_tag = c->reward_tag;
    d->win_scalar = c->win_margin_scalar; d->loss_scalar = c->loss_margin_scalar; d->punish = c->opp_capture_punishment;
    for (int t = 0; t < 4; t++) { d->type_hp[t] = c->type_hp[t]; d->type_damage[t] = c->type_damage[t]; }
    memcpy(d->team, c->agent_team, sizeof(d->team));
    memcpy(d->type, c->agent_type, sizeof(d->type));
    memcpy(d->opp, c->opponents, sizeof(d->opp));
    memcpy(d->flag_pos, c->flag_pos, sizeof(d->flag_pos));
    memcpy(d->capture_pos, c->capture_pos, sizeof(d->capture_pos));
    memcpy(d->spawn_pos, c->spawn_pos, sizeof(d->spawn_pos));
    memcpy(d->start_pos, c->start_pos, sizeof(d->start_pos));
    for (int t = 0; t < 2; t++) {
        d->flag_pack |= ((uint32_t)(uint8_t)c->flag_pos[t][0] | ((uint32_t)(uint8_t)c->flag_pos[t][1] << 8)) << (16 * t);
        d->capture_pack |= ((uint32_t)(uint8_t)c->capture_pos[t][0] | ((uint32_t)(uint8_t)c->capture_pos[t][1] << 8)) << (16 * t);
        d->spawn_pack |= ((uint32_t)(uint8_t)c->spawn_pos[t][0] | ((uint32_t)(uint8_t)c->spawn_pos[t][1] << 8)) << (16 * t);
    }
    for (int i = 0; i < N; i++) d->default_reverse |= (c->agent_team[i] == 1) ? (1 << i) : 0;
    for (int i = 0; i < N; i++) {
        d->team_mask |= (uint32_t)c->agent_team[i] << i;
        d->type_pack |= (uint32_t)c->agent_type[i] << (2 * i);
        uint64_t idx = 15;
        const int own = 1 - c->agent_team[i];  // OPPONENTS[1 - team] is the agent's own team
        for (int k = c->n_opponents[own] - 1; k >= 0; k--)
            if (c->opponents[own][k] == i) idx = (uint64_t)k;
        d->self_idx_pack |= idx << (4 * i);
    }
    for (int t = 0; t < 2; t++)
        for (int k = 0; k < c->n_opponents[t]; k++) d->opp_pack[t] |= (uint64_t)c->opponents[t][k] << (4 * k);

    // channel of every tile value as seen by a viewer of team t (reference gridworld_ctf.py:987-994: for a
    // team-1 viewer own/opponent agent tiles 4..7 <-> 8..11 and the flags 12 <-> 13 swap), 15 = no plane
    for (int t = 0; t < 2; t++) {
        uint64_t lut = 0;
        for (int v = 0; v < 16; v++) {
            int std_tile = v;
            if (t == 1) {
                if (v >= 4 && v <= 7) std_tile = v + 4;
                else if (v >= 8 && v <= 11) std_tile = v - 4;
                else if (v == 12) std_tile = 13;
                else if (v == 13) std_tile = 12;
            }
            uint64_t code = CTF_TILE_NONE;
            if (v >= 1 && v <= 13)
                for (int k = 1; k < C; k++)
                    if (c->tile_of_channel[k] == std_tile) code = (uint64_t)k;
            lut |= code << (4 * v);
        }
        d->chan_lut[t] = lut;
    }
    // metadata order (gridworld_ctf.py:1053-1067): own-team list minus self, then the opponents list
    for (int i = 0; i < N; i++) {
        int n = 0, team = c->agent_team[i];
        for (int k = 0; k < CTF_MAX_AGENTS; k++) d->meta_order[i][k] = -1;
        for (int k = 0; k < c->n_opponents[1 - team]; k++)
            if (c->opponents[1 - team][k] != i && n < N - 1) d->meta_order[i][n++] = c->opponents[1 - team][k];
        for (int k = 0; k < c->n_opponents[team]; k++)
            if (n < N - 1) d->meta_order[i][n++] = c->opponents[team][k];
    }
    return CTF_OK;
}

