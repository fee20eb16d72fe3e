/*
 * ctf_oracle.c — CPU restatement (plain C, scalar, one env at a time) of the reference
 * GridworldCtf step / reset / observation path.
 *
 * TEST INFRASTRUCTURE ONLY (see ctf_oracle.h).  Written from the behaviour of
 * /root/reference/gridworld_ctf.py; every function cites the lines it follows.
 * Parity: PINNED against the reference run in the build container (tests/golden/).
 *
 * Random numbers: the reference draws from two process-global MT19937 generators,
 *   - CPython `random` (shuffle -> _randbelow_with_getrandbits), gridworld_ctf.py:740
 *   - NumPy legacy `np.random` (rand(), randint()), gridworld_ctf.py:771,815
 * Both are restated here from the published algorithm (Matsumoto & Nishimura 1998; CPython
 * Lib/random.py 3.10 shuffle/_randbelow; NumPy 1.23 legacy random_sample / masked bounded uint32).
 */
#include "ctf_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* MT19937 (standard, eager block regeneration)                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[CTF_MT_N];
    uint32_t pos;
    /* counter mode (ctf_env.h CTF_RNG_COUNTER): the "generator" is a tape — word n = Philox4x32-10(key = seed, counter =
     * (n / 4, stream, "CTF1"))[n % 4], read by the same three functions below (py_randbelow, np_rand, np_randint), i.e. the
     * reference's random.shuffle / np.random.rand / np.random.randint with their word source patched to the tape */
    int counter_mode;
    uint32_t stream;
    uint64_t seed, n;
} mt_t;
static void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1);

static void mt_init_genrand(mt_t* g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < CTF_MT_N; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->pos = CTF_MT_N;
}

static void mt_init_by_array(mt_t* g, const uint32_t* key, int len) {
    mt_init_genrand(g, 19650218u);
    uint32_t* mt = g->mt;
    int i = 1, j = 0;
    int k = CTF_MT_N > len ? CTF_MT_N : len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= CTF_MT_N) { mt[0] = mt[CTF_MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = CTF_MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= CTF_MT_N) { mt[0] = mt[CTF_MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
    g->pos = CTF_MT_N;
}

static uint32_t mt_next(mt_t* g) {
    uint32_t* mt = g->mt;
    if (g->counter_mode) {
        uint32_t ctr[4] = {(uint32_t)(g->n >> 2), (uint32_t)((g->n >> 2) >> 32), g->stream, 0x43544631u};
        philox4x32_10(ctr, (uint32_t)g->seed, (uint32_t)(g->seed >> 32));
        return ctr[g->n++ & 3u];
    }
    if (g->pos >= CTF_MT_N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < CTF_MT_N - 397; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < CTF_MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - CTF_MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[CTF_MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[CTF_MT_N - 1] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        g->pos = 0;
    }
    uint32_t y = mt[g->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* CPython random.Random._randbelow_with_getrandbits(n), n >= 1 */
static uint32_t py_randbelow(mt_t* g, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++; /* n.bit_length() */
    uint32_t r = mt_next(g) >> (32 - k);  /* getrandbits(k), k <= 32 */
    while (r >= n) r = mt_next(g) >> (32 - k);
    return r;
}

/* NumPy legacy random_sample(): 53-bit double from two words */
static double np_rand(mt_t* g) {
    uint32_t a = mt_next(g) >> 5, b = mt_next(g) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* NumPy legacy randint(k) for a Python-int k >= 1: masked rejection on one uint32; k == 1 draws nothing */
static uint32_t np_randint(mt_t* g, uint32_t k) {
    uint32_t rng = k - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_next(g) & mask; } while (v > rng);
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* env                                                                                         */
/* ------------------------------------------------------------------------------------------ */
struct octf_env {
    ctf_config cfg;
    mt_t py, np;
    uint8_t grid[CTF_MAX_CELLS];
    int pos[CTF_MAX_AGENTS][2];
    double hp[CTF_MAX_AGENTS];
    uint8_t has_flag[CTF_MAX_AGENTS];
    int32_t inventory[CTF_MAX_AGENTS];
    uint8_t perm[CTF_MAX_AGENTS];
    int32_t step_count;
    int32_t done;
    int32_t team_captures[2];
    int32_t metrics[CTF_N_METRICS][CTF_MAX_AGENTS];
    uint8_t visitation[CTF_MAX_AGENTS][CTF_MAX_CELLS];
    /* per-step scratch */
    int capture_this_move;
    double capture_team_this_move[2];
};

/* tile codes, gridworld_ctf.py:250-270 */
enum { OPEN_TILE = 0, BLOCK_TILE = 1, DESTR1 = 2, DESTR2 = 3, FLAG_TILE0 = 12 };
static int agent_tile(const ctf_config* c, int i) { return 4 + c->agent_type[i] + 4 * c->agent_team[i]; }
static int flag_tile(int team) { return FLAG_TILE0 + team; }

/* ACTION_DELTAS, gridworld_ctf.py:100-145 */
static void action_delta(int type, int action, int* dr, int* dc) {
    static const int base[5][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}, {0, 0}};
    if (action <= 4) { *dr = base[action][0]; *dc = base[action][1]; return; }
    int scale = (type == 2) ? 2 : (type == 3) ? 1 : 0;
    *dr = base[action - 5][0] * scale;
    *dc = base[action - 5][1] * scale;
}

static int iabs(int x) { return x < 0 ? -x : x; }
/* max_dim_distance_to_xy / agent_distance_to_xy, gridworld_ctf.py:744-759 */
static int cheb(int r0, int c0, int r1, int c1) {
    int a = iabs(r0 - r1), b = iabs(c0 - c1);
    return a > b ? a : b;
}

octf_env* octf_create(const ctf_config* cfg) {
    octf_env* e = (octf_env*)calloc(1, sizeof(octf_env));
    if (!e) return NULL;
    e->cfg = *cfg;
    for (int i = 0; i < cfg->n_agents; i++) e->perm[i] = (uint8_t)i; /* _arr, :244 (once, not in reset) */
    octf_seed(e, 0, 0);
    octf_reset(e);
    return e;
}

void octf_destroy(octf_env* e) { free(e); }

void octf_seed(octf_env* e, uint64_t py_seed, uint64_t np_seed) {
    if (e->cfg.rng_mode == CTF_RNG_COUNTER) { /* both tapes from word 0 */
        e->py.counter_mode = e->np.counter_mode = 1;
        e->py.stream = 0; e->np.stream = 1;
        e->py.seed = py_seed; e->np.seed = np_seed;
        e->py.n = e->np.n = 0;
        return;
    }
    e->py.counter_mode = e->np.counter_mode = 0;
    uint32_t key[2] = {(uint32_t)py_seed, (uint32_t)(py_seed >> 32)};
    mt_init_by_array(&e->py, key, key[1] ? 2 : 1); /* CPython random_seed: 32-bit chunks of abs(seed) */
    mt_init_genrand(&e->np, (uint32_t)np_seed);      /* NumPy _legacy_seeding(int) */
}

void octf_get_rng_counters(const octf_env* e, uint64_t* out2) { out2[0] = e->py.n; out2[1] = e->np.n; }
void octf_set_rng_counters(octf_env* e, const uint64_t* in2) { e->py.n = in2[0]; e->np.n = in2[1]; }
void octf_set_rng_state(octf_env* e, const uint32_t* py_mt625, const uint32_t* np_mt625) {
    if (py_mt625) { memcpy(e->py.mt, py_mt625, 4 * CTF_MT_N); e->py.pos = py_mt625[CTF_MT_N]; }
    if (np_mt625) { memcpy(e->np.mt, np_mt625, 4 * CTF_MT_N); e->np.pos = np_mt625[CTF_MT_N]; }
}

void octf_get_rng_state(const octf_env* e, uint32_t* py_mt625, uint32_t* np_mt625) {
    if (py_mt625) { memcpy(py_mt625, e->py.mt, 4 * CTF_MT_N); py_mt625[CTF_MT_N] = e->py.pos; }
    if (np_mt625) { memcpy(np_mt625, e->np.mt, 4 * CTF_MT_N); np_mt625[CTF_MT_N] = e->np.pos; }
}

/* update_visitation_map, gridworld_ctf.py:479-486 (uint8 counters wrap) */
static void update_visitation(octf_env* e) {
    const ctf_config* c = &e->cfg;
    if (!c->log_metrics) return;
    for (int i = 0; i < c->n_agents; i++) e->visitation[i][e->pos[i][0] * c->grid_size + e->pos[i][1]]++;
}

/* reset, gridworld_ctf.py:383-477 (grid comes pre-painted in cfg->init_grid: load_scenario :352-381) */
void octf_reset(octf_env* e) {
    const ctf_config* c = &e->cfg;
    int cells = c->grid_size * c->grid_size;
    e->step_count = 0;
    e->done = 0;
    memcpy(e->grid, c->init_grid, (size_t)cells);
    for (int i = 0; i < c->n_agents; i++) {
        e->pos[i][0] = c->start_pos[i][0];
        e->pos[i][1] = c->start_pos[i][1];
        e->has_flag[i] = 0;
        e->hp[i] = c->type_hp[c->agent_type[i]];
        e->inventory[i] = 0;
    }
    e->capture_this_move = 0;
    e->capture_team_this_move[0] = e->capture_team_this_move[1] = 0;
    e->team_captures[0] = e->team_captures[1] = 0;
    memset(e->metrics, 0, sizeof(e->metrics));
    memset(e->visitation, 0, sizeof(e->visitation));
    update_visitation(e);
}

/* dice_roll, gridworld_ctf.py:734-742: random.shuffle(self._arr) */
static void dice_roll(octf_env* e) {
    int n = e->cfg.n_agents;
    for (int i = n - 1; i >= 1; i--) {
        uint32_t j = py_randbelow(&e->py, (uint32_t)i + 1u);
        uint8_t t = e->perm[i]; e->perm[i] = e->perm[j]; e->perm[j] = t;
    }
}

#define GRID(e, r, c_) ((e)->grid[(r) * (e)->cfg.grid_size + (c_)])
#define METRIC(e, m, i) do { if ((e)->cfg.log_metrics) (e)->metrics[m][i]++; } while (0)

/* movement_handler, gridworld_ctf.py:569-612 */
static void movement_handler(octf_env* e, int a, int nr, int nc) {
    const ctf_config* c = &e->cfg;
    int team = c->agent_team[a];
    GRID(e, e->pos[a][0], e->pos[a][1]) = OPEN_TILE;
    GRID(e, nr, nc) = (uint8_t)agent_tile(c, a);
    e->pos[a][0] = nr;
    e->pos[a][1] = nc;
    const int8_t* of = c->flag_pos[1 - team]; /* opponents' flag */
    const int8_t* hf = c->flag_pos[team];     /* home flag       */
    /* flag pickup :583-591 — the flag cell turns into a BLOCK tile while carried */
    if (cheb(nr, nc, of[0], of[1]) <= 1 && GRID(e, of[0], of[1]) == flag_tile(1 - team)) {
        e->has_flag[a] = 1;
        GRID(e, of[0], of[1]) = BLOCK_TILE;
        METRIC(e, CTF_M_FLAG_PICKUPS, a);
    }
    /* flag capture :594-610 */
    if (cheb(nr, nc, hf[0], hf[1]) <= 1 && e->has_flag[a] == 1) {
        if (!c->home_flag_capture || GRID(e, hf[0], hf[1]) == flag_tile(team)) {
            e->has_flag[a] = 0;
            GRID(e, of[0], of[1]) = (uint8_t)flag_tile(1 - team);
            e->team_captures[team]++;
            METRIC(e, CTF_M_FLAG_CAPTURES, a);
            e->capture_this_move = 1;
            e->capture_team_this_move[team] = 1.0;
        }
    }
}

/* act, gridworld_ctf.py:700-732 (with is_valid_move :636, move_to_open_tile :643, update_vaulter_hp :652,
 * can_add_blocks :659 / add_block :614, can_mine_blocks :669 / mine_block :677) */
static double act(octf_env* e, int a, int action) {
    const ctf_config* c = &e->cfg;
    int G = c->grid_size, type = c->agent_type[a], team = c->agent_team[a];
    int dr, dc;
    action_delta(type, action, &dr, &dc);
    int nr = e->pos[a][0] + dr, nc = e->pos[a][1] + dc;
    if (nr >= 0 && nr < G && nc >= 0 && nc < G) {
        int cell = GRID(e, nr, nc);
        if (cell == OPEN_TILE &&
            (action <= 3 || (action >= 5 && type == 2 && (e->hp[a] - c->vault_hp_cost) > c->vault_min_hp))) {
            movement_handler(e, a, nr, nc);
            if (action >= 5 && type == 2) e->hp[a] -= c->vault_hp_cost;
        } else if (action >= 5 && type == 3 && e->inventory[a] > 0 && cell == OPEN_TILE &&
                   cheb(nr, nc, c->spawn_pos[team][0], c->spawn_pos[team][1]) > 1 &&
                   cheb(nr, nc, c->spawn_pos[1 - team][0], c->spawn_pos[1 - team][1]) > 1) {
            GRID(e, nr, nc) = DESTR1;
            e->inventory[a]--;
            if (c->log_metrics) {
                int d_own = cheb(e->pos[a][0], e->pos[a][1], c->capture_pos[team][0], c->capture_pos[team][1]);
                int d_opp = cheb(e->pos[a][0], e->pos[a][1], c->capture_pos[1 - team][0], c->capture_pos[1 - team][1]);
                e->metrics[CTF_M_BLOCKS_LAID][a]++;
                e->metrics[CTF_M_BLOCKS_LAID_DIST_OWN_FLAG][a] += d_own;
                e->metrics[CTF_M_BLOCKS_LAID_DIST_OPP_FLAG][a] += d_opp;
            }
        } else if (action < 5 && type == 3 && (cell == DESTR1 || cell == DESTR2)) {
            if (cell == DESTR1) {
                GRID(e, nr, nc) = DESTR2;
            } else {
                GRID(e, nr, nc) = OPEN_TILE;
                if (e->inventory[a] < 1000) e->inventory[a]++; /* MAX_AGENT_BLOCKS :241 */
                METRIC(e, CTF_M_BLOCKS_MINED, a);
            }
        }
    }
    double reward = c->reward_step;
    if (e->capture_this_move) {
        reward += c->reward_capture;
        e->capture_this_move = 0;
    }
    return reward;
}

/* respawn, gridworld_ctf.py:761-794 */
static uint32_t respawn(octf_env* e, int o) {
    const ctf_config* c = &e->cfg;
    int G = c->grid_size, team = c->agent_team[o];
    int x = c->spawn_pos[team][0], y = c->spawn_pos[team][1];
    int r0 = x - 1 > 0 ? x - 1 : 0, c0 = y - 1 > 0 ? y - 1 : 0;
    int r1 = x + 2 < G ? x + 2 : G, c1 = y + 2 < G ? y + 2 : G;
    int cand[9][2], k = 0;
    for (int r = r0; r < r1; r++)
        for (int cc = c0; cc < c1; cc++)
            if (GRID(e, r, cc) == OPEN_TILE) { cand[k][0] = r - r0; cand[k][1] = cc - c0; k++; }
    if (k == 0) return CTF_ST_NO_RESPAWN; /* reference: ValueError from np.random.randint(0) */
    uint32_t rnd = np_randint(&e->np, (uint32_t)k);
    /* offsets are relative to the CLIPPED window yet "-1" is applied regardless (the WARNING at :773) */
    int nr = x + cand[rnd][0] - 1, nc = y + cand[rnd][1] - 1;
    uint32_t st = 0;
    if (nr < 0 || nc < 0) { st |= CTF_ST_SPAWN_EDGE; nr = nr < 0 ? nr + G : nr; nc = nc < 0 ? nc + G : nc; }
    int orow = e->pos[o][0], ocol = e->pos[o][1];
    GRID(e, orow, ocol) = OPEN_TILE;
    GRID(e, nr, nc) = (uint8_t)agent_tile(c, o);
    e->pos[o][0] = nr;
    e->pos[o][1] = nc;
    e->hp[o] = c->type_hp[c->agent_type[o]];
    if (e->has_flag[o] == 1) {
        e->has_flag[o] = 0;
        if (c->drop_flag_when_no_hp)
            GRID(e, orow, ocol) = (uint8_t)flag_tile(1 - team);
        else
            GRID(e, c->flag_pos[1 - team][0], c->flag_pos[1 - team][1]) = (uint8_t)flag_tile(1 - team);
    }
    return st;
}

/* tagging_logic, gridworld_ctf.py:796-837 */
static double tagging_logic(octf_env* e, int a, uint32_t* status) {
    const ctf_config* c = &e->cfg;
    int type = c->agent_type[a], team = c->agent_team[a];
    double tagging_reward = 0;
    if (c->type_damage[type] > 0) {
        double mult = 1.0;
        if (cheb(e->pos[a][0], e->pos[a][1], c->flag_pos[team][0], c->flag_pos[team][1]) <= 3 && type == 1)
            mult = c->guardian_damage_multiplier; /* GUARDIAN_DEFENSE_DISTANCE = 3, :226 */
        for (int k = 0; k < c->n_opponents[team]; k++) {
            int o = c->opponents[team][k];
            double u = np_rand(&e->np); /* drawn first, unconditionally (:815) */
            if (u < c->tag_probability &&
                cheb(e->pos[a][0], e->pos[a][1], e->pos[o][0], e->pos[o][1]) <= 1) { /* GUARDIAN_TAGGING_RANGE = 1 */
                e->hp[o] -= c->type_damage[type] * mult;
                METRIC(e, CTF_M_TAG_COUNT, a);
                if (e->hp[o] <= 0) {
                    if (e->has_flag[o] == 1) METRIC(e, CTF_M_FLAG_DISPOSSESSIONS, a);
                    *status |= respawn(e, o);
                    tagging_reward = c->reward_tag;
                    METRIC(e, CTF_M_RESPAWN_TAG_COUNT, a);
                }
            }
        }
    }
    return tagging_reward;
}

/* step, gridworld_ctf.py:849-918 */
uint32_t octf_step(octf_env* e, const int8_t* actions, double* rewards, uint8_t* done) {
    const ctf_config* c = &e->cfg;
    int n = c->n_agents;
    uint32_t status = 0;
    double rw[CTF_MAX_AGENTS];
    e->step_count++;
    e->capture_team_this_move[0] = e->capture_team_this_move[1] = 0;
    for (int i = 0; i < n; i++) rw[i] = 0;

    dice_roll(e);
    for (int k = 0; k < n; k++) {
        int a = e->perm[k];
        int team = c->agent_team[a];
        int action = actions[a];
        if (action < 0 || action >= CTF_N_ACTIONS) { status |= CTF_ST_BAD_ACTION; action = 4; } /* reference: KeyError */
        rw[a] = act(e, a, action);
        rw[a] += tagging_logic(e, a, &status);
        /* metric-only section :879-902 */
        if (c->log_metrics) {
            int pr = e->pos[a][0], pc = e->pos[a][1];
            if (cheb(pr, pc, c->capture_pos[team][0], c->capture_pos[team][1]) <= 3) e->metrics[CTF_M_STEPS_DEFENDING_ZONE][a]++;
            if (cheb(pr, pc, c->capture_pos[1 - team][0], c->capture_pos[1 - team][1]) <= 3) e->metrics[CTF_M_STEPS_ATTACKING_ZONE][a]++;
            for (int j = 0; j < c->n_opponents[1 - team]; j++) { /* "teammates" = OPPONENTS[1-team], includes self */
                int m = c->opponents[1 - team][j];
                if (cheb(pr, pc, e->pos[m][0], e->pos[m][1]) <= 1) e->metrics[CTF_M_STEPS_ADJ_TEAMMATE][a]++;
            }
            for (int j = 0; j < c->n_opponents[team]; j++) {
                int o = c->opponents[team][j];
                if (cheb(pr, pc, e->pos[o][0], e->pos[o][1]) <= 1) e->metrics[CTF_M_STEPS_ADJ_OPPONENT][a]++;
            }
        }
    }

    /* heal_agents :839-847 — a second shuffle, then heal */
    dice_roll(e);
    for (int k = 0; k < n; k++) {
        int a = e->perm[k];
        double mx = c->type_hp[c->agent_type[a]];
        if (e->hp[a] < mx) {
            e->hp[a] += c->heal_per_step;
            if (e->hp[a] > mx) e->hp[a] = mx;
        }
    }

    /* get_adjusted_rewards :957-966 */
    if (c->use_adjusted_rewards)
        for (int i = 0; i < n; i++)
            rw[i] -= e->capture_team_this_move[1 - c->agent_team[i]] * c->reward_capture * c->opp_capture_punishment;

    update_visitation(e);

    /* end of game :914-916, get_terminal_rewards :920-940 */
    if (e->step_count == c->game_steps) {
        e->done = 1;
        int c0 = e->team_captures[0], c1 = e->team_captures[1];
        int margin = iabs(c0 - c1);
        int winner = c0 > c1 ? 0 : (c0 < c1 ? 1 : -1);
        if (winner >= 0)
            for (int i = 0; i < n; i++) {
                if (c->agent_team[i] == winner) rw[i] += margin * c->win_margin_scalar;
                else rw[i] -= margin * c->loss_margin_scalar;
            }
    }
    for (int i = 0; i < n; i++) rewards[i] = rw[i];
    if (done) *done = (uint8_t)e->done;
    return status;
}

/* ------------------------------------------------------------------------------------------ */
/* observation                                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* NumPy npy_double_to_half: direct round-to-nearest-even f64 -> binary16 */
uint16_t octf_f64_to_f16(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
    int64_t be = (int64_t)((b >> 52) & 0x7FF);
    uint64_t m = b & 0xFFFFFFFFFFFFFull;
    if (be == 0x7FF) return (uint16_t)(sign | 0x7C00u | (m ? (0x200u | (uint16_t)(m >> 42)) : 0u));
    int64_t E = be - 1023;
    if (be == 0) return sign; /* f64 subnormals and zero are far below half's range */
    if (E > 15) return (uint16_t)(sign | 0x7C00u);
    if (E >= -14) {
        uint32_t h = (uint32_t)(((E + 15) << 10) | (int64_t)(m >> 42));
        uint64_t rem = m & ((1ull << 42) - 1), half = 1ull << 41;
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    if (E < -25) return sign;
    uint64_t full = m | (1ull << 52);
    int shift = (int)(28 - E); /* 43..53 */
    uint64_t h = full >> shift, rem = full & ((1ull << shift) - 1), half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

/* standardise_state, gridworld_ctf.py:975-1009, for agent i; out = u8[C][G][G] */
static void standardise_state(const octf_env* e, int i, int reverse, uint8_t* out) {
    const ctf_config* c = &e->cfg;
    int G = c->grid_size, C = c->n_channels, team = c->agent_team[i];
    memset(out, 0, (size_t)C * G * G);
    for (int r = 0; r < G; r++)
        for (int cc = 0; cc < G; cc++) {
            int v = e->grid[r * G + cc];
            /* relabel to the viewer's team (:987-994): for team 1 own/opponent tiles and flags swap */
            if (team == 1) {
                if (v >= 4 && v <= 7) v += 4;
                else if (v >= 8 && v <= 11) v -= 4;
                else if (v == 12) v = 13;
                else if (v == 13) v = 12;
            }
            /* destination cell after the optional flip (:1003-1007) */
            int dr = r, dc = cc;
            if (reverse) {
                if (c->flip_axis == -1) { dr = G - 1 - r; dc = G - 1 - cc; }      /* np.flip(plane, None) */
                else if (c->flip_axis == 0) { dr = G - 1 - r; }                   /* np.flip(plane, 0)    */
                else if (c->flip_axis == 1) { dc = G - 1 - cc; }                  /* np.flip(plane, 1)    */
                else { dr = G - 1 - cc; dc = G - 1 - r; }                         /* np.rot90(plane.T, 2) */
            }
            if (r == e->pos[i][0] && cc == e->pos[i][1]) out[dr * G + dc] = 1; /* plane 0: own position */
            if (v != 0)
                for (int k = 1; k < C; k++)
                    if (c->tile_of_channel[k] == v) out[(k * G + dr) * G + dc] = 1;
        }
}

/* get_env_metadata, gridworld_ctf.py:1027-1069, for agent i; out = f16 bits [2N+6] */
static void env_metadata(const octf_env* e, int i, uint16_t* out) {
    const ctf_config* c = &e->cfg;
    int n = c->n_agents, team = c->agent_team[i], M = 2 * n + 6;
    uint8_t hpq[CTF_MAX_AGENTS];
    /* the quirk at :1039-1041: agent_hp is indexed by the TYPE id of agent j, then truncated to uint8 */
    for (int j = 0; j < n; j++) {
        int v = c->agent_type[j];
        double q = (v < n) ? e->hp[v] / c->type_hp[c->agent_type[j]] : 0.0;
        hpq[j] = (uint8_t)(int64_t)q;
    }
    double m[2 * CTF_MAX_AGENTS + 6];
    for (int k = 0; k < M; k++) m[k] = 0;
    m[0] = (double)e->step_count / (double)c->game_steps;
    m[1] = (double)(e->team_captures[team] + 1) / (double)(e->team_captures[1 - team] + 1);
    m[2 + c->agent_type[i]] = 1.0;
    m[6] = hpq[i];
    m[7] = e->has_flag[i];
    int idx = 8;
    for (int k = 0; k < c->n_opponents[1 - team]; k++) {
        int t = c->opponents[1 - team][k];
        if (t != i && idx + 1 < M) { m[idx++] = hpq[t]; m[idx++] = e->has_flag[t]; }
    }
    for (int k = 0; k < c->n_opponents[team]; k++) {
        int o = c->opponents[team][k];
        if (idx + 1 < M) { m[idx++] = hpq[o]; m[idx++] = e->has_flag[o]; }
    }
    for (int k = 0; k < M; k++) out[k] = octf_f64_to_f16(m[k]);
}

void octf_observe(const octf_env* e, uint8_t* obs, uint16_t* meta, uint32_t reverse_mask) {
    const ctf_config* c = &e->cfg;
    int n = c->n_agents, G = c->grid_size, C = c->n_channels, M = 2 * n + 6;
    for (int i = 0; i < n; i++) {
        int rev = (reverse_mask == CTF_REVERSE_DEFAULT) ? (c->agent_team[i] == 1) : (int)((reverse_mask >> i) & 1u);
        if (obs) standardise_state(e, i, rev, obs + (size_t)i * C * G * G);
        if (meta) env_metadata(e, i, meta + (size_t)i * M);
    }
}

void octf_get_state(const octf_env* e, ctf_state_view* out) {
    memset(out, 0, sizeof(*out));
    int cells = e->cfg.grid_size * e->cfg.grid_size;
    memcpy(out->grid, e->grid, (size_t)cells);
    for (int i = 0; i < e->cfg.n_agents; i++) {
        out->pos[i][0] = (int8_t)e->pos[i][0];
        out->pos[i][1] = (int8_t)e->pos[i][1];
        out->hp[i] = e->hp[i];
        out->has_flag[i] = e->has_flag[i];
        out->inventory[i] = e->inventory[i];
        out->perm[i] = e->perm[i];
        memcpy(out->visitation[i], e->visitation[i], (size_t)cells);
    }
    out->step_count = e->step_count;
    out->done = e->done;
    out->team_captures[0] = e->team_captures[0];
    out->team_captures[1] = e->team_captures[1];
    memcpy(out->metrics, e->metrics, sizeof(out->metrics));
}

void octf_set_state(octf_env* e, const ctf_state_view* in) {
    int cells = e->cfg.grid_size * e->cfg.grid_size;
    memcpy(e->grid, in->grid, (size_t)cells);
    for (int i = 0; i < e->cfg.n_agents; i++) {
        e->pos[i][0] = in->pos[i][0];
        e->pos[i][1] = in->pos[i][1];
        e->hp[i] = in->hp[i];
        e->has_flag[i] = in->has_flag[i];
        e->inventory[i] = in->inventory[i];
        e->perm[i] = in->perm[i];
        memcpy(e->visitation[i], in->visitation[i], (size_t)cells);
    }
    e->step_count = in->step_count;
    e->done = in->done;
    e->team_captures[0] = in->team_captures[0];
    e->team_captures[1] = in->team_captures[1];
    memcpy(e->metrics, in->metrics, sizeof(e->metrics));
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic actions (Philox4x32-10, Salmon et al. 2011) + batch driver                        */
/* ------------------------------------------------------------------------------------------ */
static void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0], p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1, n3 = (uint32_t)p0;
        ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void octf_philox_actions(int8_t* actions, int32_t n_agents, uint64_t seed, uint32_t step, uint32_t env_index) {
    for (int blk = 0; blk * 8 < n_agents; blk++) {
        uint32_t ctr[4] = {env_index, step, (uint32_t)blk, 0u};
        philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int j = 0; j < 8 && blk * 8 + j < n_agents; j++) {
            uint32_t h = (ctr[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            actions[blk * 8 + j] = (int8_t)((h * 9u) >> 16);
        }
    }
}

uint64_t octf_run_batch(const ctf_config* cfg, int32_t n_envs, int32_t n_steps, uint64_t seed_base,
                        uint64_t action_seed, int32_t with_observe, int32_t n_threads) {
    uint64_t total = 0;
    size_t obs_bytes = (size_t)cfg->n_agents * cfg->n_channels * cfg->grid_size * cfg->grid_size;
    size_t meta_elems = (size_t)cfg->n_agents * (2 * cfg->n_agents + 6);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(+ : total)
    for (int32_t ei = 0; ei < n_envs; ei++) {
        octf_env* e = octf_create(cfg);
        uint8_t* obs = (uint8_t*)malloc(obs_bytes);
        uint16_t* meta = (uint16_t*)malloc(meta_elems * 2);
        int8_t actions[CTF_MAX_AGENTS];
        double rewards[CTF_MAX_AGENTS];
        uint8_t done;
        uint64_t acc = 0;
        octf_seed(e, seed_base + (uint64_t)ei, seed_base + (uint64_t)ei);
        for (int32_t t = 0; t < n_steps; t++) {
            octf_philox_actions(actions, cfg->n_agents, action_seed, (uint32_t)t, (uint32_t)ei);
            if (e->done) octf_reset(e);
            octf_step(e, actions, rewards, &done);
            for (int i = 0; i < cfg->n_agents; i++) acc += (uint64_t)(int64_t)(rewards[i] * 16.0);
            if (with_observe) {
                octf_observe(e, obs, meta, CTF_REVERSE_DEFAULT);
                for (size_t b = 0; b < obs_bytes; b += 97) acc += obs[b];
                acc += meta[0];
            }
        }
        total += acc + (uint64_t)e->grid[0];
        free(obs); free(meta);
        octf_destroy(e);
    }
    return total;
}


/* ------------------------------------------------------------------------------------------ */
/* bench.py's protocol on the CPU, every env, reduced to digests (tests/test_gpu_parity.py)     */
/* ------------------------------------------------------------------------------------------ */
/* weight of element i of a digest: sum of value_i * w(i) modulo 2^64 (what torch's wrapping int64 arithmetic gives too) */
static uint64_t digest_w(uint64_t i) { return (i + 1u) * 0x9E3779B97F4A7C15ull; }

/*
 * For each env e < n_envs (seeded seeds[e] for both generators, actions = the Philox stream of global env index env_offset + e):
 *   1. bench.stagger_phases: `period` steps with action seed stagger_seed and auto-reset, a reset() of env e after step s
 *      when e % period == s;
 *   2. n_steps more steps with action seed action_seed and auto-reset; after each of them the digests of that env's float64
 *      rewards (their bit patterns), observation block, metadata rows (f16 bits) and its done flag.
 * At the end: the digests of both generators' states (624 words + position), env_step_count, the two capture counters.
 * out arrays: obs_d / meta_d / rew_d u64 [n_steps][n_envs], done u8 [n_steps][n_envs], rng_d u64 [n_envs][2], misc i32 [n_envs][3].
 */
void octf_bench_digest(const ctf_config* cfg, int32_t n_envs, const uint64_t* seeds, uint32_t env_offset, int32_t period,
                       uint64_t stagger_seed, int32_t n_steps, uint64_t action_seed, int32_t n_threads, uint64_t* obs_d,
                       uint64_t* meta_d, uint64_t* rew_d, uint8_t* done_out, uint64_t* rng_d, int32_t* misc) {
    const size_t obs_bytes = (size_t)cfg->n_agents * cfg->n_channels * cfg->grid_size * cfg->grid_size;
    const size_t meta_elems = (size_t)cfg->n_agents * (2 * cfg->n_agents + 6);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 16)
    for (int32_t ei = 0; ei < n_envs; ei++) {
        octf_env* e = octf_create(cfg);
        uint8_t* obs = (uint8_t*)malloc(obs_bytes);
        uint16_t* meta = (uint16_t*)malloc(meta_elems * 2);
        int8_t actions[CTF_MAX_AGENTS];
        double rewards[CTF_MAX_AGENTS];
        uint8_t done;
        octf_seed(e, seeds[ei], seeds[ei]);
        for (int32_t s = 0; s < period; s++) {
            octf_philox_actions(actions, cfg->n_agents, stagger_seed, (uint32_t)s, env_offset + (uint32_t)ei);
            if (e->done) octf_reset(e);
            octf_step(e, actions, rewards, &done);
            if (ei % period == s) octf_reset(e);
        }
        for (int32_t t = 0; t < n_steps; t++) {
            octf_philox_actions(actions, cfg->n_agents, action_seed, (uint32_t)t, env_offset + (uint32_t)ei);
            if (e->done) octf_reset(e);
            octf_step(e, actions, rewards, &done);
            octf_observe(e, obs, meta, CTF_REVERSE_DEFAULT);
            uint64_t a = 0, b = 0, c = 0;
            for (size_t i = 0; i < obs_bytes; i++) a += obs[i] ? digest_w(i) * obs[i] : 0;
            for (size_t i = 0; i < meta_elems; i++) b += (uint64_t)meta[i] * digest_w(i);
            for (int i = 0; i < cfg->n_agents; i++) {
                uint64_t bits;
                memcpy(&bits, &rewards[i], 8);
                c += bits * digest_w((uint64_t)i);
            }
            obs_d[(size_t)t * n_envs + ei] = a;
            meta_d[(size_t)t * n_envs + ei] = b;
            rew_d[(size_t)t * n_envs + ei] = c;
            done_out[(size_t)t * n_envs + ei] = done;
        }
        uint32_t py[CTF_MT_N + 1], np_[CTF_MT_N + 1];
        octf_get_rng_state(e, py, np_);
        uint64_t r0 = 0, r1 = 0;
        for (int i = 0; i <= CTF_MT_N; i++) { r0 += (uint64_t)py[i] * digest_w((uint64_t)i); r1 += (uint64_t)np_[i] * digest_w((uint64_t)i); }
        rng_d[2 * (size_t)ei] = r0;
        rng_d[2 * (size_t)ei + 1] = r1;
        misc[3 * (size_t)ei] = e->step_count;
        misc[3 * (size_t)ei + 1] = e->team_captures[0];
        misc[3 * (size_t)ei + 2] = e->team_captures[1];
        free(obs); free(meta);
        octf_destroy(e);
    }
}
