/*
 * ctf_env.h — C ABI of the batched GridworldCtf hot path on MI355X (gfx950).
 *
 * This header is the drop-in boundary.  The reference (g-nightingale/marl-ctf-development) is
 * pure Python and has no FFI of its own: its boundary is the class `GridworldCtf`
 * (gridworld_ctf.py:13).  Every entry point below names the reference method(s) whose work it
 * replaces; the Python facade in `marl-ctf-development_amd/gridworld_ctf.py` binds them with
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch / C++ types cross this ABI;
 *   - every `*_dev` pointer is a caller-owned DEVICE pointer (e.g. torch tensor .data_ptr());
 *     the library never allocates or frees I/O buffers;
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream); calls that take a
 *     stream only enqueue work on it and return;
 *   - ctf_step, ctf_observe, ctf_observe_codes, ctf_step_observe and ctf_reset are kernel launches and nothing else (no
 *     allocation, no copy, no synchronisation, no host-side state that moves from call to call), so a caller may capture them
 *     into a hipGraph on `stream` and replay it: every replay is the next step (tests/test_gpu_hipgraph.py);
 *   - return value 0 = OK, negative = error (see CTF_E_*); `ctf_last_error()` has the text;
 *   - a handle is not thread-safe; distinct handles are independent (one per GPU / shard).
 */
#ifndef CTF_ENV_H
#define CTF_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTF_ABI_VERSION 2

#define CTF_MAX_AGENTS 16   /* N  <= 16                                   */
#define CTF_MAX_GRID 32     /* G  <= 32 (G*G <= 1024 cells)               */
#define CTF_MAX_CELLS (CTF_MAX_GRID * CTF_MAX_GRID)
#define CTF_MAX_CHANNELS 16 /* C  <= 16 (tiles 1..13 + own-position plane) */
#define CTF_N_ACTIONS 9     /* gridworld_ctf.py:100-145: actions 0..8      */
#define CTF_N_METRICS 13    /* per-agent counters, gridworld_ctf.py:456-468 */
#define CTF_MT_N 624        /* MT19937 state words                         */

/* error codes */
#define CTF_OK 0
#define CTF_E_INVALID (-1) /* bad argument / config                        */
#define CTF_E_HIP (-2)     /* a HIP runtime call failed                    */
#define CTF_E_NOMEM (-3)
#define CTF_E_RANGE (-4)   /* env index out of range                       */

/* sticky per-handle device status bits (ctf_status) — situations in which the reference raises */
#define CTF_ST_BAD_ACTION 1u   /* action outside 0..8: reference raises KeyError (gridworld_ctf.py:710) */
#define CTF_ST_NO_RESPAWN 2u   /* no open cell round the spawn: np.random.randint(0) ValueError (:771)  */
#define CTF_ST_SPAWN_EDGE 4u   /* respawn offset went negative (spawn on row/col 0, the WARNING at :773) */
#define CTF_ST_RNG_OVERRUN 8u  /* one step drew more than ~624 words from one generator: its rejection loops cannot be
                                  followed further (a run of > 500 rejected draws: does not happen)                  */

/* ctf_config.rng_mode */
#define CTF_RNG_MT19937 0 /* the reference's two MT19937 generators, bit for bit (default)                          */
#define CTF_RNG_COUNTER 1 /* opt-in: counter-based streams — word n of a stream = Philox4x32-10(key = the stream's
                             seed, counter = (n / 4, stream, "CTF1"))[n % 4]; the draws are made from those words by the
                             reference's own rules (random.shuffle, np.random.rand, np.random.randint): a reference run
                             whose three functions read the same tape gives the same trajectory                     */

/* ctf_step flags */
#define CTF_STEP_AUTO_RESET 1u /* an env whose `done` is set is reset before it is stepped (not in the
                                  reference, which never auto-resets; off by default)               */

/* indices of the per-agent counters in ctf_state_view.metrics (gridworld_ctf.py:456-468) */
enum {
    CTF_M_TAG_COUNT = 0,
    CTF_M_RESPAWN_TAG_COUNT = 1,
    CTF_M_FLAG_PICKUPS = 2,
    CTF_M_FLAG_CAPTURES = 3,
    CTF_M_FLAG_DISPOSSESSIONS = 4,
    CTF_M_BLOCKS_LAID = 5,
    CTF_M_BLOCKS_MINED = 6,
    CTF_M_BLOCKS_LAID_DIST_OWN_FLAG = 7,
    CTF_M_BLOCKS_LAID_DIST_OPP_FLAG = 8,
    CTF_M_STEPS_DEFENDING_ZONE = 9,
    CTF_M_STEPS_ATTACKING_ZONE = 10,
    CTF_M_STEPS_ADJ_TEAMMATE = 11,
    CTF_M_STEPS_ADJ_OPPONENT = 12
};

/*
 * Flat description of one GridworldCtf configuration.  Built in Python by the facade from the
 * reference's constructor kwargs (gridworld_ctf.py:19-52) and scenario dict (scenarios.py), so the
 * numpy-slice painting of load_scenario (:352-381) and the set-ordering of get_tiles_used (:488-499)
 * stay on the host side of the boundary.
 */
typedef struct ctf_config {
    int32_t abi_version;          /* = CTF_ABI_VERSION                                          */
    int32_t n_agents;             /* N_AGENTS (:68)                                             */
    int32_t grid_size;            /* GRID_SIZE after load_scenario (:359)                        */
    int32_t n_channels;           /* len(TILES_USED)+1 (:997)                                   */
    int32_t game_steps;           /* GAME_STEPS (:60)                                           */
    int32_t flip_axis;            /* FLIP_AXIS (:358): -1 = None (both axes), 0, 1, 2           */
    int32_t home_flag_capture;    /* HOME_FLAG_CAPTURE (:65)                                    */
    int32_t use_adjusted_rewards; /* USE_ADJUSTED_REWARDS (:82)                                 */
    int32_t drop_flag_when_no_hp; /* DROP_FLAG_WHEN_NO_HP (:64)                                 */
    int32_t log_metrics;          /* 1 = keep counters + visitation maps (always on in the reference) */
    int32_t n_opponents[2];       /* len(OPPONENTS[t]) (:392-395)                               */
    int32_t rng_mode;             /* CTF_RNG_MT19937 (the reference's generators) or CTF_RNG_COUNTER */
    int32_t reserved0[5];         /* keeps the doubles 8-byte aligned without implicit padding */

    double heal_per_step;         /* AGENT_HP_HEALING_PER_STEP (:212)                           */
    double tag_probability;       /* TAG_PROBABILITY (:231)                                     */
    double guardian_damage_multiplier; /* (:228)                                                */
    double vault_hp_cost;         /* (:234)                                                     */
    double vault_min_hp;          /* (:235)                                                     */
    double reward_capture;        /* REWARD_CAPTURE = 1 (:77)                                   */
    double reward_step;           /* REWARD_STEP = 0 (:78)                                      */
    double reward_tag;            /* REWARD_TAG = 0.0 (:79)                                     */
    double win_margin_scalar;     /* 0.1 (:75)                                                  */
    double loss_margin_scalar;    /* 0.0 (:76)                                                  */
    double opp_capture_punishment;/* OPP_FLAG_CAPTURE_PUNISHMENT_SCALAR = 0.5 (:80)             */
    double type_hp[4];            /* AGENT_TYPE_HP (:209)                                       */
    double type_damage[4];        /* AGENT_TYPE_DAMAGE (:215)                                   */

    int8_t agent_team[CTF_MAX_AGENTS];     /* AGENT_TEAMS (:203)                                */
    int8_t agent_type[CTF_MAX_AGENTS];     /* AGENT_TYPES (:206): 0 scout 1 guardian 2 vaulter 3 miner */
    int8_t opponents[2][CTF_MAX_AGENTS];   /* OPPONENTS[t][k] (:392-395), ascending agent idx   */
    int8_t flag_pos[2][2];                 /* FLAG_POSITIONS[t] = (row, col) (:360)             */
    int8_t capture_pos[2][2];              /* CAPTURE_POSITIONS[t] (:361)                       */
    int8_t spawn_pos[2][2];                /* SPAWN_POSITIONS[t] (:362)                         */
    int8_t start_pos[CTF_MAX_AGENTS][2];   /* AGENT_STARTING_POSITIONS[i] (:363)                */
    uint8_t tile_of_channel[CTF_MAX_CHANNELS]; /* [k] = TILES_USED[k-1] for k>=1; [0] unused    */
    uint8_t init_grid[CTF_MAX_CELLS];      /* grid painted by load_scenario, row-major G*G      */
} ctf_config;

/* Host-side copy of ONE env's state (attributes the reference exposes: grid, agent_positions,
 * agent_hp, has_flag, block_inventory, _arr, env_step_count, done, metrics). */
typedef struct ctf_state_view {
    uint8_t grid[CTF_MAX_CELLS];           /* self.grid, row-major G*G                          */
    int8_t pos[CTF_MAX_AGENTS][2];         /* self.agent_positions[i] = (row, col)              */
    double hp[CTF_MAX_AGENTS];             /* self.agent_hp[i]                                  */
    uint8_t has_flag[CTF_MAX_AGENTS];      /* self.has_flag[i]                                  */
    int32_t inventory[CTF_MAX_AGENTS];     /* self.block_inventory[i]                           */
    uint8_t perm[CTF_MAX_AGENTS];          /* self._arr (persists across reset, :244)           */
    int32_t step_count;                    /* self.env_step_count                               */
    int32_t done;                          /* self.done                                         */
    int32_t team_captures[2];              /* metrics['team_flag_captures'][t]                  */
    int32_t metrics[CTF_N_METRICS][CTF_MAX_AGENTS]; /* agent-level counters; team / type sums are derived */
    uint8_t visitation[CTF_MAX_AGENTS][CTF_MAX_CELLS]; /* metrics['agent_visitation_maps'][i], u8 wraps */
} ctf_state_view;

typedef struct ctf_env ctf_env; /* opaque; owns all device-side SoA state of n_envs envs */

/* GridworldCtf.__init__ + first reset() (gridworld_ctf.py:19-350, :383-477) for n_envs envs on
 * HIP device `device_id`.  RNG streams start as seeds 0 (see ctf_seed). */
int ctf_create(const ctf_config* cfg, int32_t n_envs, int32_t device_id, ctf_env** out);
void ctf_destroy(ctf_env* env);

int32_t ctf_n_envs(const ctf_env* env);
/* bytes of one env's observation block u8[N][C][G][G], elements of one env's metadata f16[N][M] */
int64_t ctf_obs_bytes_per_env(const ctf_env* env);
int64_t ctf_meta_elems_per_env(const ctf_env* env);

/* Per-env twin MT19937 streams.  Env e behaves as a reference process after
 * `random.seed(py_seeds[e]); np.random.seed(np_seeds[e])` (CPython init_by_array / NumPy legacy
 * init_genrand).  Host arrays of n_envs seeds; np seeds must be < 2^32.
 * (On the device a stream is kept as the ring of its NEXT 624 outputs, so that a step finds its random words in memory
 * and regenerates behind itself; the hand-over functions below convert to and from the standard form, exactly.)
 * In counter mode the seeds are the keys of the env's two counter streams (any 64-bit values), both at word 0. */
int ctf_seed(ctf_env* env, const uint64_t* py_seeds, const uint64_t* np_seeds, void* stream);
/* Exact state hand-over for one env: 624 words + position, i.e. random.getstate()[1] and
 * np.random.get_state()[1:3].  Either pointer may be NULL.  Synchronous. */
int ctf_set_rng_state(ctf_env* env, int32_t env_index, const uint32_t* py_mt625, const uint32_t* np_mt625);
int ctf_get_rng_state(ctf_env* env, int32_t env_index, uint32_t* py_mt625, uint32_t* np_mt625);

/* The same for ALL envs at once, stream-ordered and without a device synchronisation: py_dev / np_dev are DEVICE arrays
 * uint32 [E][625] (624 state words + position per env, either may be NULL).  ctf_get_rng_states returns the standard
 * form.  What the facade's global-RNG contract uses per step
 * (random.getstate() / np.random.get_state() in, the advanced states out) and what a checkpoint of a batch needs. */
int ctf_set_rng_states(ctf_env* env, const uint32_t* py_dev, const uint32_t* np_dev, void* stream);
int ctf_get_rng_states(ctf_env* env, uint32_t* py_dev, uint32_t* np_dev, void* stream);
/* Counter mode only (the four functions above return CTF_E_INVALID there, these two in MT19937 mode): the whole RNG state
 * of an env is how many words it has consumed from each stream.  counters_dev: DEVICE array uint64 [E][2] = (random,
 * np.random).  Stream-ordered.  ctf_set_rng_counters is what a checkpoint restore calls after ctf_seed. */
int ctf_get_rng_counters(ctf_env* env, uint64_t* counters_dev, void* stream);
int ctf_set_rng_counters(ctf_env* env, const uint64_t* counters_dev, void* stream);

/* GridworldCtf.reset() (gridworld_ctf.py:383-477) for the envs whose mask byte is non-zero
 * (NULL = all).  Draws no random numbers and keeps `_arr`, like the reference. */
int ctf_reset(ctf_env* env, const uint8_t* env_mask_dev, void* stream);

/* GridworldCtf.step(actions) (gridworld_ctf.py:849-918) for every env.
 *   actions_dev      int8 [E][N]
 *   rewards_f32_dev  float  [E][N] or NULL   (what ppo.py:107 stores)
 *   rewards_f64_dev  double [E][N] or NULL   (the reference's Python floats, bit-exact)
 *   done_dev         uint8 [E] or NULL */
int ctf_step(ctf_env* env, const int8_t* actions_dev, float* rewards_f32_dev, double* rewards_f64_dev,
             uint8_t* done_dev, uint32_t flags, void* stream);

/* standardise_state(i, reverse_grid) + get_env_metadata(i) for every agent of every env
 * (gridworld_ctf.py:975-1009, :1027-1069).
 *   obs_dev      uint8 [E][N][C][G][G] or NULL
 *   meta_dev     IEEE binary16 bits [E][N][M], M = 2N+6, or NULL
 *   reverse_mask bit i = reverse_grid for agent i; CTF_REVERSE_DEFAULT = (team(i) == 1), the value
 *                every caller in the reference passes (ppo.py:69,87; utils.py:535)
 * One launch: k_observe_tiles (one wave per 8 KiB of the flat buffer) when an env's block is a multiple of 16 bytes and at
 * least 8 KiB and obs_dev is 16-byte aligned, k_observe (one wave per env) otherwise; identical bytes either way.  When the
 * observations of all live handles of this process on the device together exceed 320 MB (more than the caches absorb) the tile
 * kernel's stores carry the nontemporal hint: the observations pass by the caches, which keep the env state for the next step
 * (CTF_OBS_NT=0 / 1 at ctf_create forces plain / hinted stores; other PROCESSES' handles on the device are not seen); smaller
 * batches' observations are left in the memory-side cache for their consumer.
 *
 * PLACEMENT of obs_dev (the caller's buffer, so the caller's business): on MI355X roughly one hipMalloc allocation
 * of > 1 GiB in ten is of a kind this launch streams into at 0.87 of the HBM peak, the others cost it 15-20 % — a property of the
 * allocation's physical backing, independent from one allocation to the next even after a free.  A caller that cares allocates a
 * candidate, times ctf_observe into it against a plain fill of the same bytes (the fill does not depend on the kind: render / fill
 * <= 1.08 is the fast kind, >= 1.15 the slow one), frees it if it is slow and tries again.  The Python facade does exactly that on first use of its
 * observation buffer: knobs placement_tries (default 256 candidates, CTF_PLACEMENT_TRIES), CTF_PLACEMENT_SECONDS (default 3.0: once a
 * buffer of the fast kind is in hand the search for a better one ends after this long; ten seconds while none has turned up) and placement_gib (default 16: the cap
 * on what the search may HOLD — it holds two buffers, the candidate and the best so far; CTF_PLACEMENT_GIB), tune_placement=False
 * to switch it off; VecGridworldCtf.placement reports kind, ratio, candidates tried, bytes held and the time it took. */
#define CTF_REVERSE_DEFAULT 0xFFFFFFFFu
int ctf_observe(ctf_env* env, uint8_t* obs_dev, uint16_t* meta_dev, uint32_t reverse_mask, void* stream);
/* which of the two a ctf_observe into obs_dev launches: 1 = k_observe_tiles, 0 = k_observe (profiling: attributing a measured
 * duration to the right kernel) */
int32_t ctf_observe_kernel(const ctf_env* env, const uint8_t* obs_dev);
/* 1 when that launch stores the observation with the nontemporal hint (the tile kernel while the live handles' observations on
 * the device exceed 320 MB, or CTF_OBS_NT=1), 0 for plain stores */
int32_t ctf_observe_stores_hinted(const ctf_env* env, const uint8_t* obs_dev);

/* The same observation in compact form: the tile planes 1..C-1 of standardise_state are one-hot per cell
 * (plane k+1 = (relabelled grid == TILES_USED[k]), gridworld_ctf.py:990-1001) and plane 0 holds the single
 * own-position bit, so
 *   codes_dev    uint8 [E][N][G][G] or NULL: low 7 bits = index of the plane that is 1 at this cell (0 = none of
 *                1..C-1), bit 7 = plane 0;  obs[e][i][k][r][c] == (k ? (codes & 127) == k : codes >> 7)
 *   meta_dev     as ctf_observe
 *   selfcell_dev uint16 [E][N] or NULL: the cell index (row-major over G x G, after the flip) at which agent i's row has
 *                bit 7 set — redundant with codes_dev, for consumers that treat a team's agents together
 * 1/C of the bytes of ctf_observe; what a GPU policy (include/ctf_policy.h) and a rollout buffer consume. */
int ctf_observe_codes(ctf_env* env, uint8_t* codes_dev, uint16_t* meta_dev, uint16_t* selfcell_dev, uint32_t reverse_mask,
                      void* stream);

/* ctf_step immediately followed by ctf_observe (the rollout inner loop, ppo.py:59-98): the two launches enqueued by one
 * call.  (A single fused launch was built and measured in round 2 — bit-exact but 25 % slower, because a group's step
 * is an ~85 us dependent chain that only the step kernel's "every group in flight at once" shape hides:
 * profiles/r02_fused_step_observe_ablation.md.) */
int ctf_step_observe(ctf_env* env, const int8_t* actions_dev, float* rewards_f32_dev,
                     double* rewards_f64_dev, uint8_t* done_dev, uint8_t* obs_dev, uint16_t* meta_dev,
                     uint32_t reverse_mask, uint32_t flags, void* stream);

/* AGENT_TYPE_ACTION_MASK expanded as agent_network.py:66-75 does: mask_host[i][a] = 1 if action a is
 * legal for agent i (flag 1 => actions 0..4 only).  Host buffer uint8 [N][9]. */
int ctf_action_mask(const ctf_env* env, uint8_t* mask_host);

/* Host views of one env (synchronous; parity tests and the facade's attribute access). */
int ctf_get_state(ctf_env* env, int32_t env_index, ctf_state_view* host_out);
int ctf_set_state(ctf_env* env, int32_t env_index, const ctf_state_view* host_in);

/* ONE env step for a caller that lives in HOST memory: the batch-of-one mode behind the reference's own class API, i.e. what a
 * rollout loop over GridworldCtf does per step (ppo.py:59-98): step(actions) (gridworld_ctf.py:849-918), then standardise_state(i) +
 * get_env_metadata(i) for every agent (:975-1069), with the process-global generators handed in and back (random.getstate() /
 * np.random.get_state(); the reference draws from them inside step).  For a handle of n_envs == 1 (CTF_E_INVALID otherwise).
 * Everything — the hand-over of both generator states, the step, the render, the state read-back — is enqueued on `stream` as kernels
 * that read their inputs from, and write their outputs into, ONE pinned, device-mapped host block the handle owns (no copy operations),
 * and the call waits for the stream once: round 5 measured 408 us per env step for the same work through ctf_set_rng_states + ctf_step
 * + ctf_get_rng_states + ctf_observe + ctf_get_state + ctf_status with a wait behind each.
 * All pointers are HOST pointers; any OUT pointer may be NULL (not wanted).
 *   actions_host    int8 [N], or NULL: no step is made (state, observation and generator states are still returned: reset())
 *   py_mt625_in / np_mt625_in   uint32 [625] (624 words + position) to install BEFORE the step, or NULL (keep the device's)
 *   reverse_mask, flags         as ctf_observe / ctf_step
 *   rewards_host    double [N] (the reference's Python floats, bit-exact)      done_host    int32
 *   status_host     the sticky CTF_ST_* bits raised since the last read (cleared, like ctf_status)
 *   view_host       the env's state after the step (ctf_get_state's view)
 *   py_mt625_out / np_mt625_out uint32 [625]: the generators AFTER the step, standard form
 *   obs_host        uint8 [N][C][G][G]      meta_host   IEEE binary16 bits [N][M] */
int ctf_host_step(ctf_env* env, const int8_t* actions_host, const uint32_t* py_mt625_in, const uint32_t* np_mt625_in,
                  uint32_t reverse_mask, uint32_t flags, double* rewards_host, int32_t* done_host, uint32_t* status_host,
                  ctf_state_view* view_host, uint32_t* py_mt625_out, uint32_t* np_mt625_out, uint8_t* obs_host, uint16_t* meta_host,
                  void* stream);

/* Bulk export of the counters the duel / evaluation harness reads (utils.py:557-571, metrics_logger.py:137-159):
 *   metrics_dev   int32 [E][13][N] agent-level counters (CTF_M_* order) or NULL (zeros when log_metrics == 0)
 *   captures_dev  int32 [E][2]     metrics['team_flag_captures'] or NULL
 *   steps_dev     int32 [E]        env_step_count or NULL */
int ctf_export_counters(ctf_env* env, int32_t* metrics_dev, int32_t* captures_dev, int32_t* steps_dev, void* stream);

/* Sticky status bits raised by any env since the last call (synchronises `stream`, clears them). */
int ctf_status(ctf_env* env, uint32_t* out_bits, void* stream);

/* Synthetic workload helper for bench/tests: actions_dev[e][i] uniform on 0..8 from Philox4x32-10,
 * key = (seed lo, seed hi), counter = (global env index = env_offset + e, step, 0, 0). */
int ctf_random_actions(ctf_env* env, int8_t* actions_dev, uint64_t seed, uint32_t step,
                       uint32_t env_offset, void* stream);

const char* ctf_last_error(void); /* thread-local */
int32_t ctf_abi_version(void);
/* sizeof(ctf_config) / sizeof(ctf_state_view) as compiled, so a binding can verify its struct mirror */
int32_t ctf_sizeof_config(void);
int32_t ctf_sizeof_state_view(void);

#ifdef __cplusplus
}
#endif
#endif /* CTF_ENV_H */
