/* ctf_policy.h — C ABI of the policy network's convolutional front on MI355X (same shared library as ctf_env.h).
 *
 * Replaces, for inference over whole batches of agents, the first half of the reference's
 *   Agent.forward(x, x2)                      agent_network.py:30-36
 *     x = tanh(conv1(x)); x = tanh(conv2(x))  conv1 = Conv2d(C, 16, 3), conv2 = Conv2d(16, 32, 3)   (:13-14)
 *     x = x.view(-1, 32 * (G-4)^2); x = concat((x, x2), dim=1)
 * and hands the result to fc1 as a bf16 matrix.  fc1 / fc2 / the heads are plain GEMMs and stay with the BLAS library
 * (host side: marl-ctf-development_amd/policy_native.py).
 *
 * Input is the env's compact observation (ctf_observe_codes, include/ctf_env.h): one byte per (agent, cell) instead of C
 * one-hot bytes.  All pointers except agent_sel are device pointers owned by the caller; the call only enqueues on
 * `stream`.  Returns 0, or -1 with ctf_policy_last_error().
 */
#ifndef CTF_POLICY_H
#define CTF_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Row length (in bf16 elements) of the activation matrix: 32 * PP + meta_len rounded up to a multiple of 64, where
 * PP = (G-4)^2 rounded up to a multiple of 32 (whole 128-byte lines per store instruction and per row). */
int32_t ctf_policy_act_stride(int32_t grid_size, int32_t meta_len);

/* act[k * n_envs + e][:] = features of agent agent_sel[k] of env e  (agent-major rows, the order
 * PPOTrainer.get_single_rollout interleaves agents within a step, ppo.py:74-84).
 *
 *   codes_dev        uint8  [n_envs][n_agents][G][G]      from ctf_observe_codes
 *   meta_dev         binary16 bits [n_envs][n_agents][meta_len]
 *   agent_sel        HOST int32 [n_sel]: which agents this network plays (e.g. one team), n_sel <= 16
 *   conv1_frag_dev   bf16 [5][64][8]: conv1.weight * 2 log2(e) in MFMA 16x16x32 A-operand order:
 *                    [s][lane][j] = W1[out = lane & 15][in = 8 * ((lane >> 4) & 1) + j][tap = 2 s + (lane >> 5)], 0 where
 *                    in >= C or tap == 9  (tap = 3 * ky + kx)
 *   conv1_bias_dev   float [16]: conv1.bias * 2 log2(e)
 *   conv2_frag_dev   bf16 [9][64][8]: [tap][lane][j] = W2[out = lane & 31][in = 8 * (lane >> 5) + j][tap] * 2 log2(e)
 *   conv2_bias_dev   float [32]: conv2.bias * 2 log2(e)
 *   act_dev          bf16 [n_sel * n_envs][ctf_policy_act_stride()], 16-byte aligned.  Column of conv2 output channel c
 *                    at position p (row-major over (G-4)^2): ((c / 4) * PP + p) * 4 + c % 4; columns of positions
 *                    (G-4)^2 .. PP-1 hold finite don't-care values (give them zero weight); at 32 * PP the meta_len
 *                    metadata values; at 32 * PP + meta_len the constant 1.0 (a caller may keep fc1's bias in that column of
 *                    its weight, else zero weight); then zeros.  (fc1.weight's columns are permuted to this order once.)
 *   shared_view_selfcell_dev   NULL, or uint16 [n_envs][n_agents] from ctf_observe_codes (the cell of every agent's bit 7)
 *                    together with the caller's guarantee that the selected agents see the same tile planes — same team
 *                    and same reverse flag (standardise_state relabels by team, gridworld_ctf.py:981-988), so that their
 *                    code rows differ only in bit 7.  The convolutions are then evaluated once per env and patched per
 *                    agent (3 x 3 conv1 / 5 x 5 conv2 outputs around the own cell); results are bit-identical to the NULL
 *                    path.  Honoured for n_sel <= 4 and G in {11, 15}, ignored otherwise.
 */
int ctf_policy_features(const uint8_t* codes_dev, const uint16_t* meta_dev, int32_t n_envs, int32_t n_agents,
                        int32_t grid_size, int32_t meta_len, const int32_t* agent_sel, int32_t n_sel,
                        const void* conv1_frag_dev, const float* conv1_bias_dev, const void* conv2_frag_dev,
                        const float* conv2_bias_dev, uint16_t* act_dev, const uint16_t* shared_view_selfcell_dev,
                        int32_t device_id, void* stream);

/* The same front as the FORWARD of a training step (the learner's re-evaluation of a minibatch, ppo.py:199-203 -> agent_network.py:30-36):
 * one sample per row of codes_dev, the activation row as above (bit-identical to ctf_policy_features on the same codes), plus what a
 * backward pass needs and would otherwise have to recompute, both channels-last so that a library's weight- / data-gradient kernels take
 * them as they are:
 *   codes_dev   uint8 [n_samples][G][G]           meta_dev   binary16 bits [n_samples][meta_len]
 *   act_dev     bf16 [n_samples][ctf_policy_act_stride()]
 *   h0_dev      bf16 [n_samples][G*G][16]         the one-hot input image (planes C..15 zero); NULL: not written (ctf_policy_front_wgrad
 *                                                 builds it from the codes itself)
 *   h1_dev      bf16 [n_samples][(G-2)^2][16]     tanh(conv1)
 * all 16-byte aligned; grid_size 11 or 15. */
int ctf_policy_features_train(const uint8_t* codes_dev, const uint16_t* meta_dev, int64_t n_samples, int32_t grid_size,
                              int32_t meta_len, const void* conv1_frag_dev, const float* conv1_bias_dev,
                              const void* conv2_frag_dev, const float* conv2_bias_dev, uint16_t* act_dev, uint16_t* h0_dev,
                              uint16_t* h1_dev, int32_t device_id, void* stream);

/* The whole data path of that backward in one launch (grid_size 11 / 15): dz2 = d_act * (1 - act^2) out channels-last, conv2's data
 * gradient by MFMA on it, dz1 = that * (1 - h1^2) out channels-last, both bias gradients — what is left to a library are the two
 * weight gradients (of conv2 from h1 and dz2, of conv1 from h0 and dz1).
 *   d_act_dev, act_dev   bf16 [n_samples][ctf_policy_act_stride()]
 *   h1_dev               bf16 [n_samples][(G-2)^2][16] as ctf_policy_features_train wrote it
 *   conv2_t_frag_dev     bf16 [9][64][8]: [tap][lane][j] = conv2.weight[out = 8 * (lane >> 4) + j][in = lane & 15][tap], UNscaled
 *   dz2_dev              bf16 [n_samples][(G-4)^2][32]        dz1_dev   bf16 [n_samples][(G-2)^2][16]
 *   bias2_grad_dev / bias1_grad_dev   NULL, or float [32] / [16]: += per-channel sums (the caller zeroes them) */
int ctf_policy_front_dgrad(const uint16_t* d_act_dev, const uint16_t* act_dev, const uint16_t* h1_dev, const void* conv2_t_frag_dev,
                           int64_t n_samples, int32_t grid_size, int32_t meta_len, uint16_t* dz2_dev, uint16_t* dz1_dev,
                           float* bias2_grad_dev, float* bias1_grad_dev, int32_t device_id, void* stream);

/* ... and the two weight gradients, contraction over positions on the matrix cores (two launches):
 *   dw2[o][i][tap] += sum over samples and conv2 output positions of dz2[o][y][x] * h1[i][y + dy][x + dx]          float [32][16][9]
 *   dw1[o][c][tap] += sum over samples and conv1 output positions of dz1[o][y][x] * onehot(codes)[c][y + dy][x + dx]  float [16][16][9]
 * (conv weight layout [out][in][ky][kx], in-channels padded to 16; the caller zeroes both).  dz2 / dz1 as ctf_policy_front_dgrad wrote
 * them, h1 as ctf_policy_features_train did, codes_dev uint8 [n_samples][G][G]. */
int ctf_policy_front_wgrad(const uint16_t* dz2_dev, const uint16_t* h1_dev, const uint16_t* dz1_dev, const uint8_t* codes_dev,
                           int64_t n_samples, int32_t grid_size, float* dw2_dev, float* dw1_dev, int32_t device_id, void* stream);

/* ctf_policy_front_dgrad + both weight gradients as ONE call (round 4): conv2's data and weight gradient in one pass — dz2 stays on the
 * CU, never written or re-read — then conv1's weight gradient from the dz1 that pass wrote (a scratch of the caller's: bf16
 * [n_samples][(G-2)^2][16], 16-byte aligned) and the code bytes.  The same sums as the separate calls; dw2 / dw1 / the bias gradients are
 * added to (the caller zeroes them). */
int ctf_policy_front_backward(const uint16_t* d_act_dev, const uint16_t* act_dev, const uint16_t* h1_dev, const uint8_t* codes_dev,
                              const void* conv2_t_frag_dev, int64_t n_samples, int32_t grid_size, int32_t meta_len, uint16_t* dz1_dev,
                              float* dw2_dev, float* dw1_dev, float* bias2_grad_dev, float* bias1_grad_dev, int32_t device_id,
                              void* stream);

/* The weight and bias gradients of the small dense layers in the learner's backward (fc2: Linear(256, 128), agent_network.py:16; the two
 * heads as one Linear(128, 16): rows 0..A-1 the action head, row A the value head, the rest zero, :17-18):
 *     dw[n][k] += sum over samples of dy[m][n] * x[m][k]        db[n] += sum over samples of dy[m][n]      (the caller zeroes dw / db)
 *   dy_dev bf16 [n_samples][n_out], x_dev bf16 [n_samples][n_in], both 16-byte aligned; (n_out, n_in) = (128, 256) or (16, 128);
 *   dw_dev float [n_out][n_in]; db_dev float [n_out] or NULL.
 * One pass over dy and x (HBM-bound) in place of the library's GEMM with a 128 x 256 output and K = n_samples plus a column reduction. */
int ctf_policy_linear_wgrad(const uint16_t* dy_dev, const uint16_t* x_dev, int64_t n_samples, int32_t n_out, int32_t n_in, float* dw_dev,
                            float* db_dev, int32_t device_id, void* stream);

/* The rest of Agent.get_action_and_value (agent_network.py:37-40, 63-81) in one kernel:
 *   x = tanh(fc1 out); x = tanh(fc2(x)); value = value_head(x); logits = action_head(x)
 *   logits += (mask - 1) * 1e9 with mask = [1]*5 + [0]*(A-5) where the decision is 1, all ones otherwise
 *   action ~ Categorical(logits) (or the given action); log_prob(action); entropy
 *
 *   fc1_out_dev        bf16 [n_samples][256]: fc1's output INCLUDING bias, times 2 log2(e) (scale fc1's weight and bias
 *                      once; the BLAS GEMM then produces this directly), 16-byte aligned
 *   fc2_frag_dev       bf16 [4][16][64][8]: [w][s][lane][j] = fc2.weight[32 w + (lane & 31)][16 s + 8 (lane >> 5) + j] * 2 log2(e)
 *   fc2_bias_dev       float [128]: fc2.bias * 2 log2(e)
 *   head_frag_dev      bf16 [4][64][8]: [s][lane][j] = H[lane & 15][32 s + 8 (lane >> 4) + j], H rows 0..A-1 =
 *                      action_head.weight, row A = value_head.weight, other rows 0
 *   head_bias_dev      float [16] in the same row order
 *   mask_decision_dev  float [n_samples] (1 => only actions 0..4 legal) or NULL (no masking)
 *   given_action_dev   int32 [n_samples] to evaluate instead of sampling, or NULL
 *   seed, offset       Philox4x32-10 key / counter words of the sampler: the uniform of sample i is
 *                      philox(counter = (i, offset), key = seed).x >> 8 scaled to [0, 1); the action is the inverse CDF
 *   outputs            action int32, logprob / entropy / value float [n_samples]; logits float [n_samples][A] or NULL
 */
int ctf_policy_head(const uint16_t* fc1_out_dev, int64_t n_samples, const void* fc2_frag_dev, const float* fc2_bias_dev,
                    const void* head_frag_dev, const float* head_bias_dev, const float* mask_decision_dev,
                    const int32_t* given_action_dev, int32_t n_actions, uint64_t seed, uint64_t offset, int32_t* action_dev,
                    float* logprob_dev, float* entropy_dev, float* value_dev, float* logits_dev, int32_t device_id,
                    void* stream);

/* ---- fc1 without the per-agent activation matrix (round 4; csrc/ctf_policy_fact.hip) -----------------------------------------------
 * For agents that share a view (same team and reverse flag: their code rows differ only in bit 7), fc1 — linear in tanh(conv2) — splits
 * into a part per (env, view) and a correction per agent (agent_network.py:15,30-37 computes the same sum per agent in one product):
 *     fc1(h2_a ++ meta_a) = W_flat . h2_view + W[:, patch(a)] . (h2_a - h2_view)[patch(a)] + W_meta . meta_a + b
 * with patch(a) = the 5 x 5 conv2 positions x 32 channels an own-position bit reaches.  Four calls per step and view group:
 *   1. ctf_policy_fact_bucket     agents -> slots, bucketed by own cell (a tile of 128 slots shares one 800-column slice of W)
 *   2. ctf_policy_features_fact   conv front: ONE row of the view matrix per env + one patch row per agent, written into its slot
 *   3. (the caller's BLAS)        yview[E][256] = view[E][KV] x W_flat^T   (fp32 out, no bias)
 *   4. ctf_policy_fc1_patch       y1[agent] = bf16(W_patch(cell) . patch row + yview[env] + b): the input of ctf_policy_head
 * Activation traffic per 65 536-env arena step: 2 x (1.07 GB view + 0.46 GB patch rows) written and read once, against 2 x 2.2 GB.
 * grid_size 11 or 15, n_sel <= 4, meta_len <= 32.  Every pointer but agent_sel is a device pointer; calls only enqueue on `stream`. */

/* Row lengths (bf16 elements): the view matrix (32 * PP, PP = (G-4)^2 rounded up to 32; column of channel c, position p as in
 * ctf_policy_features: ((c / 4) * PP + p) * 4 + c % 4) and the patch rows (25 * 32 deltas, k = 32 * (5 dy + dx) + c; then meta_len
 * metadata values; zero padding to a multiple of 64).  ctf_policy_fact_max_tiles: tiles of 128 slots a launch can need at most. */
int32_t ctf_policy_fact_view_stride(int32_t grid_size);
int32_t ctf_policy_fact_row_stride(int32_t meta_len);
int32_t ctf_policy_fact_max_tiles(int32_t n_envs, int32_t n_sel, int32_t grid_size);

/*   selfcell_dev     uint16 [n_envs][n_agents] from ctf_observe_codes
 *   work_dev         int32 [576 + max_tiles]: scratch of the bucketing that ctf_policy_fc1_patch reads (tile count, tile -> own cell);
 *                    ZERO it once before the first call (every call leaves its histogram part zeroed for the next)
 *   slot_of_dev      int32 [n_sel * n_envs] out: row k * n_envs + e -> slot
 *   row_of_slot_dev  int32 [max_tiles * 128] out: slot -> row, -1 in the padding of a bucket's last tile */
int ctf_policy_fact_bucket(const uint16_t* selfcell_dev, int32_t n_envs, int32_t n_agents, int32_t grid_size, const int32_t* agent_sel,
                           int32_t n_sel, int32_t* work_dev, int32_t* slot_of_dev, int32_t* row_of_slot_dev, int32_t device_id,
                           void* stream);

/* codes / meta / fragments / biases as for ctf_policy_features (same arithmetic, bit for bit, up to tanh(conv2) in bf16).
 *   view_dev   bf16 [n_envs][ctf_policy_fact_view_stride()]: tanh(conv2) of the view WITHOUT any own-position bit
 *   prow_dev   bf16 [max_tiles * 128][ctf_policy_fact_row_stride()]: row slot_of[k][e] = bf16(bf16(h2 of agent k) - bf16(h2 of the
 *              view)) on the agent's 25 patch positions (0 where a position lies outside the image), then its metadata as bf16 */
int ctf_policy_features_fact(const uint8_t* codes_dev, const uint16_t* meta_dev, const uint16_t* selfcell_dev, int32_t n_envs,
                             int32_t n_agents, int32_t grid_size, int32_t meta_len, const int32_t* agent_sel, int32_t n_sel,
                             const void* conv1_frag_dev, const float* conv1_bias_dev, const void* conv2_frag_dev,
                             const float* conv2_bias_dev, const int32_t* slot_of_dev, uint16_t* view_dev, uint16_t* prow_dev,
                             int32_t device_id, void* stream);

/* yview = view x W_flat^T as a stream over the view matrix (a block: 128 rows x all 256 outputs, W resident in L2): what the library's
 * GEMM computes for the factored path (torch.mm(view, W_flat^T, out_dtype=float32)), same operands, float32 accumulation in k order.
 *   view_dev     bf16 [n_rows][kv] from ctf_policy_features_fact (kv = ctf_policy_fact_view_stride(): 2048 or 4096; a multiple of 64 is required)
 *   w_rows_dev   bf16 [256][kv]: fc1.weight's conv columns * 2 log2(e) in the view's column order (zero on the row's padding)
 *   yview_dev    float [n_rows][256] out */
int ctf_policy_view_gemm(const uint16_t* view_dev, const uint16_t* w_rows_dev, int32_t n_rows, int32_t kv, float* yview_dev,
                         int32_t device_id, void* stream);

/*   yview_dev        float [n_envs][256]: view x W_flat^T with W_flat = fc1.weight's conv columns * 2 log2(e) in the view's column order
 *   patch_frag_dev   bf16 [(G-4)^2 + 2][2][8][64][8]: MFMA 32x32x16 A-operand fragments of fc1.weight * 2 log2(e):
 *                    [p][s][t][lane][j] = W[out = 32 t + (lane & 31)][column of channel 16 s + 8 (lane >> 5) + j at position p];
 *                    block p = (G-4)^2: zeros; block (G-4)^2 + 1: the metadata columns (k = 16 s + 8 (lane >> 5) + j < meta_len)
 *   fc1_bias_dev     float [256]: fc1.bias * 2 log2(e)
 *   y1_dev           bf16 [n_sel * n_envs][256] out: fc1's pre-activation * 2 log2(e), row k * n_envs + e — ctf_policy_head's input */
int ctf_policy_fc1_patch(const uint16_t* prow_dev, const int32_t* row_of_slot_dev, const int32_t* work_dev, const float* yview_dev,
                         const void* patch_frag_dev, const float* fc1_bias_dev, int32_t n_envs, int32_t n_sel, int32_t grid_size,
                         int32_t meta_len, uint16_t* y1_dev, int32_t device_id, void* stream);

/* ctf_policy_fc1_patch with ctf_policy_head fused behind it: a tile of 128 slots is a tile of 128 samples of the network's tail, so
 * fc1's pre-activation never goes to HBM — it is rounded to bf16 exactly as ctf_policy_fc1_patch stores it, and the tail runs on it in
 * LDS.  Arguments: ctf_policy_fc1_patch's (without y1_dev) then ctf_policy_head's (without fc1_out_dev / n_samples; the sample index of
 * every output, of mask_decision_dev / given_action_dev and of the Philox counter is the row k * n_envs + e).  Results are bit-identical
 * to the two separate calls. */
int ctf_policy_fc1_patch_head(const uint16_t* prow_dev, const int32_t* row_of_slot_dev, const int32_t* work_dev, const float* yview_dev,
                              const void* patch_frag_dev, const float* fc1_bias_dev, int32_t n_envs, int32_t n_sel, int32_t grid_size,
                              int32_t meta_len, const void* fc2_frag_dev, const float* fc2_bias_dev, const void* head_frag_dev,
                              const float* head_bias_dev, const float* mask_decision_dev, const int32_t* given_action_dev,
                              int32_t n_actions, uint64_t seed, uint64_t offset, int32_t* action_dev, float* logprob_dev,
                              float* entropy_dev, float* value_dev, float* logits_dev, int32_t device_id, void* stream);

/* fc1's data gradient in the learner's backward (ppo.py:231-233 -> autograd through agent_network.py:16): d_act = dy x W, what
 * torch.mm(dy, W) computes (float32 accumulation, one rounding to bf16), as an HBM-write-bound stream.
 *   dy_dev     bf16 [n_samples][256]: the gradient at fc1's output; n_samples a multiple of 32
 *   wt_dev     bf16 [kp][256]: W^T, W = the [256][kp] matrix fc1's forward multiplied by (kp a multiple of 64)
 *   d_act_dev  bf16 [n_samples][kp] out */
int ctf_policy_fc1_dgrad(const uint16_t* dy_dev, const uint16_t* wt_dev, int32_t n_samples, int32_t kp, uint16_t* d_act_dev,
                         int32_t device_id, void* stream);

/* The rollout collector's per-step bookkeeping as one launch — what PPOTrainer.get_single_rollout stores per trained agent and the joint
 * action it hands to env.step (ppo.py:74-93): for trained slot k (agent trained_sel[k]) and env e, row k * n_envs + e of the outputs gets
 * the agent's code bytes and metadata (binary16 -> float32), its action (int32 -> float32), log-prob and value; env_actions_out[e][n] gets
 * agent n's action — from act_trained_dev / act_other_dev, rows slot * n_envs + e — mapped through reversed_action_lut[9] where bit n of
 * team1_mask is set (REVERSED_ACTION_MAP, gridworld_ctf.py:147-196; the stored action is the un-mapped one, ppo.py:77).  trained_sel /
 * other_sel / reversed_action_lut are HOST arrays; every agent is in exactly one of the two lists. */
int ctf_rollout_store_step(const uint8_t* codes_dev, const uint16_t* meta_dev, int32_t n_envs, int32_t n_agents, int32_t cells,
                           int32_t meta_len, const int32_t* trained_sel, int32_t n_trained, const int32_t* other_sel, int32_t n_other,
                           const int32_t* act_trained_dev, const float* logprob_dev, const float* value_dev, const int32_t* act_other_dev,
                           const uint8_t* reversed_action_lut, uint32_t team1_mask, uint8_t* grid_codes_out, float* metadata_out,
                           float* actions_out, float* logprobs_out, float* values_out, int8_t* env_actions_out, int32_t device_id,
                           void* stream);

/* DETERMINISTIC MODE of the training kernels (the reference's update is deterministic under its seeds, ppo.py:174-242).  The weight / bias
 * gradient kernels — ctf_policy_front_backward, ctf_policy_front_dgrad (bias gradients), ctf_policy_front_wgrad, ctf_policy_linear_wgrad —
 * normally end in one float atomicAdd per element and block on the gradient, whose order of arrival differs from run to run.  With a
 * workspace registered for the device (float32, 16-byte aligned, the caller's, alive until it is unregistered) every block stores its
 * partial sums in its own slice of the workspace instead, and a second launch adds the slices IN BLOCK ORDER: two identical calls give
 * bit-identical gradients.  The launches of one device that use the workspace must be stream-ordered (one learner at a time).
 * 24 M floats (96 MB) cover every launch of the 8_arena network on a 256-CU device; a launch that needs more fails with a message.
 * workspace_dev = NULL switches the mode off.  ctf_policy_deterministic_workspace: floats registered (0 = off). */
int ctf_policy_set_deterministic(int32_t device_id, float* workspace_dev, int64_t workspace_floats);
int64_t ctf_policy_deterministic_workspace(int32_t device_id);

const char* ctf_policy_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
