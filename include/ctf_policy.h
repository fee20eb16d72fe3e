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
 *                    metadata values; then don't-care padding.  (fc1.weight's columns are permuted to this order once.)
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

const char* ctf_policy_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
