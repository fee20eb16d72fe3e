"""The policy / value network at the env's speed: the reference ``Agent`` (agent_network.py:5-81) for whole batches of
agents, fed by the compact observation and run by a hand-written MFMA kernel + bf16 GEMMs.

``CtfPolicyNative`` keeps ``CtfPolicy``'s parameters (so a reference ``state_dict`` loads unchanged, and training
through the stock ``forward`` keeps working) and adds an inference path that never materialises the one-hot planes:

    codes, meta = vec.observe_codes()                              # 1 byte per (agent, cell)
    action, logprob, entropy, value = net.act_from_codes(codes, meta, agent_idx, mask_decision)

* conv1 -> tanh -> conv2 -> tanh -> flatten ++ metadata: ``ctf_policy_features`` (include/ctf_policy.h,
  csrc/ctf_policy.hip), one wave per agent, activations in LDS, bf16 MFMA with float32 accumulation;
* fc1 -> tanh -> fc2 -> tanh: bf16 GEMMs (hipBLASLt through torch) on the kernel's activation matrix — fc1's weight
  columns are permuted once to the order the kernel writes;
* heads, mask rule ``logits + (mask - 1) * 1e9`` (agent_network.py:66-75) and sampling in float32.

Numerics: bf16 operands, float32 accumulation; against the float32 reference network the logits / values differ by a few
1e-2 (tests/test_gpu_policy_native.py states the tolerance).  ``prepare()`` must be called again after the parameters
change (it is called lazily on first use).
"""
import ctypes as C
import math

import numpy as np
import torch
from torch.distributions.categorical import Categorical

from . import _abi
from .policy import CtfPolicy

_TWO_LOG2E = 2.0 / math.log(2.0)


def conv_fragments(conv1_w, conv1_b, conv2_w, conv2_b):
    """float32 conv parameters -> the kernel's operands (include/ctf_policy.h): bf16 MFMA A-fragments of both
    convolutions and float32 biases, all scaled by 2 log2(e) (the kernel's tanh works on base-2 exponents)."""
    w1 = np.asarray(conv1_w, np.float64) * _TWO_LOG2E  # [16, C, 3, 3]
    w2 = np.asarray(conv2_w, np.float64) * _TWO_LOG2E  # [32, 16, 3, 3]
    c_in = w1.shape[1]
    if w1.shape[0] != 16 or c_in > 16 or w2.shape[:2] != (32, 16):
        raise ValueError("the native front is built for Conv2d(C<=16, 16, 3) -> Conv2d(16, 32, 3)")
    w1t = np.zeros((16, 16, 10), np.float64)
    w1t[:, :c_in, :9] = w1.reshape(16, c_in, 9)
    lane = np.arange(64)
    j = np.arange(8)
    f1 = np.zeros((5, 64, 8), np.float32)
    for s in range(5):
        tap = 2 * s + (lane >> 5)
        cin = 8 * ((lane >> 4) & 1)[:, None] + j[None, :]
        f1[s] = w1t[(lane & 15)[:, None], cin, tap[:, None]]
    w2t = w2.reshape(32, 16, 9)
    f2 = np.zeros((9, 64, 8), np.float32)
    for tap in range(9):
        f2[tap] = w2t[(lane & 31)[:, None], 8 * (lane >> 5)[:, None] + j[None, :], tap]
    b1 = (np.asarray(conv1_b, np.float64) * _TWO_LOG2E).astype(np.float32)
    b2 = (np.asarray(conv2_b, np.float64) * _TWO_LOG2E).astype(np.float32)
    return f1, b1, f2, b2


def act_column_order(grid_size, meta_len):
    """new column -> reference column of fc1's input (-1 = zero padding): the kernel writes conv2 channel c, position p at
    ((c // 4) * P2 + p) * 4 + c % 4, the reference's flatten at c * P2 + p; metadata follows in both."""
    p2 = (grid_size - 4) ** 2
    kp = (32 * p2 + meta_len + 31) // 32 * 32
    src = np.full(kp, -1, np.int64)
    c, p = np.meshgrid(np.arange(32), np.arange(p2), indexing="ij")
    src[((c // 4) * p2 + p) * 4 + c % 4] = c * p2 + p
    src[32 * p2:32 * p2 + meta_len] = 32 * p2 + np.arange(meta_len)
    return src


class CtfPolicyNative(CtfPolicy):
    def __init__(self, n_actions, n_channels, grid_size, metadata_size):
        super().__init__(n_actions, n_channels, grid_size, metadata_size, compute_dtype=torch.bfloat16)
        self.grid_size, self.metadata_size, self.n_channels = grid_size, metadata_size, n_channels
        self._prep = None

    # -- weights in the kernel's / the GEMMs' layouts ----------------------------------------------
    def prepare(self):
        dev = self.conv1.weight.device
        if dev.type != "cuda":
            raise _abi.CtfLibraryError("CtfPolicyNative runs on a HIP device only (there is no CPU fallback)")
        lib = _abi.load_library()
        with torch.no_grad():
            f1, b1, f2, b2 = conv_fragments(self.conv1.weight.float().cpu().numpy(), self.conv1.bias.float().cpu().numpy(),
                                            self.conv2.weight.float().cpu().numpy(), self.conv2.bias.float().cpu().numpy())
            bf = torch.bfloat16
            order = act_column_order(self.grid_size, self.metadata_size)
            w = self.fc1.weight.float()
            fc1 = torch.zeros((w.shape[0], len(order)), dtype=torch.float32, device=dev)
            keep = torch.from_numpy(order >= 0).to(dev)
            fc1[:, keep] = w[:, torch.from_numpy(order[order >= 0]).to(dev)]
            self._prep = dict(
                lib=lib, kp=len(order),
                f1=torch.from_numpy(f1).to(dev).to(bf).contiguous(), b1=torch.from_numpy(b1).to(dev),
                f2=torch.from_numpy(f2).to(dev).to(bf).contiguous(), b2=torch.from_numpy(b2).to(dev),
                fc1_w=fc1.to(bf).contiguous(), fc1_b=self.fc1.bias.to(bf),
                fc2_w=self.fc2.weight.to(bf).contiguous(), fc2_b=self.fc2.bias.to(bf),
                head_w=torch.cat((self.action_head.weight, self.value_head.weight), dim=0).float().t().contiguous(),
                head_b=torch.cat((self.action_head.bias, self.value_head.bias), dim=0).float(),
            )
            assert lib.ctf_policy_act_stride(self.grid_size, self.metadata_size) == len(order)
        return self

    def _ready(self):
        if self._prep is None:
            self.prepare()
        return self._prep

    # -- inference from the compact observation ----------------------------------------------------
    def features_from_codes(self, codes, meta, agent_idx, out=None):
        """codes uint8 [E, N, G, G], meta float16 [E, N, M], agent_idx: the agents this network plays ->
        bf16 [len(agent_idx) * E, Kp] (row k * E + e = agent agent_idx[k] of env e)."""
        p = self._ready()
        E, N = int(codes.shape[0]), int(codes.shape[1])
        if not (codes.is_cuda and codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape[2:]) == (self.grid_size,) * 2):
            raise ValueError("codes must be a contiguous uint8 CUDA tensor [E, N, G, G]")
        if not (meta.is_cuda and meta.dtype == torch.float16 and meta.is_contiguous() and tuple(meta.shape) == (E, N, self.metadata_size)):
            raise ValueError("meta must be a contiguous float16 CUDA tensor [E, N, M]")
        sel = [int(i) for i in (agent_idx.tolist() if hasattr(agent_idx, "tolist") else agent_idx)]
        if out is None:
            out = torch.empty((len(sel) * E, p["kp"]), dtype=torch.bfloat16, device=codes.device)
        elif not (out.dtype == torch.bfloat16 and out.is_contiguous() and tuple(out.shape) == (len(sel) * E, p["kp"])):
            raise ValueError("out must be a contiguous bfloat16 tensor [len(agent_idx) * E, Kp]")
        sel_arr = (C.c_int32 * len(sel))(*sel)
        rc = p["lib"].ctf_policy_features(
            C.c_void_p(codes.data_ptr()), C.c_void_p(meta.data_ptr()), E, N, self.grid_size, self.metadata_size, sel_arr, len(sel),
            C.c_void_p(p["f1"].data_ptr()), C.c_void_p(p["b1"].data_ptr()), C.c_void_p(p["f2"].data_ptr()),
            C.c_void_p(p["b2"].data_ptr()), C.c_void_p(out.data_ptr()), codes.device.index,
            C.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream))
        if rc != 0:
            raise _abi.CtfLibraryError("ctf_policy_features: " + (p["lib"].ctf_policy_last_error() or b"").decode())
        return out

    def trunk_from_codes(self, codes, meta, agent_idx):
        """-> (value [B, 1], logits [B, A]) float32, B = len(agent_idx) * E, agent-major."""
        p = self._ready()
        x = self.features_from_codes(codes, meta, agent_idx)
        x = torch.tanh_(torch.nn.functional.linear(x, p["fc1_w"], p["fc1_b"]))
        x = torch.tanh_(torch.nn.functional.linear(x, p["fc2_w"], p["fc2_b"]))
        y = torch.addmm(p["head_b"], x.float(), p["head_w"])
        return y[:, self.n_actions:], y[:, :self.n_actions]

    def act_from_codes(self, codes, meta, agent_idx, masking_decision_tensor, action=None):
        """get_action_and_value (agent_network.py:63-81) for agents ``agent_idx`` of every env, from the compact observation."""
        value, logits = self.trunk_from_codes(codes, meta, agent_idx)
        decision = masking_decision_tensor.reshape(-1, 1).to(logits.dtype)
        mask = torch.where(decision == 1, self.mask_5.unsqueeze(0), torch.ones_like(logits))
        logits = logits + (mask - 1.0) * 1e9
        dist = Categorical(logits=logits)
        if action is None:
            action = dist.sample()
        return action, dist.log_prob(action), dist.entropy(), value
