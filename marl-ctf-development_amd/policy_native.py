"""The policy / value network at the env's speed: the reference ``Agent`` (agent_network.py:5-81) for whole batches of
agents, fed by the compact observation and run by a hand-written MFMA kernel + bf16 GEMMs.

``CtfPolicyNative`` keeps ``CtfPolicy``'s parameters (so a reference ``state_dict`` loads unchanged, and training
through the stock ``forward`` keeps working) and adds an inference path that never materialises the one-hot planes:

    codes, meta = vec.observe_codes()                              # 1 byte per (agent, cell)
    action, logprob, entropy, value = net.act_from_codes(codes, meta, agent_idx, mask_decision)

* conv1 -> tanh -> conv2 -> tanh -> flatten ++ metadata: ``ctf_policy_features`` (include/ctf_policy.h,
  csrc/ctf_policy.hip), one wave per agent, activations in LDS, bf16 MFMA with float32 accumulation;
* fc1: one bf16 GEMM (hipBLASLt through torch) on the kernel's activation matrix — fc1's weight columns are permuted once
  to the order the kernel writes;
* tanh -> fc2 -> tanh -> heads -> mask rule ``logits + (mask - 1) * 1e9`` (agent_network.py:66-75) -> sampling, log-prob,
  entropy: ``ctf_policy_head``, one fused MFMA kernel (bf16 operands, float32 accumulation and distribution math).

Numerics: bf16 operands, float32 accumulation; against the float32 reference network the logits / values differ by a few
1e-2 (tests/test_gpu_policy_native.py states the tolerance).  The kernel-side copies of the weights are rebuilt by
``prepare()``; it runs by itself on first use and whenever a parameter was updated in place or moved since.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _abi
from .policy import CtfPolicy

_TWO_LOG2E = 2.0 / math.log(2.0)


def _has_mm_out_dtype():
    """torch.mm / torch.bmm with ``out_dtype=`` (bf16 operands, float32 result: torch >= 2.8).  Without it the view GEMM of the factored
    front runs as the native kernel (ctf_policy_view_gemm: bit-identical, tested) and fc1's weight gradient as one bf16 GEMM."""
    try:
        return all("dtype" in op.overloads() and "dtype_out" in op.overloads() for op in (torch.ops.aten.mm, torch.ops.aten.bmm))
    except Exception:  # pragma: no cover - a torch whose op registry looks different: take the conservative path
        return False


MM_OUT_DTYPE = _has_mm_out_dtype()


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


def conv2_transposed_fragments(conv2_w):
    """conv2.weight [32, 16, 3, 3] (unscaled) -> the data-gradient kernel's A operands (ctf_policy_front_dgrad):
    [tap][lane][j] = W2[out = 8 * (lane >> 4) + j][in = lane & 15][tap]."""
    w2t = np.asarray(conv2_w, np.float32).reshape(32, 16, 9)
    lane, j = np.arange(64), np.arange(8)
    f = np.zeros((9, 64, 8), np.float32)
    for tap in range(9):
        f[tap] = w2t[8 * (lane >> 4)[:, None] + j[None, :], (lane & 15)[:, None], tap]
    return f


def act_column_order(grid_size, meta_len):
    """new column -> reference column of fc1's input (-1 = padding, zero weight): the kernel writes conv2 channel c,
    position p at ((c // 4) * PP + p) * 4 + c % 4 with PP = the positions rounded up to whole 32-position tiles (every
    store instruction then covers whole 128-byte lines); the reference's flatten has it at c * P2 + p; metadata follows."""
    p2 = (grid_size - 4) ** 2
    pp = (p2 + 31) // 32 * 32
    kp = (32 * pp + meta_len + 63) // 64 * 64
    src = np.full(kp, -1, np.int64)
    c, p = np.meshgrid(np.arange(32), np.arange(p2), indexing="ij")
    src[((c // 4) * pp + p) * 4 + c % 4] = c * p2 + p
    src[32 * pp:32 * pp + meta_len] = 32 * p2 + np.arange(meta_len)
    return src


def fc1_patch_fragments(fc1_w, grid_size, meta_len):
    """fc1.weight [256, 32 * P2 + M] (reference column order: channel c of conv2 position p at c * P2 + p, then the metadata) -> the
    A-operand fragments of ctf_policy_fc1_patch (include/ctf_policy.h): float32 [P2 + 2][2][8][64][8], scaled by 2 log2(e);
    block P2 is zero (a patch position outside the image), block P2 + 1 holds the metadata columns."""
    w = np.asarray(fc1_w, np.float64) * _TWO_LOG2E
    p2 = (grid_size - 4) ** 2
    if w.shape != (256, 32 * p2 + meta_len) or meta_len > 32:
        raise ValueError("the factored fc1 path is built for fc1 = Linear(32 * (G - 4)^2 + M, 256) with M <= 32")
    lane, j = np.arange(64), np.arange(8)
    out = np.zeros((p2 + 2, 2, 8, 64, 8), np.float32)
    n = (32 * np.arange(8)[:, None, None] + (lane & 31)[None, :, None])            # [t, lane, 1]
    for s in range(2):
        c = (16 * s + 8 * (lane >> 5)[:, None] + j[None, :])[None, :, :]              # [1, lane, j]: channel / metadata index
        for p in range(p2):
            out[p, s] = w[n, c * p2 + p]
        mcol = np.where(c < meta_len, 32 * p2 + np.minimum(c, meta_len - 1), 0)
        out[p2 + 1, s] = np.where(c < meta_len, w[n, mcol], 0.0)
    return out


def tail_fragments(fc2_w, fc2_b, action_w, action_b, value_w, value_b):
    """fc2 / head parameters -> the fused tail kernel's operands (include/ctf_policy.h, ctf_policy_head): bf16 MFMA
    A-fragments (fc2 scaled by 2 log2(e): its output only feeds a tanh) and float32 biases."""
    w2 = np.asarray(fc2_w, np.float64) * _TWO_LOG2E  # [128, 256]
    if w2.shape != (128, 256):
        raise ValueError("the fused tail is built for fc1 -> 256 -> fc2 -> 128 (agent_network.py:15-16)")
    lane, j = np.arange(64), np.arange(8)
    f2 = np.zeros((4, 16, 64, 8), np.float32)
    for w in range(4):
        for s in range(16):
            f2[w, s] = w2[(32 * w + (lane & 31))[:, None], 16 * s + 8 * (lane >> 5)[:, None] + j[None, :]]
    n_actions = np.asarray(action_w).shape[0]
    if n_actions > 15:
        raise ValueError("at most 15 actions")
    wh = np.zeros((16, 128), np.float64)
    wh[:n_actions] = np.asarray(action_w, np.float64)
    wh[n_actions] = np.asarray(value_w, np.float64).reshape(-1)
    fh = np.zeros((4, 64, 8), np.float32)
    for s in range(4):
        fh[s] = wh[(lane & 15)[:, None], 32 * s + 8 * (lane >> 4)[:, None] + j[None, :]]
    bh = np.zeros(16, np.float32)
    bh[:n_actions] = np.asarray(action_b, np.float32)
    bh[n_actions] = np.asarray(value_b, np.float32).reshape(-1)[0]
    return f2, (np.asarray(fc2_b, np.float64) * _TWO_LOG2E).astype(np.float32), fh, bh


def gather_maps(n_channels, n_actions):
    """Every kernel-side weight layout above is a gather of the module's parameters (times a scale, with zero padding).  The index
    maps — derived once by pushing integer-valued stand-ins through the very functions that define the layouts — let ``prepare()``
    rebuild all of them on the device after every optimiser step without a host round trip.  -> dict name -> int64 array of indices
    into the concatenation of that group's parameters (-1: a zero of the layout)."""
    def stand_ins(shapes):
        out, k = [], 1
        for sh in shapes:
            n = int(np.prod(sh))
            out.append(np.arange(k, k + n, dtype=np.float64).reshape(sh))
            k += n
        return out

    idx = lambda a, scale: np.rint(np.asarray(a, np.float64) / scale).astype(np.int64) - 1
    f1, b1, f2, b2 = conv_fragments(*stand_ins([(16, n_channels, 3, 3), (16,), (32, 16, 3, 3), (32,)]))
    t2, tb2, th, tbh = tail_fragments(*stand_ins([(128, 256), (128,), (n_actions, 128), (n_actions,), (1, 128), (1,)]))
    f2t = conv2_transposed_fragments(stand_ins([(32, 16, 3, 3)])[0])
    return dict(f1=idx(f1, _TWO_LOG2E), b1=idx(b1, _TWO_LOG2E), f2=idx(f2, _TWO_LOG2E), b2=idx(b2, _TWO_LOG2E),
                t2=idx(t2, _TWO_LOG2E), tb2=idx(tb2, _TWO_LOG2E), th=idx(th, 1.0), tbh=idx(tbh, 1.0), f2t=idx(f2t, 1.0))


def fc1_patch_map(grid_size, meta_len):
    """fc1_patch_fragments as a gather of fc1.weight.reshape(-1) (-1: a zero), derived like gather_maps: an integer stand-in pushed
    through the host-side definition."""
    n = 256 * (32 * (grid_size - 4) ** 2 + meta_len)
    stand_in = np.arange(1, n + 1, dtype=np.float64).reshape(256, -1)
    return np.rint(np.asarray(fc1_patch_fragments(stand_in, grid_size, meta_len), np.float64) / _TWO_LOG2E).astype(np.int64) - 1


class _NativeFront(torch.autograd.Function):
    """conv1 -> tanh -> conv2 -> tanh -> flatten ++ metadata of a training step with native kernels for everything but the two
    weight gradients: the FORWARD is ctf_policy_features_train (the activation row fc1 consumes, plus the one-hot image and
    tanh(conv1) channels-last); the BACKWARD's data path is ctf_policy_front_dgrad (tanh' of both layers, conv2's data gradient by
    MFMA, both bias gradients, one launch), and the library's weight-gradient kernels take the channels-last tensors those two wrote."""

    @staticmethod
    def forward(ctx, net, codes, meta, w1, b1, w2, b2):
        ctx.native_wgrad = bool(net.native_wgrad)
        ctx.fused = bool(net.fused_backward)
        act, h0, h1 = net.features_train(codes, meta, want_h0=not ctx.native_wgrad)
        ctx.save_for_backward(act, h0 if h0 is not None else codes, h1, w2)
        ctx.geom = (net.grid_size, int(w1.shape[1]), net.metadata_size, net._ready()["lib"])
        ctx.f2t = net._ready()["f2t"]  # conv2's weights transposed into the data-gradient kernel's operand order (this step's values)
        return act

    @staticmethod
    def backward(ctx, d_act):
        act, h0, h1, w2 = ctx.saved_tensors
        g, c_in, m, lib = ctx.geom
        g1, g2 = g - 2, g - 4
        b, dev = act.shape[0], act.device
        bf, cl = torch.bfloat16, torch.channels_last
        conv_bwd = torch.ops.aten.convolution_backward
        ptr = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        d_act = d_act.to(bf).contiguous()
        if ctx.native_wgrad and ctx.fused:
            # round 4: conv2's data AND weight gradient in one pass (dz2 never leaves the CU), then conv1's weight gradient from the dz1
            # scratch that pass wrote and the code bytes (h0 here is the codes) — ctf_policy_front_backward
            dz1 = torch.empty((b, g1, g1, 16), dtype=bf, device=dev)
            grads = torch.zeros(48 + 32 * 16 * 9 + 16 * 16 * 9, dtype=torch.float32, device=dev)
            db2, db1, dw2, dw1 = grads[:32], grads[32:48], grads[48:48 + 32 * 16 * 9], grads[48 + 32 * 16 * 9:]
            if lib.ctf_policy_front_backward(ptr(d_act), ptr(act), ptr(h1), ptr(h0), ptr(ctx.f2t), b, g, m, ptr(dz1), ptr(dw2), ptr(dw1), ptr(db2),
                                             ptr(db1), dev.index, stream) != 0:
                raise _abi.CtfLibraryError("ctf_policy_front_backward: " + (lib.ctf_policy_last_error() or b"").decode())
            return None, None, None, dw1.view(16, 16, 3, 3)[:, :c_in].contiguous(), db1, dw2.view(32, 16, 3, 3), db2
        # one launch: tanh' of conv2's output (rows of the activation matrix in, channels-last out), conv2's data gradient, tanh' of
        # conv1's output, both bias gradients in float32; the library is left with the two weight gradients
        dz2 = torch.empty((b, g2, g2, 32), dtype=bf, device=dev)
        dz1 = torch.empty((b, g1, g1, 16), dtype=bf, device=dev)
        db = torch.zeros(48, dtype=torch.float32, device=dev)
        db2, db1 = db[:32], db[32:]
        if lib.ctf_policy_front_dgrad(ptr(d_act), ptr(act), ptr(h1), ptr(ctx.f2t), b, g, m, ptr(dz2), ptr(dz1), ptr(db2), ptr(db1), dev.index, stream) != 0:
            raise _abi.CtfLibraryError("ctf_policy_front_dgrad: " + (lib.ctf_policy_last_error() or b"").decode())
        if ctx.native_wgrad:  # both weight gradients on the matrix cores too (h0 here is the codes: the kernel builds the one-hot image itself)
            dw = torch.zeros(32 * 16 * 9 + 16 * 16 * 9, dtype=torch.float32, device=dev)
            dw2, dw1 = dw[:32 * 16 * 9], dw[32 * 16 * 9:]
            if lib.ctf_policy_front_wgrad(ptr(dz2), ptr(h1), ptr(dz1), ptr(h0), b, g, ptr(dw2), ptr(dw1), dev.index, stream) != 0:
                raise _abi.CtfLibraryError("ctf_policy_front_wgrad: " + (lib.ctf_policy_last_error() or b"").decode())
            return None, None, None, dw1.view(16, 16, 3, 3)[:, :c_in].contiguous(), db1, dw2.view(32, 16, 3, 3), db2
        dz2, dz1 = dz2.permute(0, 3, 1, 2), dz1.permute(0, 3, 1, 2)
        h1i = h1.view(b, g1, g1, 16).permute(0, 3, 1, 2)
        _, dw2, _ = conv_bwd(dz2, h1i, w2.to(bf).contiguous(memory_format=cl), [32], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
        h0i = h0.view(b, g, g, 16).permute(0, 3, 1, 2)
        w1_shape = torch.empty((16, 16, 3, 3), dtype=bf, device=act.device).contiguous(memory_format=cl)  # only its shape is used
        _, dw1, _ = conv_bwd(dz1, h0i, w1_shape, [16], [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
        return None, None, None, dw1[:, :c_in].float(), db1, dw2.float(), db2


class _TailLinear(torch.autograd.Function):
    """y = x @ w^T + b for the small dense layers of a TRAINING step (fc2; the two heads as one 16-output layer), bf16 operands.  The
    forward and the data gradient are the library's GEMMs (small K, fine); the weight and bias gradients — reductions over up to a
    million samples into a few thousand numbers, which the library runs at 2 % of either roof — are ctf_policy_linear_wgrad."""

    @staticmethod
    def forward(ctx, x, w, b, lib):
        bf = torch.bfloat16
        x = x.to(bf).contiguous()
        wb = w.to(bf)
        ctx.save_for_backward(x, wb)
        ctx.lib, ctx.has_bias = lib, b is not None
        return torch.nn.functional.linear(x, wb, None if b is None else b.to(bf))

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        dy = dy.to(torch.bfloat16).contiguous()
        dev = dy.device
        n_out, n_in = wb.shape
        grads = torch.zeros(n_out * n_in + n_out, dtype=torch.float32, device=dev)
        dw, db = grads[:n_out * n_in], (grads[n_out * n_in:] if ctx.has_bias else None)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        if ctx.lib.ctf_policy_linear_wgrad(ptr(dy), ptr(x), dy.shape[0], n_out, n_in, ptr(dw), ptr(db), dev.index,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) != 0:
            raise _abi.CtfLibraryError("ctf_policy_linear_wgrad: " + (ctx.lib.ctf_policy_last_error() or b"").decode())
        dx = torch.mm(dy, wb) if ctx.needs_input_grad[0] else None
        return dx, dw.view(n_out, n_in), db, None


class _Fc1Linear(torch.autograd.Function):
    """y = act @ w^T for fc1 of a TRAINING step (w [256, Kp] carries the bias in the column where the front writes 1.0).  Forward and
    weight gradient are the library's GEMMs; the data gradient d_act = dy @ w — 8.3 KB written per sample, which the library's kernel
    does at a quarter of the HBM rate — is ctf_policy_fc1_dgrad."""

    WGRAD_RANGES = 32

    @staticmethod
    def forward(ctx, act, w, lib):
        wb = w.to(torch.bfloat16)
        ctx.save_for_backward(act, wb)
        ctx.lib = lib
        return torch.nn.functional.linear(act, wb)

    @staticmethod
    def backward(ctx, dy):
        act, wb = ctx.saved_tensors
        dy = dy.to(torch.bfloat16).contiguous()
        dev = dy.device
        dw = None
        if ctx.needs_input_grad[1]:
            m = int(dy.shape[0])
            if MM_OUT_DTYPE and m % _Fc1Linear.WGRAD_RANGES == 0 and m // _Fc1Linear.WGRAD_RANGES >= 2048:
                # dW = dy^T act as 32 partial products over sample ranges (float32 out) + their sum: the library runs the ONE GEMM (a
                # 256 x Kp output, K = the batch) as split-K with a fix-up pass at half the rate — 1.10 ms against 0.65 per 262 144
                # samples, 4.11 against 2.50 per 1 048 576 (tools/fc1_wgrad_probe.py); float32 partials instead of one bf16 rounding
                s_ = _Fc1Linear.WGRAD_RANGES
                dw = torch.bmm(dy.view(s_, m // s_, -1).transpose(1, 2), act.view(s_, m // s_, -1), out_dtype=torch.float32).sum(0)
            else:
                dw = torch.mm(dy.t(), act).float()  # (bf16 out, as autocast's linear backward has it)
        dx = None
        if ctx.needs_input_grad[0]:
            m, kp = int(dy.shape[0]), int(wb.shape[1])
            if m % 32 == 0 and kp % 64 == 0:
                dx = torch.empty((m, kp), dtype=torch.bfloat16, device=dev)
                wt = wb.t().contiguous()
                ptr = lambda t: C.c_void_p(t.data_ptr())
                if ctx.lib.ctf_policy_fc1_dgrad(ptr(dy), ptr(wt), m, kp, ptr(dx), dev.index,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) != 0:
                    raise _abi.CtfLibraryError("ctf_policy_fc1_dgrad: " + (ctx.lib.ctf_policy_last_error() or b"").decode())
            else:
                dx = torch.mm(dy, wb)
        return dx, dw, None


class CtfPolicyNative(CtfPolicy):
    tune_placement = True   # the factored front's view buffer: a few candidate allocations timed with the real work on first use (_fact_run)
    native_training = True  # trunk_codes with gradients: the native front as the forward (False: the stock modules, as on CPU)
    native_wgrad = True     # ... and the two convolution weight gradients by ctf_policy_front_wgrad (False: the library's kernels)
    fused_head = False      # the network's tail fused behind the patch product (ctf_policy_fc1_patch_head): bit-identical, and measured
                            # no faster — 0.262 ms per call against 0.179 + 0.076 (the tail then runs at one block per CU) — so off
    native_view_gemm = False  # the factored path's view GEMM by ctf_policy_view_gemm: bit-identical to torch.mm -> hipBLASLt (tested) and
                              # measured slower, 0.205 against 0.185 ms a call: both wait for the same bytes (profiles/r04_view_gemm.md)
    native_fc1_wgrad = False  # fc1's weight gradient by ctf_policy_linear_wgrad too: correct (tested) but no faster than the library's GEMM
                              # (1.08 + 0.10 ms against 1.06 per 262 144 samples: one wave per SIMD, 17 M float atomics, dy re-read per slab)
    native_fc1_dgrad = os.environ.get("CTF_FC1_DGRAD", "1") != "0"  # fc1's data gradient by ctf_policy_fc1_dgrad (off: the library's GEMM)
    native_tail_wgrad = True  # fc2's and the heads' weight / bias gradients by ctf_policy_linear_wgrad (False: the library's GEMMs + reductions)
    fused_backward = True   # ... conv2's weight gradient inside the data-gradient pass (ctf_policy_front_backward; False: three launches)
    factored_fc1 = os.environ.get("CTF_POLICY_FACT", "1") != "0"  # act_from_codes(shared_view=True): fc1 as one GEMM row per (env, view) + a per-agent patch product
                            # (ctf_policy_features_fact / ctf_policy_fc1_patch) instead of one activation row per agent

    def __init__(self, n_actions, n_channels, grid_size, metadata_size, seed=None):
        super().__init__(n_actions, n_channels, grid_size, metadata_size, compute_dtype=torch.bfloat16)
        self.grid_size, self.metadata_size, self.n_channels = grid_size, metadata_size, n_channels
        self._prep = None
        self._maps = None          # device-side index maps of prepare(), built on first use
        # Philox key of the action sampler and its running offset.  The key is unique per INSTANCE: a default-constructed
        # module derives it from the state of torch's generator WITHOUT drawing from it (reproducible under torch.manual_seed,
        # different for every module built, and no later torch draw moves because a module was built or copied), and a
        # copy.deepcopy — how ppo.py-style loops make an opponent — gets a new one: two networks that share (key, offset) would
        # sample from identical uniforms, i.e. perfectly correlated exploration of agent and opponent in self-play.  A pickle /
        # torch.save round trip keeps (key, offset): a restored checkpoint resumes its action stream where it stopped.
        self._seed = self._fresh_key() if seed is None else int(seed) & (2 ** 64 - 1)
        self._calls = 0
        self._act_bufs = {}        # persistent activation matrices of act_from_codes, by (rows, row length, device)
        self.placement_probe_ms = None

    @staticmethod
    def _fresh_key(parent=None):
        """A new module: a key that depends on where torch's CPU generator stands (read, not advanced) — building a module moves
        it (weight init), so every module built gets its own, and the same torch.manual_seed gives the same keys.  A copy:
        derived from its parent's key and the number of copies the parent has handed out."""
        import hashlib

        if parent is None:
            material = torch.get_rng_state().numpy().tobytes()
        else:
            parent._n_copies = getattr(parent, "_n_copies", 0) + 1
            material = parent._seed.to_bytes(8, "little") + parent._n_copies.to_bytes(8, "little")
        return int.from_bytes(hashlib.blake2b(material, digest_size=8).digest(), "little") >> 2

    def reseed(self, seed):
        """Set the action sampler's Philox key explicitly (and restart its offset)."""
        self._seed, self._calls = int(seed) & (2 ** 64 - 1), 0
        return self

    def __getstate__(self):  # copies / pickles carry the parameters, not the kernel-side operands or device buffers
        d = dict(self.__dict__)
        d["_prep"], d["_act_bufs"], d["_maps"] = None, {}, None
        return d

    def __deepcopy__(self, memo):  # a COPY samples from its own stream (see __init__); a restored pickle keeps the original's
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        new._seed, new._calls, new._n_copies = self._fresh_key(parent=self), 0, 0
        return new

    def clone(self, reseed=True):
        """A deep copy; ``reseed=False`` keeps the action sampler's (key, offset), e.g. to replay the original's draws."""
        import copy

        new = copy.deepcopy(self)
        if not reseed:
            new._seed, new._calls = self._seed, self._calls
        return new

    # -- weights in the kernels' / the GEMM's layouts ----------------------------------------------
    def prepare(self):
        """The kernel-side copies of the weights (MFMA operand fragments, scaled biases, fc1's permuted and scaled weight), rebuilt on
        the device: gathers through index maps that are computed once (gather_maps), float64 products rounded to float32 and then to
        bf16 exactly as the host-side definitions round them — no host round trip, so an optimiser step per minibatch costs the
        learner no pipeline stall."""
        dev = self.conv1.weight.device
        if dev.type != "cuda":
            raise _abi.CtfLibraryError("CtfPolicyNative runs on a HIP device only (there is no CPU fallback)")
        lib = _abi.load_library()
        if self._maps is None or self._maps["dev"] != dev:
            order = act_column_order(self.grid_size, self.metadata_size)
            maps = {k: torch.from_numpy(v).to(dev) for k, v in gather_maps(self.n_channels, self.n_actions).items()}
            maps.update(dev=dev, kp=len(order), col_src=torch.from_numpy(np.maximum(order, 0)).to(dev),
                        col_keep=torch.from_numpy((order >= 0).astype(np.float32)).to(dev))
            one_col = np.zeros(len(order), np.float32)
            pp = ((self.grid_size - 4) ** 2 + 31) // 32 * 32
            one_col[32 * pp + self.metadata_size] = 1.0  # the column in which the front kernels write 1.0 (include/ctf_policy.h)
            assert order[32 * pp + self.metadata_size] < 0
            maps["one_col"] = torch.from_numpy(one_col).to(dev)
            if self.fact_supported():
                maps["pf"] = torch.from_numpy(fc1_patch_map(self.grid_size, self.metadata_size)).to(dev)
            assert lib.ctf_policy_act_stride(self.grid_size, self.metadata_size) == len(order)
            self._maps = maps
        m = self._maps
        with torch.no_grad():
            bf = torch.bfloat16
            flat = lambda *ps: torch.cat([q.detach().reshape(-1) for q in ps]).double()
            conv = flat(self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
            tail = flat(self.fc2.weight, self.fc2.bias, self.action_head.weight, self.action_head.bias, self.value_head.weight, self.value_head.bias)
            w2 = flat(self.conv2.weight)

            def take(src, ix, scale, dt):  # float64 product -> float32 -> dt, as conv_fragments / tail_fragments define it
                v = torch.where(ix >= 0, src[ix.clamp(min=0)] * scale, torch.zeros((), dtype=torch.float64, device=dev))
                return v.to(torch.float32).to(dt).contiguous()

            f32 = torch.float32
            fc1 = (self.fc1.weight.float() * _TWO_LOG2E).index_select(1, m["col_src"]) * m["col_keep"]  # zero weight on the row's padding
            self._prep = dict(
                lib=lib, kp=m["kp"], stamp=self._stamp(),
                f1=take(conv, m["f1"], _TWO_LOG2E, bf), b1=take(conv, m["b1"], _TWO_LOG2E, f32),
                f2=take(conv, m["f2"], _TWO_LOG2E, bf), b2=take(conv, m["b2"], _TWO_LOG2E, f32),
                fc1_w=fc1.to(bf).contiguous(), fc1_b=(self.fc1.bias.float() * _TWO_LOG2E).to(bf),
                t2=take(tail, m["t2"], _TWO_LOG2E, bf), tb2=take(tail, m["tb2"], _TWO_LOG2E, f32),
                th=take(tail, m["th"], 1.0, bf), tbh=take(tail, m["tbh"], 1.0, f32),
                col_src=m["col_src"], col_keep=m["col_keep"], f2t=take(w2, m["f2t"], 1.0, bf), one_col=m["one_col"],
            )
            if "pf" in m:  # the factored fc1 path (ctf_policy_fc1_patch): W_flat in the view's column order, the per-position fragments, the bias
                kv = lib.ctf_policy_fact_view_stride(self.grid_size)
                self._prep.update(fc1_view_w=self._prep["fc1_w"][:, :kv].contiguous(),  # [256, KV], k-contiguous: the view GEMM's right operand
                                  pf=take(self.fc1.weight.detach().reshape(-1).double(), m["pf"], _TWO_LOG2E, bf),
                                  fc1_b32=(self.fc1.bias.detach().double() * _TWO_LOG2E).to(f32).contiguous())
        return self

    def _stamp(self):
        # in-place updates (optimiser steps, load_state_dict) bump a tensor's _version; .to() / .cuda() replace the storage
        return tuple((q.data_ptr(), q._version) for q in self.parameters())

    def _ready(self):
        if self._prep is None or self._prep["stamp"] != self._stamp():
            self.prepare()
        return self._prep

    # -- inference from the compact observation ----------------------------------------------------
    def features_from_codes(self, codes, meta, agent_idx, out=None, shared_view=False, self_cells=None):
        """codes uint8 [E, N, G, G], meta float16 [E, N, M], agent_idx: the agents this network plays ->
        bf16 [len(agent_idx) * E, Kp] (row k * E + e = agent agent_idx[k] of env e).  ``shared_view``: the caller's promise
        that these agents see the same tile planes (same team, same reverse flag): the convolutions then run once per env
        plus a small patch per agent, with bit-identical results.  ``self_cells``: int16 [E, N] of ``vec.self_cells`` (where
        each row's bit 7 sits); derived from ``codes`` when not given."""
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
        sc_ptr = None
        if shared_view and len(sel) <= 4:
            if self_cells is None:
                self_cells = (codes >> 7).flatten(2).argmax(dim=2).to(torch.int16)
            if not (self_cells.is_cuda and self_cells.dtype == torch.int16 and self_cells.is_contiguous() and tuple(self_cells.shape) == (E, N)):
                raise ValueError("self_cells must be a contiguous int16 CUDA tensor [E, N]")
            sc_ptr = C.c_void_p(self_cells.data_ptr())
        rc = p["lib"].ctf_policy_features(
            C.c_void_p(codes.data_ptr()), C.c_void_p(meta.data_ptr()), E, N, self.grid_size, self.metadata_size, sel_arr, len(sel),
            C.c_void_p(p["f1"].data_ptr()), C.c_void_p(p["b1"].data_ptr()), C.c_void_p(p["f2"].data_ptr()),
            C.c_void_p(p["b2"].data_ptr()), C.c_void_p(out.data_ptr()), sc_ptr, codes.device.index,
            C.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream))
        if rc != 0:
            raise _abi.CtfLibraryError("ctf_policy_features: " + (p["lib"].ctf_policy_last_error() or b"").decode())
        return out

    # -- fc1 carried through the shared view (include/ctf_policy.h, "fc1 without the per-agent activation matrix") ------------------
    def fact_supported(self):
        return self.grid_size in (11, 15) and self.metadata_size <= 32 and self.metadata_size % 2 == 0

    def _fact_buffers(self, E, A, dev):
        p = self._ready()
        lib = p["lib"]
        key = ("fact", E, A, dev.index)
        b = self._act_bufs.get(key)
        if b is None:
            kv, kr = lib.ctf_policy_fact_view_stride(self.grid_size), lib.ctf_policy_fact_row_stride(self.metadata_size)
            tiles = lib.ctf_policy_fact_max_tiles(E, A, self.grid_size)
            i32 = dict(dtype=torch.int32, device=dev)
            b = dict(kv=kv, kr=kr, tiles=tiles,
                     view=torch.empty((E, kv), dtype=torch.bfloat16, device=dev),
                     prow=torch.zeros((tiles * 128, kr), dtype=torch.bfloat16, device=dev),  # (zeros: the padding slots of a bucket hold finite values)
                     yview=torch.empty((E, 256), dtype=torch.float32, device=dev),
                     y1=torch.empty((A * E, 256), dtype=torch.bfloat16, device=dev),
                     work=torch.zeros(576 + tiles, **i32), slot_of=torch.zeros(A * E, **i32), row_of_slot=torch.zeros(tiles * 128, **i32))
            self._act_bufs[key] = b
        return b

    def fc1_from_codes_factored(self, codes, meta, agent_idx, self_cells):
        """fc1's scaled pre-activation (bf16 [len(agent_idx) * E, 256], what ctf_policy_head consumes) for agents that SHARE A VIEW, without
        the per-agent activation matrix: bucket by own cell, conv front -> view rows + patch rows, one GEMM over the E view rows, the
        per-agent patch product."""
        return self._fact_run(codes, meta, agent_idx, self_cells, None)

    def _fact_run(self, codes, meta, agent_idx, self_cells, head):
        """The factored path up to the patch product; ``head`` = None: -> y1 (ctf_policy_fc1_patch); ``head`` = dict(mask=, given=,
        want_logits=): the network's tail fused behind the patch product (ctf_policy_fc1_patch_head) -> (action, logprob, entropy, value,
        logits or None)."""
        p = self._ready()
        lib = p["lib"]
        E, N = int(codes.shape[0]), int(codes.shape[1])
        sel = [int(i) for i in (agent_idx.tolist() if hasattr(agent_idx, "tolist") else agent_idx)]
        A, dev = len(sel), codes.device
        if not (codes.is_cuda and codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape[2:]) == (self.grid_size,) * 2):
            raise ValueError("codes must be a contiguous uint8 CUDA tensor [E, N, G, G]")
        if not (meta.is_cuda and meta.dtype == torch.float16 and meta.is_contiguous() and tuple(meta.shape) == (E, N, self.metadata_size)):
            raise ValueError("meta must be a contiguous float16 CUDA tensor [E, N, M]")
        if not (self_cells is not None and self_cells.is_cuda and self_cells.dtype == torch.int16 and self_cells.is_contiguous() and tuple(self_cells.shape) == (E, N)):
            raise ValueError("self_cells must be a contiguous int16 CUDA tensor [E, N]")
        if not (1 <= A <= 4 and self.fact_supported()):
            raise ValueError("the factored fc1 path takes 1..4 agents of one view, grid_size 11 or 15, metadata_size <= 32")
        b = self._fact_buffers(E, A, dev)
        sel_arr = (C.c_int32 * A)(*sel)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        G, M = self.grid_size, self.metadata_size

        def ok(rc, what):
            if rc != 0:
                raise _abi.CtfLibraryError(what + ": " + (lib.ctf_policy_last_error() or b"").decode())

        ok(lib.ctf_policy_fact_bucket(ptr(self_cells), E, N, G, sel_arr, A, ptr(b["work"]), ptr(b["slot_of"]), ptr(b["row_of_slot"]),
                                      dev.index, stream), "ctf_policy_fact_bucket")

        def front_and_gemm(view):
            ok(lib.ctf_policy_features_fact(ptr(codes), ptr(meta), ptr(self_cells), E, N, G, M, sel_arr, A, ptr(p["f1"]), ptr(p["b1"]), ptr(p["f2"]),
                                            ptr(p["b2"]), ptr(b["slot_of"]), ptr(view), ptr(b["prow"]), dev.index, stream), "ctf_policy_features_fact")
            if self.native_view_gemm or not MM_OUT_DTYPE:  # float32 out either way: the patch product is added before the one rounding
                ok(lib.ctf_policy_view_gemm(ptr(view), ptr(p["fc1_view_w"]), E, b["kv"], ptr(b["yview"]), dev.index, stream), "ctf_policy_view_gemm")
            elif E % 4 == 0:
                # the library's fastest form of this product (tools/view_gemm_forms_probe.py, 65 536 x 4 096: 0.154 ms; the plain
                # mm(view, W^T stored [KV, 256]) 0.185, mm against the k-contiguous W 0.170): four row ranges as a batch, W k-contiguous
                torch.bmm(view.view(4, E // 4, b["kv"]), p["fc1_view_w"].t().expand(4, b["kv"], 256), out_dtype=torch.float32,
                          out=b["yview"].view(4, E // 4, 256))
            else:
                torch.mm(view, p["fc1_view_w"].t(), out_dtype=torch.float32, out=b["yview"])

        if not b.get("placed"):
            # Large allocations on this pool come in two kinds: the slow one costs the front's stores and the GEMM's reads
            # of the view matrix ~25 % (0.22 against 0.16 ms for the GEMM of a 65 536-env step).  Once per buffer: candidates are timed
            # with the real work, a loser goes back to the driver at once (two held at most), until both kinds were seen or 8 tries.
            b["placed"] = True
            import os

            # (several ranks on ONE device — CTF_BENCH_ONE_DEVICE, a rehearsal — would time each other: no search there)
            if b["view"].numel() * 2 > (256 << 20) and self.tune_placement and not os.environ.get("CTF_BENCH_ONE_DEVICE"):
                def probe(view):
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    front_and_gemm(view)
                    t0.record()
                    for _ in range(2):
                        front_and_gemm(view)
                    t1.record()
                    t1.synchronize()
                    return t0.elapsed_time(t1) / 2
                times = [probe(b["view"])]
                while len(times) < 8 and max(times) < 1.08 * min(times):
                    try:
                        cand = torch.empty_like(b["view"])
                    except torch.cuda.OutOfMemoryError:
                        break
                    times.append(probe(cand))
                    if times[-1] < min(times[:-1]):
                        b["view"], cand = cand, b["view"]
                    # the loser goes back to the DRIVER: the next candidate must be a fresh allocation, not this block again.  (empty_cache
                    # returns the allocator's FREE blocks only — no live tensor of a co-resident learner is touched; it re-allocates what it
                    # had cached.  tune_placement = False, or prepare_placement() from a warm-up, keeps this out of a timed collect().)
                    del cand
                    torch.cuda.empty_cache()
                self.placement_probe_ms = times
        front_and_gemm(b["view"])
        if head is not None:
            B = A * E
            f32 = dict(dtype=torch.float32, device=dev)
            action = torch.empty(B, dtype=torch.int32, device=dev)
            logprob, entropy, value = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
            logits = torch.empty((B, self.n_actions), **f32) if head.get("want_logits") else None
            mask, given = head.get("mask"), head.get("given")
            if mask is not None:
                mask = mask.reshape(-1).to(torch.float32).contiguous()
                if mask.numel() != B:
                    raise ValueError("masking_decision_tensor must have one entry per sample")
            if given is not None:
                given = given.reshape(-1).to(torch.int32).contiguous()
            self._calls += 1
            optr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            ok(lib.ctf_policy_fc1_patch_head(ptr(b["prow"]), ptr(b["row_of_slot"]), ptr(b["work"]), ptr(b["yview"]), ptr(p["pf"]), ptr(p["fc1_b32"]),
                                             E, A, G, M, ptr(p["t2"]), ptr(p["tb2"]), ptr(p["th"]), ptr(p["tbh"]), optr(mask), optr(given),
                                             self.n_actions, C.c_uint64(self._seed), C.c_uint64(self._calls), ptr(action), ptr(logprob),
                                             ptr(entropy), ptr(value), optr(logits), dev.index, stream), "ctf_policy_fc1_patch_head")
            return action, logprob, entropy, value, logits
        ok(lib.ctf_policy_fc1_patch(ptr(b["prow"]), ptr(b["row_of_slot"]), ptr(b["work"]), ptr(b["yview"]), ptr(p["pf"]), ptr(p["fc1_b32"]),
                                    E, A, G, M, ptr(b["y1"]), dev.index, stream), "ctf_policy_fc1_patch")
        return b["y1"]

    # -- the forward of a training step ----------------------------------------------------------------
    def features_train(self, codes, meta, want_h0=True):
        """codes uint8 [B, G, G], meta float16 [B, M] -> (activation rows bf16 [B, Kp] as features_from_codes writes them, the one-hot
        input image bf16 [B, G*G, 16], tanh(conv1) bf16 [B, (G-2)^2, 16]) — ctf_policy_features_train."""
        p = self._ready()
        b, g = int(codes.shape[0]), self.grid_size
        if not (codes.is_cuda and codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape[1:]) == (g, g)):
            raise ValueError("codes must be a contiguous uint8 CUDA tensor [B, G, G]")
        if not (meta.is_cuda and meta.dtype == torch.float16 and meta.is_contiguous() and tuple(meta.shape) == (b, self.metadata_size)):
            raise ValueError("meta must be a contiguous float16 CUDA tensor [B, M]")
        new = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=codes.device)
        act, h0, h1 = new(b, p["kp"]), (new(b, g * g, 16) if want_h0 else None), new(b, (g - 2) ** 2, 16)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = p["lib"].ctf_policy_features_train(ptr(codes), ptr(meta), b, g, self.metadata_size, ptr(p["f1"]), ptr(p["b1"]), ptr(p["f2"]),
                                                ptr(p["b2"]), ptr(act), ptr(h0), ptr(h1), codes.device.index,
                                                C.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream))
        if rc != 0:
            raise _abi.CtfLibraryError("ctf_policy_features_train: " + (p["lib"].ctf_policy_last_error() or b"").decode())
        return act, h0, h1

    def trunk_codes(self, codes, metadata):
        """CtfPolicy.trunk_codes; on a HIP device with gradients enabled the two convolutions' forward is the native front
        (_NativeFront) and fc1 runs on its activation rows with the weight columns gathered into the kernel's order."""
        if not (self.native_training and codes.is_cuda and torch.is_grad_enabled() and self.grid_size in (11, 15)):
            return super().trunk_codes(codes, metadata)
        p = self._ready()
        act = _NativeFront.apply(self, codes.contiguous(), metadata.to(torch.float16).contiguous(), self.conv1.weight, self.conv1.bias,
                                 self.conv2.weight, self.conv2.bias)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            # [256, Kp]: zero weight on the row's padding, and the bias in the column where the front writes 1.0 — the bias gradient then
            # falls out of the weight-gradient GEMM instead of a reduction over the batch of its own
            w = self.fc1.weight.index_select(1, p["col_src"]) * p["col_keep"] + self.fc1.bias[:, None] * p["one_col"]
            if self.native_fc1_wgrad and w.shape[0] == 256 and w.shape[1] % 64 == 0:
                x = torch.tanh(_TailLinear.apply(act, w, None, p["lib"]))  # fc1's weight gradient by the same kernel (column slabs)
            elif self.native_fc1_dgrad and w.shape[0] == 256 and w.shape[1] % 64 == 0:
                x = torch.tanh(_Fc1Linear.apply(act, w, p["lib"]))
            else:
                x = torch.tanh(torch.nn.functional.linear(act, w))
            if self.native_tail_wgrad and self.fc2.weight.shape == (128, 256) and self.n_actions < 16:
                # fc2 and the two heads (as ONE 16-output layer) with native weight / bias gradients (_TailLinear)
                x = torch.tanh(_TailLinear.apply(x, self.fc2.weight, self.fc2.bias, p["lib"]))
                pad = 16 - self.n_actions - 1
                wh = torch.cat((self.action_head.weight, self.value_head.weight, self.action_head.weight.new_zeros((pad, 128))))
                bh = torch.cat((self.action_head.bias, self.value_head.bias, self.action_head.bias.new_zeros(pad)))
                y = _TailLinear.apply(x, wh, bh, p["lib"])
                logits, value = y[:, :self.n_actions], y[:, self.n_actions:self.n_actions + 1]
            else:
                x = torch.tanh(self.fc2(x))
                value, logits = self.value_head(x), self.action_head(x)
        return value.float(), logits.float()

    def _features_tuned(self, codes, meta, agent_idx, shared_view, self_cells, tries=12, slow_over_fast=1.1):
        """features_from_codes into a persistent activation buffer.  Large allocations on this pool come in two kinds — the slow one
        costs both the kernel's stores and the GEMM's reads ~20 % — and the kind is independent from one allocation to
        the next even after a free (tools/placement_probe2.py): on first use candidates are timed with the real work (front kernel + fc1
        GEMM), a loser goes back to the driver at once (two buffers held at most), and the search stops as soon as it has seen both
        kinds, keeping the fast one."""
        p = self._ready()
        rows = len(agent_idx) * int(codes.shape[0])
        key = (rows, p["kp"], codes.device.index)
        buf = self._act_bufs.get(key)
        if buf is not None:
            return self.features_from_codes(codes, meta, agent_idx, out=buf, shared_view=shared_view, self_cells=self_cells)

        def probe(cand):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(3):
                if i == 1:
                    a.record()
                self.features_from_codes(codes, meta, agent_idx, out=cand, shared_view=shared_view, self_cells=self_cells)
                torch.nn.functional.linear(cand, p["fc1_w"], p["fc1_b"])
            b.record()
            b.synchronize()
            return a.elapsed_time(b) / 2

        new = lambda: torch.empty((rows, p["kp"]), dtype=torch.bfloat16, device=codes.device)
        buf = new()
        times = [probe(buf)] if rows * p["kp"] * 2 > (256 << 20) else [0.0]
        while 0.0 < min(times) and len(times) < tries and max(times) < slow_over_fast * min(times):  # until both kinds were seen
            try:
                cand = new()
            except torch.cuda.OutOfMemoryError:
                break
            times.append(probe(cand))
            if times[-1] < min(times[:-1]):
                buf = cand
            del cand
            torch.cuda.empty_cache()
        self._act_bufs[key] = buf
        self.placement_probe_ms = times
        return self.features_from_codes(codes, meta, agent_idx, out=buf, shared_view=shared_view, self_cells=self_cells)

    def _tail(self, feats, mask=None, given=None, want_logits=False):
        """fc1 (bf16 GEMM) then the fused tail -> (action int32, logprob, entropy, value, logits or None), each [B]."""
        p = self._ready()
        y1 = torch.nn.functional.linear(feats, p["fc1_w"], p["fc1_b"])  # pre-activation, scaled by 2 log2(e)
        return self._head(y1, mask, given, want_logits)

    def _head(self, y1, mask=None, given=None, want_logits=False):
        """ctf_policy_head on fc1's scaled pre-activation y1 (bf16 [B, 256])."""
        p = self._ready()
        B, dev = y1.shape[0], y1.device
        f32 = dict(dtype=torch.float32, device=dev)
        action = torch.empty(B, dtype=torch.int32, device=dev)
        logprob, entropy, value = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32)
        logits = torch.empty((B, self.n_actions), **f32) if want_logits else None
        if mask is not None:
            mask = mask.reshape(-1).to(torch.float32).contiguous()
            if mask.numel() != B:
                raise ValueError("masking_decision_tensor must have one entry per sample")
        if given is not None:
            given = given.reshape(-1).to(torch.int32).contiguous()
        self._calls += 1
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = p["lib"].ctf_policy_head(
            ptr(y1), B, ptr(p["t2"]), ptr(p["tb2"]), ptr(p["th"]), ptr(p["tbh"]), ptr(mask), ptr(given), self.n_actions,
            C.c_uint64(self._seed), C.c_uint64(self._calls), ptr(action), ptr(logprob), ptr(entropy), ptr(value), ptr(logits),
            dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise _abi.CtfLibraryError("ctf_policy_head: " + (p["lib"].ctf_policy_last_error() or b"").decode())
        return action, logprob, entropy, value, logits

    def trunk_from_codes(self, codes, meta, agent_idx):
        """-> (value [B, 1], logits [B, A]) float32, B = len(agent_idx) * E, agent-major."""
        out = self._tail(self.features_from_codes(codes, meta, agent_idx), want_logits=True,
                         given=torch.zeros(len(agent_idx) * codes.shape[0], dtype=torch.int32, device=codes.device))
        return out[3].reshape(-1, 1), out[4]

    def act_from_codes(self, codes, meta, agent_idx, masking_decision_tensor, action=None, shared_view=False, self_cells=None):
        """get_action_and_value (agent_network.py:63-81) for agents ``agent_idx`` of every env, from the compact observation:
        -> (action int32 [B], log_prob [B], entropy [B], value [B, 1]).  Sampling: inverse CDF of the masked softmax with one
        Philox4x32-10 uniform per sample, keyed by this module's seed and call count."""
        n_sel = len(agent_idx)
        if shared_view and self.factored_fc1 and self_cells is not None and n_sel <= 4 and self.fact_supported():
            if self.fused_head:  # the network's tail behind the patch product, fc1's output never in HBM (bit-identical to the two calls)
                act, logprob, entropy, value, _ = self._fact_run(codes, meta, agent_idx, self_cells, dict(mask=masking_decision_tensor, given=action))
            else:
                y1 = self.fc1_from_codes_factored(codes, meta, agent_idx, self_cells)
                act, logprob, entropy, value, _ = self._head(y1, mask=masking_decision_tensor, given=action)
        else:
            feats = self._features_tuned(codes, meta, agent_idx, shared_view, self_cells)
            act, logprob, entropy, value, _ = self._tail(feats, mask=masking_decision_tensor, given=action)
        return act, logprob, entropy, value.reshape(-1, 1)
