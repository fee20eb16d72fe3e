"""The native policy front (ctf_policy_features: conv1 -> tanh -> conv2 -> tanh -> flatten ++ metadata, bf16 MFMA) and the
inference path built on it, against
  * a float64 emulation of exactly what the kernel computes (bf16-rounded scaled weights, bf16-rounded hidden
    activations): agreement to one bf16 ulp — this pins every index of the kernel;
  * the float32 network (the torch reference of this floating-point kernel): tolerance 3e-2 on activations in [-1, 1];
  * the reference's own Agent outputs on reference observations (tests/golden/policy_arena.npz): logits within 0.15,
    values within 0.25 (bf16 operands through four layers; the float32 path of policy.py holds 1e-4).
"""
import importlib
import math
import os

import numpy as np
import pytest

from _cases import GOLDEN, pkg
from _policy_weights import fill_

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
native = importlib.import_module("marl-ctf-development_amd.policy_native")
S = 2.0 / math.log(2.0)


def encode(planes):
    """one-hot planes [..., C, G, G] -> codes [..., G, G] (the inverse of expand_codes)."""
    c = planes.shape[-3]
    k = np.arange(1, c, dtype=np.uint8).reshape((c - 1, 1, 1))
    return ((planes[..., 1:, :, :] * k).sum(axis=-3) | (planes[..., 0, :, :] << 7)).astype(np.uint8)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float64)


def emulate(net, planes, meta):
    """float64 evaluation of the kernel's arithmetic: planes [B, C, G, G] (0/1), meta float16 [B, M] -> [B, 32*P2 + M]."""
    F = torch.nn.functional
    t = lambda z: 1.0 - 2.0 / (torch.exp2(z) + 1.0)
    w1, b1 = bf16(net.conv1.weight.detach().cpu().double() * S), (net.conv1.bias.detach().cpu().double() * S).float().double()
    w2, b2 = bf16(net.conv2.weight.detach().cpu().double() * S), (net.conv2.bias.detach().cpu().double() * S).float().double()
    h1 = bf16(t(F.conv2d(planes.double(), w1, b1)))
    h2 = bf16(t(F.conv2d(h1, w2, b2)))
    return torch.cat((h2.flatten(1), bf16(meta.float())), dim=1)


def unpermute(net, feats):
    """kernel column order -> the reference's flatten order; the padding columns (zero fc1 weight) must merely be finite."""
    order = native.act_column_order(net.grid_size, net.metadata_size)
    f = feats.float().cpu()
    assert bool(torch.isfinite(f).all()) and float(f[:, order < 0].abs().max()) <= 1.0
    out = torch.zeros((f.shape[0], int(order.max()) + 1), dtype=torch.float64)
    out[:, order[order >= 0]] = f[:, order >= 0].double()
    return out


GOLDEN_POLICIES = ("policy_arena", "policy_split")  # outputs of the reference's own Agent on 8_arena (G = 15) / 0_the_split (G = 11)


def _golden(name="policy_arena"):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    shape = tuple(int(x) for x in z["grid_shape"])
    grids = np.unpackbits(z["grids"])[: int(np.prod(shape))].reshape(shape)  # [T*N, C, G, G], step-major
    metas = z["metas"].view(np.float16)
    return z, grids, metas, int(z["n_agents"])


@pytest.mark.parametrize("golden", GOLDEN_POLICIES)
def test_features_match_the_emulation_and_the_float32_network_on_reference_observations(golden):
    z, grids, metas, n = _golden(golden)
    T = grids.shape[0] // n
    c, g = grids.shape[1], grids.shape[2]
    net = fill_(native.CtfPolicyNative(9, c, g, metas.shape[1])).cuda()
    codes = torch.tensor(encode(grids).reshape(T, n, g, g), device="cuda")
    meta = torch.tensor(metas.reshape(T, n, -1), device="cuda")
    feats = net.features_from_codes(codes, meta, list(range(n)))       # row k * T + e
    got = unpermute(net, feats).reshape(n, T, -1).transpose(0, 1).reshape(T * n, -1)  # -> row e * n + k
    want = emulate(net, torch.tensor(grids), torch.tensor(metas))
    ulp = 2.0 ** -7
    diff = (got - want).abs()
    assert float(diff.max()) <= ulp, float(diff.max())
    assert float((diff > 0).double().mean()) < 0.02  # a rounding flip here and there (accumulation order), nothing more
    with torch.no_grad():
        cpu = fill_(native.CtfPolicy(9, c, g, metas.shape[1]))
        x = torch.tanh(cpu.conv2(torch.tanh(cpu.conv1(torch.tensor(grids, dtype=torch.float32)))))
        ref32 = torch.cat((x.flatten(1), torch.tensor(metas).float()), dim=1).double()
    assert float((got - ref32).abs().max()) < 3e-2  # bf16 operands vs the float32 network, activations in [-1, 1]
    # a subset of agents, in another order: the same rows
    sub = net.features_from_codes(codes, meta, [n - 1, 1])
    assert torch.equal(sub[:T], feats[(n - 1) * T:n * T]) and torch.equal(sub[T:], feats[T:2 * T])


@pytest.mark.parametrize("g,c,n,e", [(11, 8, 4, 70), (15, 14, 8, 33), (20, 14, 8, 9), (7, 5, 2, 130), (32, 15, 3, 5)])
def test_features_on_random_codes_for_other_grid_sizes(g, c, n, e):
    """Every kernel instantiation (G = 11, 15 and the generic one) on random codes — partial waves, odd sizes."""
    rng = np.random.default_rng(g * 100 + c)
    m = 2 * n + 6
    low = rng.integers(0, c, (e, n, g, g)).astype(np.uint8) * (rng.random((e, n, g, g)) < 0.3)
    selfbit = (rng.random((e, n, g, g)) < 0.02).astype(np.uint8) << 7
    codes = (low | selfbit).astype(np.uint8)
    metas = rng.random((e, n, m)).astype(np.float16)
    net = fill_(native.CtfPolicyNative(9, c, g, m)).cuda()
    feats = net.features_from_codes(torch.tensor(codes, device="cuda"), torch.tensor(metas, device="cuda"), list(range(n)))
    got = unpermute(net, feats).reshape(n, e, -1).transpose(0, 1).reshape(e * n, -1)
    planes = torch.tensor(pkg.expand_codes(codes, c).reshape(e * n, c, g, g))
    want = emulate(net, planes, torch.tensor(metas.reshape(e * n, m)))
    assert float((got - want).abs().max()) <= 2.0 ** -7


@pytest.mark.parametrize("golden", GOLDEN_POLICIES)
def test_native_inference_is_close_to_the_reference_agent_and_respects_the_mask(golden):
    z, grids, metas, n = _golden(golden)
    T = grids.shape[0] // n
    c, g = grids.shape[1], grids.shape[2]
    net = fill_(native.CtfPolicyNative(9, c, g, metas.shape[1])).cuda()
    codes = torch.tensor(encode(grids).reshape(T, n, g, g), device="cuda")
    meta = torch.tensor(metas.reshape(T, n, -1), device="cuda")
    with torch.no_grad():
        value, logits = net.trunk_from_codes(codes, meta, list(range(n)))
        back = lambda t: t.reshape(n, T, -1).transpose(0, 1).reshape(T * n, -1).cpu().numpy()
        assert np.abs(back(logits) - z["logits"]).max() < 0.15
        assert np.abs(back(value) - z["value"]).max() < 0.25
        masks = torch.tensor(z["masks"].reshape(T, n).T.reshape(-1).copy(), device="cuda")  # agent-major like the rows
        action, logprob, entropy, _ = net.act_from_codes(codes, meta, list(range(n)), masks)
        assert bool((action[masks == 1] < 5).all())  # decision 1: only actions 0..4 (agent_network.py:66-75)
        assert bool(torch.isfinite(logprob).all()) and bool(torch.isfinite(entropy).all())
        # the stock forward of the same module (one-hot planes in, bf16 autocast) agrees with the native path
        v2, l2 = net(torch.tensor(grids, device="cuda"), torch.tensor(metas, device="cuda"))
        assert np.abs(back(logits) - l2.cpu().numpy()).max() < 0.15


def _emulate_tail(net, y1):
    """float64 evaluation of ctf_policy_head's arithmetic on y1 = bf16 fc1 output (scaled): -> (logits [B, A], value [B])."""
    t = lambda z: 1.0 - 2.0 / (torch.exp2(z) + 1.0)
    cpu = lambda p: p.detach().cpu().double()
    x = bf16(t(y1.cpu().double()))
    z2 = x @ bf16(cpu(net.fc2.weight) * S).T + (cpu(net.fc2.bias) * S).float().double()
    h2 = bf16(t(z2))
    logits = h2 @ bf16(cpu(net.action_head.weight)).T + cpu(net.action_head.bias).float().double()
    value = h2 @ bf16(cpu(net.value_head.weight)).T + cpu(net.value_head.bias).float().double()
    return logits, value.reshape(-1)


def test_fused_tail_matches_its_emulation_and_torch_distribution_math():
    net = fill_(native.CtfPolicyNative(9, 14, 15, 22, seed=7)).cuda()
    B = 300  # two full tiles of 128 and a ragged one
    g = torch.Generator().manual_seed(1)
    y1 = (torch.randn((B, 256), generator=g) * 2.0).to(torch.bfloat16).cuda()
    want_logits, want_value = _emulate_tail(net, y1)
    p = net._ready()
    lib = p["lib"]
    import ctypes as C

    def head(mask=None, given=None, offset=1):
        f32 = dict(dtype=torch.float32, device="cuda")
        action = torch.empty(B, dtype=torch.int32, device="cuda")
        lp, ent, val, logits = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32), torch.empty((B, 9), **f32)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = lib.ctf_policy_head(ptr(y1), B, ptr(p["t2"]), ptr(p["tb2"]), ptr(p["th"]), ptr(p["tbh"]), ptr(mask), ptr(given), 9,
                                 C.c_uint64(7), C.c_uint64(offset), ptr(action), ptr(lp), ptr(ent), ptr(val), ptr(logits), 0,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.ctf_policy_last_error()
        torch.cuda.synchronize()
        return action, lp, ent, val, logits

    mask = (torch.arange(B, device="cuda") % 3 == 0).float()
    action, lp, ent, val, logits = head(mask)
    assert float((logits.cpu().double() - want_logits).abs().max()) < 5e-3  # bf16 rounding flips of the hidden layer
    assert float((val.cpu().double() - want_value).abs().max()) < 5e-3
    assert float((logits.cpu().double() - want_logits).abs().mean()) < 2e-4
    masked = logits + (torch.where(mask[:, None] == 1, net.mask_5[None, :], torch.ones_like(logits)) - 1.0) * 1e9
    dist = torch.distributions.Categorical(logits=masked)
    assert bool((action >= 0).all()) and bool((action < 9).all()) and bool((action[mask == 1] < 5).all())
    assert torch.allclose(lp, dist.log_prob(action.long()), atol=1e-4)
    assert torch.allclose(ent, dist.entropy(), atol=1e-4)
    # evaluating given actions: the reference's `action=` argument
    given = (torch.arange(B, device="cuda") % 5).to(torch.int32)
    a2, lp2, ent2, val2, _ = head(mask, given)
    assert torch.equal(a2, given) and torch.allclose(lp2, dist.log_prob(given.long()), atol=1e-4) and torch.equal(val2, val)
    # same key and counter: the same draw; another counter: another draw
    assert torch.equal(head(mask, offset=1)[0], action) and not torch.equal(head(mask, offset=2)[0], action)


def test_fused_tail_samples_the_softmax():
    net = fill_(native.CtfPolicyNative(9, 14, 15, 22, seed=11)).cuda()
    B = 1 << 17
    row = (torch.randn((1, 256), generator=torch.Generator().manual_seed(3)) * 0.7).to(torch.bfloat16)
    y1 = row.expand(B, 256).contiguous().cuda()
    for decision in (0.0, 1.0):
        mask = torch.full((B,), decision, device="cuda")
        action, lp, ent, val, logits = net._head(y1, mask=mask, want_logits=True)
        masked = logits[0] + ((net.mask_5 if decision == 1.0 else torch.ones_like(net.mask_5)) - 1.0) * 1e9
        probs = torch.softmax(masked, dim=0)
        freq = torch.bincount(action.long(), minlength=9).float() / B
        assert float((freq - probs).abs().max()) < 6e-3, (freq, probs)  # ~4 sigma at B = 131072


def test_decision_values_other_than_0_or_1_give_the_references_all_zero_mask():
    """agent_network.py:71-75: a decision that is neither 0 nor 1 leaves the all-zero mask, every logit gets -1e9 and the
    float32 distribution is uniform."""
    net = fill_(native.CtfPolicyNative(9, 14, 15, 22, seed=5)).cuda()
    B = 4096
    y1 = (torch.randn((B, 256), generator=torch.Generator().manual_seed(2)) * 0.7).to(torch.bfloat16).cuda()
    mask = torch.full((B,), 2.0, device="cuda")
    action, lp, ent, val, logits = net._head(y1, mask=mask, want_logits=True)
    dist = torch.distributions.Categorical(logits=logits + (torch.zeros_like(logits) - 1.0) * 1e9)
    assert torch.allclose(lp, dist.log_prob(action.long()), atol=1e-4) and torch.allclose(ent, dist.entropy(), atol=1e-4)
    # (in float32 -1e9 + log 9 == -1e9: torch reports log-prob 0 and entropy 0 for this distribution, and so does the kernel)
    freq = torch.bincount(action.long(), minlength=9).float() / B
    assert float((freq - 1.0 / 9.0).abs().max()) < 0.03  # the draw itself is uniform: nothing is masked off


def test_every_policy_instance_samples_from_its_own_stream():
    """Two default-constructed networks, and a deepcopy of one, must not share (Philox key, offset): identical uniforms for
    agent and opponent would correlate their exploration perfectly in self-play.  A pickle round trip is a checkpoint: it keeps
    (key, offset) and resumes the stream.  Making keys reads torch's generator but never draws from it."""
    import copy
    import pickle

    a = fill_(native.CtfPolicyNative(9, 14, 15, 22))
    b = fill_(native.CtfPolicyNative(9, 14, 15, 22))
    before = torch.get_rng_state()
    c = copy.deepcopy(a)
    c2 = copy.deepcopy(a)
    assert torch.equal(torch.get_rng_state(), before)  # copies do not advance torch's generator
    assert len({a._seed, b._seed, c._seed, c2._seed}) == 4
    assert a.clone(reseed=False)._seed == a._seed
    B = 8192
    y1 = torch.zeros((B, 256), dtype=torch.bfloat16, device="cuda")  # identical, flat logits for every sample
    a.cuda()._head(y1)  # the original has drawn once ...
    d = pickle.loads(pickle.dumps(a))  # ... and its checkpoint resumes from there
    assert (d._seed, d._calls) == (a._seed, a._calls) and a._calls > 0
    assert torch.equal(d.cuda()._head(y1)[0], a._head(y1)[0])
    draws = [net.cuda()._head(y1)[0] for net in (a, b, c, c2)]
    for i in range(4):
        for j in range(i + 1, 4):
            assert float((draws[i] == draws[j]).float().mean()) < 0.5  # independent draws agree ~1/9 of the time
    torch.manual_seed(123)
    k1 = native.CtfPolicyNative(9, 14, 15, 22)._seed
    torch.manual_seed(123)
    assert native.CtfPolicyNative(9, 14, 15, 22)._seed == k1  # reproducible under torch.manual_seed
    assert native.CtfPolicyNative(9, 14, 15, 22, seed=77).reseed(78)._seed == 78


def test_full_size_batch_is_deterministic_and_matches_the_emulation_on_a_sample():
    """BASELINE size: 65 536 arena envs x 8 agents = 524 288 samples (a 4.1 GB activation matrix: 64-bit offsets)."""
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    E = 65536
    vec = pkg.VecGridworldCtf(E, device=0, **kw)
    acts = torch.empty((E, 8), dtype=torch.int8, device="cuda")
    for t in range(6):
        vec.random_actions(acts, seed=5, step=t)
        vec.step(acts)
    codes, meta = vec.observe_codes()
    net = fill_(native.CtfPolicyNative(9, vec.N_CHANNELS, 15, vec.META_LEN, seed=3)).cuda()
    sel = list(range(8))
    f1 = net.features_from_codes(codes, meta, sel)
    for _ in range(3):  # run-to-run identical (this caught a prefetch that was waited for one loop iteration too late)
        assert torch.equal(net.features_from_codes(codes, meta, sel), f1)
    assert bool(torch.isfinite(f1.float()).all())
    real = torch.from_numpy(native.act_column_order(15, vec.META_LEN) >= 0).cuda()
    for team in (0, 1):  # the shared-view kernel at full size: identical to the per-agent rows
        team_agents = [i for i in range(8) if vec.AGENT_TEAMS[i] == team]
        for _ in range(2):
            ft = net.features_from_codes(codes, meta, team_agents, shared_view=True, self_cells=vec.self_cells)
            for j, ag in enumerate(team_agents):
                assert torch.equal(ft[j * E:(j + 1) * E][:, real], f1[ag * E:(ag + 1) * E][:, real]), (team, ag)
    rows = torch.tensor([0, 1, E - 1, E, 3 * E + 17, 8 * E - 1, 5 * E + 4242], device="cuda")  # row k * E + e
    k, e = (rows // E).cpu(), (rows % E).cpu()
    planes = torch.tensor(pkg.expand_codes(codes.cpu().numpy()[e, k], vec.N_CHANNELS))
    want = emulate(net, planes, meta.cpu()[e, k])
    assert float((unpermute(net, f1[rows]) - want).abs().max()) <= 2.0 ** -7
    mask = torch.zeros(8 * E, device="cuda")
    a1 = net.act_from_codes(codes, meta, sel, mask)
    net._calls -= 1  # replay the same Philox counter
    a2 = net.act_from_codes(codes, meta, sel, mask)
    # same Philox draw; the BLAS library's fc1 GEMM (a stream-K kernel) may round differently from run to run, which can move
    # a sample that sits on a CDF boundary
    assert float((a1[0] == a2[0]).float().mean()) > 0.995
    assert torch.allclose(a1[3], a2[3], atol=2e-2) and torch.allclose(a1[2], a2[2], atol=2e-2)
    assert int(a1[0].min()) >= 0 and int(a1[0].max()) <= 8 and bool(torch.isfinite(a1[1]).all())
    vec.close()


def test_full_size_split_batch_is_deterministic_and_matches_the_emulation():
    """G = 11 (0_the_split) at 65 536 envs x 4 agents: the NP = 2 instantiation of both front kernels, whose prefetch wait
    once listed one register several times (the compiler then copied the load's destination before the wait: timing-
    dependent activations that the small goldens never exposed).  Run-to-run identical, shared-view == per-agent, and a
    sample spread over the whole batch equal to the float64 emulation."""
    kw = dict(pkg.configs.SPLIT_KWARGS, SCENARIO=pkg.CtfScenarios.arrow)
    E = 65536
    vec = pkg.VecGridworldCtf(E, device=0, py_seeds=np.arange(E) + 11, np_seeds=np.arange(E) + 11, **kw)
    n = vec.N_AGENTS
    acts = torch.empty((E, n), dtype=torch.int8, device="cuda")
    for t in range(9):
        vec.random_actions(acts, seed=6, step=t)
        vec.step(acts)
    codes, meta = vec.observe_codes()
    net = fill_(native.CtfPolicyNative(9, vec.N_CHANNELS, 11, vec.META_LEN, seed=3)).cuda()
    sel = list(range(n))
    f1 = net.features_from_codes(codes, meta, sel)
    for _ in range(4):
        assert torch.equal(net.features_from_codes(codes, meta, sel), f1)
    real = torch.from_numpy(native.act_column_order(11, vec.META_LEN) >= 0).cuda()
    for team in (0, 1):
        team_agents = [i for i in range(n) if vec.AGENT_TEAMS[i] == team]
        for _ in range(3):
            ft = net.features_from_codes(codes, meta, team_agents, shared_view=True, self_cells=vec.self_cells)
            for j, ag in enumerate(team_agents):
                assert torch.equal(ft[j * E:(j + 1) * E][:, real], f1[ag * E:(ag + 1) * E][:, real]), (team, ag)
    g = torch.Generator().manual_seed(0)
    rows = torch.cat((torch.tensor([0, 1, E - 1, E, n * E - 1]), torch.randint(0, n * E, (251,), generator=g))).cuda()  # row k * E + e
    k, e = (rows // E).cpu(), (rows % E).cpu()
    planes = torch.tensor(pkg.expand_codes(codes.cpu().numpy()[e, k], vec.N_CHANNELS))
    want = emulate(net, planes, meta.cpu()[e, k])
    assert float((unpermute(net, f1[rows]) - want).abs().max()) <= 2.0 ** -7
    vec.close()


@pytest.mark.parametrize("scenario,kwname,team", [("arena_iii", "ARENA_KWARGS", 0), ("arena_iii", "ARENA_KWARGS", 1), ("arrow", "SPLIT_KWARGS", 1)])
def test_shared_view_kernel_is_bit_identical_to_the_per_agent_kernel(scenario, kwname, team):
    """Teammates share the tile planes: the team kernel (convolutions once per env + a 3x3 / 5x5 patch per agent) must
    reproduce the per-agent kernel bit for bit — agents in corners, next to each other, carrying flags, G = 15 and 11."""
    kw = dict(getattr(pkg.configs, kwname), SCENARIO=getattr(pkg.CtfScenarios, scenario))
    E = 777
    vec = pkg.VecGridworldCtf(E, device=0, py_seeds=np.arange(E) + 5, np_seeds=np.arange(E) + 5, **kw)
    n = vec.N_AGENTS
    agents = [i for i in range(n) if vec.AGENT_TEAMS[i] == team]
    net = fill_(native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN)).cuda()
    acts = torch.empty((E, n), dtype=torch.int8, device="cuda")
    for t in range(60):
        vec.random_actions(acts, seed=21, step=t)
        vec.step(acts, auto_reset=True)
        if t % 6 == 0:
            codes, meta = vec.observe_codes()
            real = torch.from_numpy(native.act_column_order(vec.GRID_SIZE, vec.META_LEN) >= 0).cuda()  # not the don't-care padding
            ref = net.features_from_codes(codes, meta, agents)[:, real]
            got = net.features_from_codes(codes, meta, agents, shared_view=True, self_cells=vec.self_cells)[:, real]
            assert torch.equal(ref, got), (t, int((ref != got).sum()))
            assert torch.equal(vec.self_cells, (codes >> 7).flatten(2).argmax(dim=2).to(torch.int16))  # where bit 7 sits
            rev = list(reversed(agents))  # any order of the group's agents
            assert torch.equal(net.features_from_codes(codes, meta, rev, shared_view=True)[:, real], net.features_from_codes(codes, meta, rev)[:, real])
    one = net.features_from_codes(codes, meta, agents[:1], shared_view=True)[:, real]
    assert torch.equal(one, ref[:E])
    vec.close()


def test_kernel_side_weights_follow_parameter_updates():
    z, grids, metas, n = _golden()
    T = grids.shape[0] // n
    net = fill_(native.CtfPolicyNative(9, grids.shape[1], grids.shape[2], metas.shape[1])).cuda()
    codes = torch.tensor(encode(grids).reshape(T, n, 15, 15), device="cuda")
    meta = torch.tensor(metas.reshape(T, n, -1), device="cuda")
    v0, l0 = net.trunk_from_codes(codes, meta, [0])
    opt = torch.optim.SGD(net.parameters(), lr=0.5)
    for q in net.parameters():
        q.grad = torch.ones_like(q) * 0.01
    opt.step()  # in place: no explicit prepare()
    v1, l1 = net.trunk_from_codes(codes, meta, [0])
    assert not torch.equal(l0, l1)
    v2, l2 = net(torch.tensor(grids[0::n], device="cuda"), torch.tensor(metas[0::n], device="cuda"))  # the stock forward, same weights
    assert float((l1 - l2).abs().max()) < 0.15


@pytest.mark.parametrize("g,c,n", [(15, 14, 8), (11, 8, 4)])
def test_training_forward_through_the_native_front_gives_the_stock_paths_outputs_and_gradients(g, c, n):
    """CtfPolicyNative.trunk_codes with gradients = the native front as the forward (ctf_policy_features_train: activation rows
    bit-identical to the inference kernel's, plus the one-hot image and tanh(conv1) channels-last) and the library's convolution
    gradients as the backward, against the same module through the stock channels-last path (native_training = False)."""
    rng = np.random.default_rng(g)
    m, b = 2 * n + 6, 777  # a partial wave and a partial block at the end
    low = rng.integers(0, c, (b, g, g)).astype(np.uint8) * (rng.random((b, g, g)) < 0.3)
    codes = low.copy()
    cell = rng.integers(0, g * g, b)
    codes.reshape(b, -1)[np.arange(b), cell] |= 128
    codes_t = torch.tensor(codes, device="cuda")
    meta_t = torch.tensor(rng.random((b, m)).astype(np.float16), device="cuda")
    net = fill_(native.CtfPolicyNative(9, c, g, m)).cuda()
    # the forward's three outputs
    act, h0, h1 = net.features_train(codes_t, meta_t)
    ref = net.features_from_codes(codes_t.reshape(b, 1, g, g), meta_t.reshape(b, 1, m), [0])
    assert torch.equal(act, ref)
    planes = torch.tensor(pkg.expand_codes(codes, c)).cuda()
    assert torch.equal(h0.reshape(b, g, g, 16)[..., :c].permute(0, 3, 1, 2).float(), planes.float()) and float(h0[..., c:].abs().max()) == 0.0
    w1, b1 = (net.conv1.weight.detach().double() * S).to(torch.bfloat16).double(), (net.conv1.bias.detach().double() * S).float().double()
    h1_want = (1.0 - 2.0 / (torch.exp2(torch.nn.functional.conv2d(planes.double(), w1, b1)) + 1.0)).permute(0, 2, 3, 1).reshape(b, (g - 2) ** 2, 16)
    assert float((h1.double() - h1_want).abs().max()) <= 2.0 ** -7
    # outputs and gradients of a training step, both ways
    weight = torch.tensor(rng.standard_normal((b, 10)), device="cuda", dtype=torch.float32)

    def grads(native_training, dtype=torch.bfloat16):
        net.native_training, net.compute_dtype = native_training, dtype
        net.zero_grad()
        value, logits = net(codes_t, meta_t.float())
        loss = (torch.cat((logits, value), dim=1) * weight).sum() / b
        loss.backward()
        return loss.item(), logits.detach().clone(), {k: q.grad.detach().clone() for k, q in net.named_parameters()}

    _, lgf, gf = grads(False, torch.float32)  # the float32 network: what both bf16 paths approximate
    l0, lg0, g0 = grads(False)
    l1, lg1, g1 = grads(True)
    assert bool(((lg0 - lg1).abs() <= 4e-2 + 2.0 ** -7 * lg0.abs()).all()) and abs(l0 - l1) <= 2e-2 * max(1.0, abs(l0))  # bf16 outputs
    report = {}
    for k in gf:
        den = max(float(gf[k].norm()), 1e-12)
        e_stock, e_native = float((g0[k] - gf[k]).norm()) / den, float((g1[k] - gf[k]).norm()) / den
        report[k] = (round(e_stock, 4), round(e_native, 4))
        # as close to the float32 gradients as the stock bf16 path is (the two round at different places)
        assert e_native <= 1.5 * e_stock + 1e-2 and e_native <= 5e-2, (k, e_stock, e_native)
    print("relative gradient error against float32 (stock bf16 path, native front):", report)


def test_native_path_fails_loudly_off_gpu():
    net = native.CtfPolicyNative(9, 14, 15, 22)
    with pytest.raises(pkg._abi.CtfLibraryError):
        net.prepare()


def test_rollout_and_duel_with_the_native_policy_run_end_to_end():
    rollout = importlib.import_module("marl-ctf-development_amd.rollout")
    duel = importlib.import_module("marl-ctf-development_amd.duel")
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    vec = pkg.VecGridworldCtf(512, device=0, **kw)
    a = native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).cuda()
    b = native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN).cuda()
    out = rollout.BatchedRolloutCollector(vec, 8, 0).collect(a, b)
    assert out["grid_codes"].shape == (8 * 4, 512, 15, 15) and vec._obs is None
    assert bool(torch.isfinite(out["values"]).all()) and bool(torch.isfinite(out["logprobs"]).all())
    assert float(out["actions"].max()) <= 8 and vec.status() == 0
    # the stored codes are what the policy saw: re-evaluating them gives the stored values
    codes = out["grid_codes"][:4].transpose(0, 1).contiguous()            # step 0: [E, 4 trained agents, G, G]
    meta = out["metadata_states"][:4].transpose(0, 1).contiguous().to(torch.float16)
    with torch.no_grad():
        value, _ = a.trunk_from_codes(codes, meta, [0, 1, 2, 3])
    assert torch.allclose(value.reshape(4, 512), out["values"][:4], atol=2e-2)  # the BLAS GEMM between the two kernels may round differently per call
    res = duel.batched_duel(vec, a, b, max_steps=40)
    assert res["steps"] == 41 and res["metrics"].shape == (512, 13, 8) and vec.status() == 0
    vec.close()


def test_overlapping_the_two_teams_policy_kernels_changes_no_result():
    """BatchedRolloutCollector in compact mode runs the opponent's conv front on a side stream beside the trained team's fc1 GEMM
    and head (rollout._two_teams_overlapped): every tensor of the rollout equals the one-stream order's, bit for bit."""
    rollout = importlib.import_module("marl-ctf-development_amd.rollout")
    kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
    outs = []
    for overlap in (False, True):
        vec = pkg.VecGridworldCtf(3000, device=0, **kw)
        torch.manual_seed(11)
        a = fill_(native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN, seed=101)).cuda()
        b = fill_(native.CtfPolicyNative(9, vec.N_CHANNELS, vec.GRID_SIZE, vec.META_LEN, seed=202)).cuda()
        a.factored_fc1 = b.factored_fc1 = False  # the overlapped order exists for the activation-matrix path only (rollout._two_teams_overlapped)
        col = rollout.BatchedRolloutCollector(vec, 12, 0)
        col.overlap_teams = overlap
        out = col.collect(a, b)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in out.items()})
        assert vec.status() == 0
        vec.close()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("g,c", [(15, 14), (11, 8)])
def test_backward_kernels_of_the_conv_front_match_their_float64_definitions(g, c):
    """ctf_policy_front_dgrad and ctf_policy_front_wgrad on random bf16 inputs against float64 evaluations of exactly what they
    define: dz2 = d_act (1 - act^2) relaid channels-last, dh1 = conv2's data gradient of it, dz1 = bf16(dh1) (1 - h1^2), the two bias
    gradients, and the two weight gradients as contractions over positions (conv1's against the one-hot image of the codes)."""
    import ctypes as C

    abi = importlib.import_module("marl-ctf-development_amd._abi")
    lib = abi.load_library()
    rng = np.random.default_rng(100 + g)
    b, m = 301, 22 if g == 15 else 14
    g1, g2 = g - 2, g - 4
    p1, p2 = g1 * g1, g2 * g2
    pp = (p2 + 31) // 32 * 32
    kp = lib.ctf_policy_act_stride(g, m)
    dev = "cuda"
    bf = torch.bfloat16
    t = lambda a: torch.tensor(a, device=dev)
    act = t(np.tanh(rng.standard_normal((b, kp))).astype(np.float32)).to(bf)
    d_act = t((rng.standard_normal((b, kp)) * 0.1).astype(np.float32)).to(bf)
    h1 = t(np.tanh(rng.standard_normal((b, p1, 16))).astype(np.float32)).to(bf)
    w2 = t((rng.standard_normal((32, 16, 3, 3)) * 0.2).astype(np.float32)).to(bf)
    codes = (rng.integers(0, c, (b, g, g)) * (rng.random((b, g, g)) < 0.4)).astype(np.uint8)
    codes.reshape(b, -1)[np.arange(b), rng.integers(0, g * g, b)] |= 128
    codes_t = t(codes)
    f2t = t(native.conv2_transposed_fragments(w2.float().cpu().numpy())).to(bf).contiguous()
    dz2 = torch.empty((b, p2, 32), dtype=bf, device=dev)
    dz1 = torch.empty((b, p1, 16), dtype=bf, device=dev)
    db = torch.zeros(48, dtype=torch.float32, device=dev)
    ptr = lambda x: C.c_void_p(x.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.ctf_policy_front_dgrad(ptr(d_act), ptr(act), ptr(h1), ptr(f2t), b, g, m, ptr(dz2), ptr(dz1), ptr(db[:32]), ptr(db[32:]), 0, stream) == 0
    dw = torch.zeros(32 * 16 * 9 + 16 * 16 * 9, dtype=torch.float32, device=dev)
    assert lib.ctf_policy_front_wgrad(ptr(dz2), ptr(h1), ptr(dz1), ptr(codes_t), b, g, ptr(dw[:4608]), ptr(dw[4608:]), 0, stream) == 0
    torch.cuda.synchronize()
    # ---- float64 definitions (every bf16 rounding the kernels make is made here too)
    r16 = lambda x: x.to(torch.float32).to(bf).double()
    order = native.act_column_order(g, m)  # kernel column -> reference column c * P2 + p
    cols = torch.tensor(np.where((order >= 0) & (order < 32 * p2))[0], device=dev)
    refcol = torch.tensor(order[(order >= 0) & (order < 32 * p2)], device=dev)
    gfl = lambda x: torch.zeros((b, 32 * p2), dtype=torch.float64, device=dev).index_copy_(1, refcol, x.double()[:, cols]).reshape(b, 32, g2, g2)
    dz2_want = r16(gfl(d_act) * (1.0 - gfl(act) ** 2))                                   # [B, 32, G2, G2]
    assert torch.equal(dz2.double().reshape(b, g2, g2, 32).permute(0, 3, 1, 2), dz2_want)
    assert torch.allclose(db[:32].double(), (gfl(d_act) * (1.0 - gfl(act) ** 2)).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-4)
    h1i = h1.double().reshape(b, g1, g1, 16).permute(0, 3, 1, 2)
    dh1 = torch.nn.functional.conv_transpose2d(dz2_want, w2.double())                    # conv2's data gradient
    dz1_exact = r16(dh1) * (1.0 - h1i ** 2)
    got1 = dz1.double().reshape(b, g1, g1, 16).permute(0, 3, 1, 2)
    assert float((got1 - dz1_exact).abs().max()) <= 2.0 ** -7 * float(dz1_exact.abs().max())  # float32 accumulation order, then one bf16 rounding
    assert torch.allclose(db[32:].double(), dz1_exact.sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2)
    # weight gradients of exactly the tensors the kernels were handed (bf16 values, float32 accumulation)
    x0 = torch.tensor(pkg.expand_codes(codes, c), device=dev).double()
    dz2g = dz2.double().reshape(b, g2, g2, 32).permute(0, 3, 1, 2)
    dw2_want = torch.stack([torch.einsum("boyx,biyx->oi", dz2g, h1i[:, :, dy:dy + g2, dx:dx + g2]) for dy in range(3) for dx in range(3)], dim=-1)
    dw1_want = torch.stack([torch.einsum("boyx,bcyx->oc", got1, x0[:, :, dy:dy + g1, dx:dx + g1]) for dy in range(3) for dx in range(3)], dim=-1)
    dw2_got, dw1_got = dw[:4608].double().reshape(32, 16, 9), dw[4608:].double().reshape(16, 16, 9)
    assert torch.allclose(dw2_got, dw2_want, rtol=1e-4, atol=1e-4 * float(dw2_want.abs().max()))
    assert torch.allclose(dw1_got[:, :c], dw1_want, rtol=1e-4, atol=1e-4 * float(dw1_want.abs().max())) and float(dw1_got[:, c:].abs().max()) == 0.0
    # ---- round 4: the same backward as ONE call (conv2's weight gradient inside the data-gradient pass: dz2 never leaves the CU)
    dz1_f = torch.empty((b, p1, 16), dtype=bf, device=dev)
    db_f = torch.zeros(48, dtype=torch.float32, device=dev)
    dw_f = torch.zeros(32 * 16 * 9 + 16 * 16 * 9, dtype=torch.float32, device=dev)
    assert lib.ctf_policy_front_backward(ptr(d_act), ptr(act), ptr(h1), ptr(codes_t), ptr(f2t), b, g, m, ptr(dz1_f), ptr(dw_f[:4608]), ptr(dw_f[4608:]),
                                         ptr(db_f[:32]), ptr(db_f[32:]), 0, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(dz1_f, dz1)                                                      # the data path is the same arithmetic
    assert torch.allclose(db_f, db, rtol=1e-5, atol=1e-4)
    assert torch.allclose(dw_f[:4608].double().reshape(32, 16, 9), dw2_want, rtol=1e-4, atol=1e-4 * float(dw2_want.abs().max()))
    assert torch.allclose(dw_f[4608:].double().reshape(16, 16, 9)[:, :c], dw1_want, rtol=1e-4, atol=1e-4 * float(dw1_want.abs().max()))


@pytest.mark.parametrize("n_out,n_in,m", [(128, 256, 5000), (16, 128, 777), (128, 256, 64), (16, 128, 131072), (256, 4160, 3000), (256, 2112, 129),
                                          (256, 256, 70000)])
def test_small_layer_weight_and_bias_gradients_match_float64(n_out, n_in, m):
    """ctf_policy_linear_wgrad: dw = dy^T x and db = sum(dy) over the samples, bf16 operands, float32 accumulation (atomics across
    blocks), against the float64 sums of the same bf16 values; added to what the buffers held."""
    import ctypes as C

    abi = importlib.import_module("marl-ctf-development_amd._abi")
    lib = abi.load_library()
    g = torch.Generator(device="cuda").manual_seed(n_out + m)
    dy = (torch.randn((m, n_out), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.tanh(torch.randn((m, n_in), generator=g, device="cuda")).to(torch.bfloat16)
    dw = torch.full((n_out, n_in), 0.5, dtype=torch.float32, device="cuda")
    db = torch.full((n_out,), -2.0, dtype=torch.float32, device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr())
    # (the fc1 form, 256 outputs, carries its bias in a weight column: no separate bias gradient)
    assert lib.ctf_policy_linear_wgrad(ptr(dy), ptr(x), m, n_out, n_in, ptr(dw), ptr(db) if n_out != 256 else None, 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    want_w = dy.double().T @ x.double() + 0.5
    want_b = dy.double().sum(0) - 2.0 if n_out != 256 else db.double()
    scale = float(want_w.abs().max())
    assert float((dw.double() - want_w).abs().max()) < 1e-5 * max(scale, 1.0) * (1 + m / 65536)
    assert float((db.double() - want_b).abs().max()) < 1e-5 * max(float(want_b.abs().max()), 1.0) * (1 + m / 65536)
    assert lib.ctf_policy_linear_wgrad(ptr(dy), ptr(x), m, 64, 64, ptr(dw), ptr(db), 0, None) != 0  # an unsupported shape says so


def test_native_tail_gradients_equal_the_library_paths():
    """CtfPolicyNative.trunk_codes with native_tail_wgrad: outputs identical, gradients of fc2 and both heads (weights and biases) equal
    to the library path's up to bf16 / summation-order effects."""
    import copy

    rng = np.random.default_rng(3)
    g, c, n, b = 15, 14, 8, 3000
    m = 2 * n + 6
    codes = torch.tensor((rng.integers(0, c, (b, g, g)) * (rng.random((b, g, g)) < 0.3)).astype(np.uint8), device="cuda")
    meta = torch.tensor(rng.random((b, m)).astype(np.float32), device="cuda")
    base = fill_(native.CtfPolicyNative(9, c, g, m)).cuda()
    outs = []
    for flag in (True, False):
        net = copy.deepcopy(base)
        net.native_tail_wgrad = flag
        value, logits = net.trunk_codes(codes, meta)
        ((logits * torch.linspace(-1, 1, 9, device="cuda")).sum() + (value ** 2).sum()).backward()
        outs.append((value.detach(), logits.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        a, bgrad = outs[0][2][k].double(), outs[1][2][k].double()
        assert float((a - bgrad).abs().max()) <= 2e-2 * max(float(bgrad.abs().max()), 1e-6), (k, float((a - bgrad).abs().max()), float(bgrad.abs().max()))


@pytest.mark.parametrize("m,kp", [(32, 64), (256, 4160), (3008, 2112), (8192 + 96, 4160)])
def test_fc1_data_gradient_is_the_float64_product_rounded_once(m, kp):
    """ctf_policy_fc1_dgrad: d_act = dy @ W (bf16 operands, float32 accumulation, ONE rounding to bf16) against float64 on the same
    operands — within one bf16 spacing of the exact value everywhere; ragged last block (waves without rows), nothing written beyond."""
    import ctypes as C

    abi = importlib.import_module("marl-ctf-development_amd._abi")
    lib = abi.load_library()
    g = torch.Generator(device="cuda").manual_seed(m + kp)
    dy = (torch.randn((m, 256), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    w = (torch.randn((256, kp), generator=g, device="cuda") * 0.3).to(torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.full((m + 1, kp), 3.0, dtype=torch.bfloat16, device="cuda")  # a guard row behind the matrix
    ptr = lambda t: C.c_void_p(t.data_ptr())
    assert lib.ctf_policy_fc1_dgrad(ptr(dy), ptr(wt), m, kp, ptr(out), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0, \
        lib.ctf_policy_last_error()
    want = dy.double() @ w.double()
    got = out[:m].double()
    spacing = torch.pow(2.0, torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 7)
    assert bool(((got - want).abs() <= spacing * 0.5 * (1 + 1e-6) + 1e-6).all()), float(((got - want).abs() / spacing).max())
    assert float((out[m].float() - 3.0).abs().max()) == 0.0
    lib_out = torch.mm(dy, w).double()  # the library's product of the same operands: same values up to a rounding flip here and there
    assert float(((got - lib_out).abs() > spacing * 1.01).double().mean()) == 0.0
    assert lib.ctf_policy_fc1_dgrad(ptr(dy), ptr(wt), m + 1, kp, ptr(out), 0, None) != 0   # unsupported shapes say so
    assert lib.ctf_policy_fc1_dgrad(ptr(dy), ptr(wt), m, kp + 8, ptr(out), 0, None) != 0


def test_native_fc1_data_gradient_equals_the_library_path():
    """CtfPolicyNative.trunk_codes with native_fc1_dgrad: outputs identical; every parameter's gradient equal to the library path's up to
    bf16 rounding flips of d_act (the conv front's gradients are the ones that see them)."""
    import copy

    rng = np.random.default_rng(4)
    g, c, n, b = 15, 14, 8, 2048
    m = 2 * n + 6
    codes = torch.tensor((rng.integers(0, c, (b, g, g)) * (rng.random((b, g, g)) < 0.3)).astype(np.uint8), device="cuda")
    meta = torch.tensor(rng.random((b, m)).astype(np.float32), device="cuda")
    base = fill_(native.CtfPolicyNative(9, c, g, m)).cuda()
    outs = []
    for flag in (True, False):
        net = copy.deepcopy(base)
        net.native_fc1_dgrad = flag
        value, logits = net.trunk_codes(codes, meta)
        ((logits * torch.linspace(-1, 1, 9, device="cuda")).sum() + (value ** 2).sum()).backward()
        outs.append((value.detach(), logits.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        a, bgrad = outs[0][2][k].double(), outs[1][2][k].double()
        assert float((a - bgrad).abs().max()) <= 1e-2 * max(float(bgrad.abs().max()), 1e-6), (k, float((a - bgrad).abs().max()), float(bgrad.abs().max()))
