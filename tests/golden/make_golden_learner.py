"""Golden outputs of the reference's own learner functions — PPOTrainer.calculate_advantages and PPOTrainer.optimise
(ppo.py:133-242) — on the reference's own Agent (formula weights) and reference observations (policy_split.npz):
the pin of marl-ctf-development_amd/learner.py.  Build container only:  python tests/golden/make_golden_learner.py"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402
from _policy_weights import fill_  # noqa: E402

ARGS = dict(gae=True, gamma=0.99, gae_lambda=0.95, update_epochs=2, num_minibatches=4, clip_coef=0.2, norm_adv=True, clip_vloss=True,
            ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, target_kl=None, learning_rate=2.5e-4)


def main():
    import torch

    _refimport.import_reference()
    mods = _refimport.import_reference.modules
    saved = list(sys.path)
    sys.path.insert(0, _refimport.REFERENCE_DIR)
    sys.modules.update(mods)
    try:
        import agent_network as ref_net
        import ppo as ref_ppo
    finally:
        sys.path[:] = saved
        for m in ("gridworld_ctf", "scenarios", "utils", "ppo", "agent_network"):
            sys.modules.pop(m, None)

    z = np.load(os.path.join(HERE, "policy_split.npz"))
    shape = tuple(int(x) for x in z["grid_shape"])
    grids = np.unpackbits(z["grids"])[: int(np.prod(shape))].reshape(shape).astype(np.float32)  # [48, C, G, G]
    metas = z["metas"].view(np.float16).astype(np.float32)
    masks = z["masks"].astype(np.float32)
    S, E = 22, 2                                   # 44 slots of the 48 samples as a [S, E] rollout, the last 2 samples = next state
    c, g, m = shape[1], shape[2], metas.shape[1]
    torch.manual_seed(0)
    net = fill_(ref_net.Agent(9, c, g, m))
    G = torch.tensor(grids[:S * E]).reshape(S, E, c, g, g)
    M = torch.tensor(metas[:S * E]).reshape(S, E, m)
    K = torch.tensor(masks[:S * E]).reshape(S, E)
    with torch.no_grad():
        logits = net(G.reshape(-1, c, g, g), M.reshape(-1, m))[1]
        actions = (logits.argmax(dim=1) % 5).float()
        _, logprobs, _, values = net.get_action_and_value(G.reshape(-1, c, g, g), M.reshape(-1, m), K.reshape(-1), actions.long())
    actions, logprobs, values = actions.reshape(S, E), (logprobs.reshape(S, E) - 0.05), values.reshape(S, E)  # old policy slightly off
    idx = torch.arange(S * E, dtype=torch.float32).reshape(S, E)
    rewards = torch.sin(idx * 0.7) * 0.5 + (idx % 7 == 0).float()
    dones = torch.zeros((S, E))
    next_done = torch.tensor([0.0, 1.0])
    next_grid, next_meta = torch.tensor(grids[S * E:S * E + E]), torch.tensor(metas[S * E:S * E + E])

    args = types.SimpleNamespace(**ARGS)
    tr = ref_ppo.PPOTrainer(args, (c, g, g), (m,))
    tr.device, tr.num_steps = "cpu", S
    adv, ret = tr.calculate_advantages(net, next_grid, next_meta, rewards, next_done, dones, values)
    tr.batch_size, tr.minibatch_size = S * E, S * E // args.num_minibatches
    tr.optimizer = torch.optim.Adam(net.parameters(), lr=args.learning_rate, eps=1e-5)
    np.random.seed(7)
    flat = lambda t: t.reshape((-1,) + tuple(t.shape[2:]))
    losses = tr.optimise(net, flat(G), flat(M), flat(logprobs), flat(actions), flat(K), flat(adv), flat(ret), flat(values))
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "learner_ref.npz"), S=np.array(S), E=np.array(E), actions=actions.numpy(), logprobs=logprobs.numpy(),
                        values=values.numpy(), rewards=rewards.numpy(), next_done=next_done.numpy(), advantages=adv.numpy(),
                        returns=ret.numpy(), losses=np.array(losses, np.float64), action_head_w=sd["action_head.weight"],
                        value_head_w=sd["value_head.weight"], conv1_b=sd["conv1.bias"], fc2_b=sd["fc2.bias"],
                        conv2_w_sum=np.array(float(sd["conv2.weight"].astype(np.float64).sum())))
    print("losses", losses, "adv range", float(adv.min()), float(adv.max()))


if __name__ == "__main__":
    main()
