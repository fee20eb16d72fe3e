"""Golden outputs of the reference's own policy network (agent_network.py) on observations of the reference env,
with formula weights (tests/_policy_weights.py).  Build container only:  python tests/golden/make_golden_policy.py"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import _refimport  # noqa: E402
import make_golden as mg  # noqa: E402
from _policy_weights import fill_  # noqa: E402


def main():
    import torch

    Ref, scn = _refimport.import_reference()
    mods = _refimport.import_reference.modules
    saved = list(sys.path)
    sys.path.insert(0, _refimport.REFERENCE_DIR)
    try:
        import agent_network as ref_net
    finally:
        sys.path[:] = saved
        sys.modules.pop("agent_network", None)
    for name, scenario, kw in (("policy_arena", "arena_iii", mg.ARENA_KW), ("policy_split", "arrow", mg.SPLIT_KW)):
        random.seed(5)
        np.random.seed(5)
        env = Ref(SCENARIO=getattr(scn, scenario), **kw)
        dims = env.get_env_dims()
        rng = np.random.default_rng(3)
        grids, metas, masks = [], [], []
        for t in range(12):
            env.step([int(a) for a in rng.integers(0, 9, env.N_AGENTS)])
            for i in range(env.N_AGENTS):
                grids.append(env.standardise_state(i, reverse_grid=env.AGENT_TEAMS[i] == 1)[0])
                metas.append(env.get_env_metadata(i)[0])
                masks.append(env.AGENT_TYPE_ACTION_MASK[env.AGENT_TYPES[i]])
        grids, metas, masks = np.stack(grids), np.stack(metas), np.array(masks, np.float32)
        net = fill_(ref_net.Agent(9, dims[0][0], env.GRID_SIZE, dims[2][0]))
        with torch.no_grad():
            g, m, k = torch.tensor(grids, dtype=torch.float32), torch.tensor(metas, dtype=torch.float32), torch.tensor(masks)
            value, logits = net(g, m)
            actions = logits.argmax(dim=1)
            _, logprob, entropy, value2 = net.get_action_and_value(g, m, k, action=actions % 5)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), grids=np.packbits(grids.reshape(-1)), grid_shape=np.array(grids.shape),
                            metas=metas.view(np.uint16), masks=masks, value=value.numpy(), logits=logits.numpy(),
                            actions=(actions % 5).numpy(), logprob=logprob.numpy(), entropy=entropy.numpy(),
                            n_agents=np.array(env.N_AGENTS))
        print(name, "samples", len(grids), "value range", float(value.min()), float(value.max()), "per-logit std over samples",
              logits.std(dim=0).numpy().round(3))


if __name__ == "__main__":
    main()
