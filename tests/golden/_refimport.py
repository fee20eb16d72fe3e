"""Import the reference env in the BUILD container (never on the GPU box: /root/reference is absent there).

Only the fixture generators (make_golden.py, make_maps.py) and the optional live cross-check tests use
this.  The reference needs IPython / seaborn / imageio / wandb / ray at import time for rendering and
orchestration only; inert stubs stand in for them.  Nothing of the reference is copied.
"""
import os
import sys
import types

REFERENCE_DIR = "/root/reference"


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "gridworld_ctf.py"))


def import_reference():
    """-> (GridworldCtf, CtfScenarios) from /root/reference.  Changes cwd (the ctor opens cwd/img/*.png)."""
    if not available():
        raise RuntimeError("reference not present")
    for name in ("IPython", "IPython.display", "seaborn", "imageio", "wandb", "ray"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["IPython.display"].clear_output = lambda *a, **k: None
    sys.modules["IPython"].display = sys.modules["IPython.display"]
    ray = sys.modules["ray"]
    ray.remote = lambda f: f
    ray.init = ray.shutdown = ray.get = ray.put = lambda *a, **k: None
    import matplotlib

    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    os.chdir(REFERENCE_DIR)
    saved = list(sys.path)
    # the product package also ships modules called gridworld_ctf / scenarios-like names: make sure the
    # reference's own files win for this import, then restore the path
    sys.path.insert(0, REFERENCE_DIR)
    for mod in ("gridworld_ctf", "scenarios", "utils"):
        sys.modules.pop(mod, None)
    try:
        from gridworld_ctf import GridworldCtf
        from scenarios import CtfScenarios
    finally:
        sys.path[:] = saved
    ref_mods = {m: sys.modules.pop(m) for m in ("gridworld_ctf", "scenarios", "utils") if m in sys.modules}
    import_reference.modules = ref_mods
    return GridworldCtf, CtfScenarios
