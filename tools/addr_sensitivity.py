"""Is k_observe's time sensitive to where the output buffer lives?  Times the same launch into differently placed buffers."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("marl-ctf-development_amd")
E = 65536
kw = dict(pkg.configs.ARENA_KWARGS, SCENARIO=pkg.CtfScenarios.arena_iii)
vec = pkg.VecGridworldCtf(E, device=0, **kw)
acts = torch.zeros((E, 8), dtype=torch.int8, device="cuda")
n = vec.obs.numel()
big = torch.empty(n + (64 << 20), dtype=torch.uint8, device="cuda")
def timeit(buf, reps=60):
    vec.obs = buf
    for _ in range(10):
        vec.step(acts, auto_reset=True); vec.observe()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        vec.step(acts, auto_reset=True); a.record(); vec.observe(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
orig = vec.obs
print("original buffer ptr %% 2MiB = %d" % (orig.data_ptr() % (2 << 20)), "%.4f ms" % timeit(orig))
for off in (0, 16, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20):
    buf = big[off:off + n].view(orig.shape)
    print("offset %9d  ptr %% 2MiB = %8d  %.4f ms" % (off, buf.data_ptr() % (2 << 20), timeit(buf)))
fresh = [torch.empty_like(orig) for _ in range(3)]
for f in fresh:
    print("fresh buffer ptr %% 2MiB = %d  %.4f ms" % (f.data_ptr() % (2 << 20), timeit(f)))
print("original again %.4f ms" % timeit(orig))
