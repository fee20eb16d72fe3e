"""Would an observation buffer assembled from 2 MiB chunks of the HIP virtual-memory API (a reliable, intermediate kind for the
bare fill, tools/alloc_probe9.hip) also be a reliable kind for the real render?  Builds such buffers through ctypes on the HIP
runtime, wraps them as torch tensors (__cuda_array_interface__) and times the arena render into them and into ordinary ones."""
import ctypes as C
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("marl-ctf-development_amd")

hip = C.CDLL("libamdhip64.so")

class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p), ("allocFlags", AllocFlags)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]

def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")

class VmmBuffer:
    def __init__(self, nbytes, chunk=2 << 20):
        self.total = (nbytes + chunk - 1) // chunk * chunk   # >= nbytes
        self.nbytes = nbytes
        prop = Prop(); prop.type = 1; prop.location = Loc(1, 0)   # hipMemAllocationTypePinned, hipMemLocationTypeDevice
        va = C.c_void_p()
        ck(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(self.total), C.c_size_t(0), None, C.c_ulonglong(0)), "reserve")
        self.va = va.value
        for off in range(0, self.total, chunk):
            h = C.c_void_p()
            ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), "create")
            ck(hip.hipMemMap(C.c_void_p(self.va + off), C.c_size_t(chunk), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
            ck(hip.hipMemRelease(h), "release")
        acc = Access(Loc(1, 0), 3)
        ck(hip.hipMemSetAccess(C.c_void_p(self.va), C.c_size_t(self.total), C.byref(acc), C.c_size_t(1)), "access")
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.va, False), "version": 2}

kw = bench.WORKLOADS["arena"][1](pkg)
E = 65536
vec = pkg.VecGridworldCtf(E, device=0, tune_placement=False, **kw)
acts = torch.zeros((E, vec.N_AGENTS), dtype=torch.int8, device="cuda")
for s in range(30):
    vec.random_actions(acts, 3, s); vec.step(acts, auto_reset=True)
shape = (E, vec.N_AGENTS, vec.N_CHANNELS, vec.GRID_SIZE, vec.GRID_SIZE)
nbytes = 1
for d in shape: nbytes *= d
def t(buf, reps=6):
    vec.obs = buf
    vec.observe(meta=False)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): vec.observe(meta=False)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
keep = [torch.empty(shape, dtype=torch.uint8, device="cuda") for _ in range(8)]
print("hipMalloc (torch)     :", " ".join(f"{t(b):.3f}" for b in keep), flush=True)
ref = keep[0]; t(ref)
holders = [VmmBuffer(nbytes) for _ in range(6)]
views = [torch.as_tensor(h, device="cuda").view(shape) for h in holders]
print("VMM, 2 MiB chunks     :", " ".join(f"{t(v):.3f}" for v in views), flush=True)
vec.obs = ref; vec.observe()
ok = torch.equal(views[0], ref) if False else True
vec.obs = views[0]; vec.observe()
print("same bytes as an ordinary buffer:", bool(torch.equal(views[0], ref)))
