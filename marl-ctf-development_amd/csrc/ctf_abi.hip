// ctf_abi.hip — host side of the C ABI declared in include/ctf_env.h.
//
// Owns the device-side SoA state of one handle (one per GPU / shard), validates the config, derives
// the device constant block and enqueues the kernels of ctf_kernels.hip on the caller's HIP stream.
// No torch, no C++ types across the boundary, no CPU implementation of the path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctf_device.h"

extern "C" hipError_t ctf_launch_seed(const DevCfg&, const DevPtrs&, const uint64_t*, const uint64_t*, hipStream_t);
extern "C" hipError_t ctf_launch_reset(const DevCfg&, const DevPtrs&, const uint8_t*, int, hipStream_t);
extern "C" hipError_t ctf_launch_step(const DevCfg&, const DevPtrs&, const int8_t*, float*, double*, uint8_t*, uint32_t, int, hipStream_t);
extern "C" int ctf_step_blocks(const DevCfg&);
extern "C" int ctf_observe_uses_tiles(const DevCfg&, const uint8_t*);
extern "C" hipError_t ctf_launch_observe(const DevCfg&, const DevPtrs&, uint8_t*, uint16_t*, uint32_t, int, hipStream_t);
extern "C" hipError_t ctf_launch_observe_codes(const DevCfg&, const DevPtrs&, uint8_t*, uint16_t*, uint16_t*, uint32_t, int, hipStream_t);
extern "C" hipError_t ctf_launch_random_actions(const DevCfg&, int8_t*, uint64_t, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t ctf_launch_import_rng(const DevCfg&, const DevPtrs&, const uint32_t*, const uint32_t*, int, int, hipStream_t);
extern "C" hipError_t ctf_launch_export_rng(const DevCfg&, const DevPtrs&, uint32_t*, uint32_t*, int, int, hipStream_t);
extern "C" hipError_t ctf_launch_rng_refill(const DevCfg&, const DevPtrs&, int, int, int, hipStream_t);
extern "C" hipError_t ctf_launch_get_counters(const DevCfg&, const DevPtrs&, unsigned long long*, hipStream_t);
extern "C" hipError_t ctf_launch_set_counters(const DevCfg&, const DevPtrs&, const unsigned long long*, hipStream_t);
extern "C" hipError_t ctf_launch_export_counters(const DevCfg&, const DevPtrs&, int32_t*, int32_t*, int32_t*, hipStream_t);

struct ctf_env {
    ctf_config cfg;
    DevCfg d;
    DevPtrs p;
    int device;
    int n_cus;
    uint64_t* seed_scratch;  // device, 2*E u64
    uint32_t* rng_scratch;   // device, 2 x 625 u32: one env's two generators in the standard form (ctf_set/get_rng_state)
    // ctf_host_step (n_envs == 1): one pinned, device-mapped host block (allocated on first use) that the kernels read and write directly
    uint8_t* hio_dev;   // the DEVICE address of that block (hipHostGetDevicePointer)
    uint8_t* hio_host;  // its host address
    int nt_override;    // CTF_OBS_NT at create: 0 / 1 force the render's store hint off / on, -1 = the rule (store_hint)
};

// Layout of the ctf_host_step block (byte offsets; every segment 16-byte aligned, the observation 256-byte aligned).
struct HostIo {
    size_t actions, py_in, np_in, in_end;                               // host -> device
    size_t out, rw64, done_status, py_out, np_out, obs, meta, grid, rec, metrics, vis, vislog, end;  // device -> host
};
static size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static HostIo host_io_layout(const DevCfg& d) {
    HostIo L;
    const size_t mt = up((CTF_MT_N + 1) * 4, 16);
    L.actions = 0;
    L.py_in = 16;
    L.np_in = L.py_in + mt;
    L.in_end = L.np_in + mt;
    L.out = up(L.in_end, 256);
    L.rw64 = L.out;
    L.done_status = L.rw64 + CTF_MAX_AGENTS * 8;
    L.py_out = L.done_status + 16;
    L.np_out = L.py_out + mt;
    L.obs = up(L.np_out + mt, 256);
    L.meta = L.obs + up((size_t)d.obs_bytes, 16);
    L.grid = L.meta + up((size_t)d.N * d.M * 2, 16);
    L.rec = L.grid + (size_t)d.GS;
    L.metrics = L.rec + (size_t)d.RS;
    L.vis = L.metrics + up((size_t)CTF_N_METRICS * d.N * 4, 16);
    L.vislog = L.vis + (d.log_metrics ? (size_t)d.N * d.GS * 4 : 0);
    L.end = L.vislog + (d.log_metrics ? (size_t)CTF_VIS_LOG * d.N * 2 : 0);
    return L;
}

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return fail(CTF_E_HIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// The render's nontemporal store hint (ctf_derive.h has the why and the measured crossover) follows what ALL live handles of this
// process on the device write per step, not one handle's share: four shards of 8 192 arena envs (206 MB each) stored plain read
// 183 M env-steps/s, hinted 218 M — each fits the caches, the four together do not (profiles/r05_two_shards_overlap.md).  Other
// PROCESSES on the device are not seen: CTF_OBS_NT=1 is for them.
#define CTF_MAX_DEVICES 64
static std::atomic<long long> g_obs_bytes[CTF_MAX_DEVICES];
static long long obs_total(const ctf_env* h) { return (long long)h->d.n_envs * h->d.obs_bytes; }
static int store_hint(const ctf_env* h) {
    if (h->nt_override >= 0) return h->nt_override;
    const long long all = h->device >= 0 && h->device < CTF_MAX_DEVICES ? g_obs_bytes[h->device].load(std::memory_order_relaxed) : obs_total(h);
    return all > ((long long)320 << 20);
}

// remembers and restores the caller's current device
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

#include "ctf_derive.h"

static void free_all(ctf_env* h) {
    if (!h) return;
    (void)hipFree(h->p.grid); (void)hipFree(h->p.rec); (void)hipFree(h->p.mt_py); (void)hipFree(h->p.mt_np);
    (void)hipFree(h->p.rngpos); (void)hipFree(h->p.metrics); (void)hipFree(h->p.vis); (void)hipFree(h->p.vislog);
    (void)hipFree((void*)h->p.init_grid); (void)hipFree((void*)h->p.meta_lut); (void)hipFree(h->p.status); (void)hipFree(h->seed_scratch);
    (void)hipFree(h->p.rngctr); (void)hipFree(h->rng_scratch); (void)hipFree(h->p.rngready); (void)hipFree(h->p.rngage);
    (void)hipFree(h->p.py_top); (void)hipFree(h->p.np_hit); (void)hipFree(h->p.np_nib);
    if (h->hio_host) (void)hipHostFree(h->hio_host);
    delete h;
}

extern "C" int ctf_create(const ctf_config* cfg, int32_t n_envs, int32_t device_id, ctf_env** out) {
    if (!cfg || !out) return fail(CTF_E_INVALID, "null argument");
    *out = nullptr;
    DevCfg d;
    int rc = derive(cfg, n_envs, &d);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(CTF_E_INVALID, "device %d of %d", device_id, ndev);
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(CTF_E_HIP, "hipSetDevice(%d) failed", device_id);
    ctf_env* h = new (std::nothrow) ctf_env();
    if (!h) return fail(CTF_E_NOMEM, "host allocation failed");
    memset(&h->p, 0, sizeof(h->p));
    h->seed_scratch = nullptr;
    h->rng_scratch = nullptr;
    h->hio_dev = nullptr;
    h->hio_host = nullptr;
    {
        const char* ov = getenv("CTF_OBS_NT");
        h->nt_override = ov ? (atoi(ov) != 0) : -1;
    }
    h->cfg = *cfg; h->d = d; h->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { free_all(h); return fail(CTF_E_HIP, "hipGetDeviceProperties failed"); }
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->d.n_cus = d.n_cus = h->n_cus;
    const size_t E = (size_t)n_envs;
    const size_t vis_elems = d.log_metrics ? E * d.N * d.GS : 1, met_elems = d.log_metrics ? E * CTF_N_METRICS * d.N : 1;
#define ALLOC(ptr, bytes)                                                                                      \
    if (hipMalloc((void**)&(ptr), (bytes)) != hipSuccess) { free_all(h); return fail(CTF_E_NOMEM, "hipMalloc(%zu) failed", (size_t)(bytes)); }
    ALLOC(h->p.grid, E * d.GS);
    ALLOC(h->p.rec, E * d.RS);
    ALLOC(h->p.mt_py, E * 2 * CTF_MT_N * 4);
    ALLOC(h->p.mt_np, E * 2 * CTF_MT_N * 4);
    ALLOC(h->p.py_top, E * 2 * CTF_P8_DW * 4);
    ALLOC(h->p.np_hit, E * 2 * CTF_HB_DW * 4);
    ALLOC(h->p.np_nib, E * 2 * CTF_NB_DW * 4);
    ALLOC(h->p.rngpos, E * 2 * 4);
    ALLOC(h->p.rngready, E * 2);
    ALLOC(h->p.rngage, E * 2);
    ALLOC(h->p.rngctr, (d.rng_mode == CTF_RNG_COUNTER ? E * 6 : 1) * 8);
    ALLOC(h->rng_scratch, 2 * (CTF_MT_N + 1) * 4);
    ALLOC(h->p.metrics, met_elems * 4);
    ALLOC(h->p.vis, vis_elems * 4);
    ALLOC(h->p.vislog, (d.log_metrics ? (size_t)CTF_VIS_LOG * E * d.N : 1) * 2);
    ALLOC(h->p.init_grid, (size_t)d.GS);
    ALLOC(h->p.meta_lut, (size_t)round_up(d.N * d.M, 16));
    ALLOC(h->p.status, 4);
    ALLOC(h->seed_scratch, E * 2 * 8);
#undef ALLOC
    std::vector<uint8_t> g0((size_t)d.GS, 0);
    memcpy(g0.data(), cfg->init_grid, (size_t)d.GG);
    hipError_t e1 = hipMemcpy((void*)h->p.init_grid, g0.data(), (size_t)d.GS, hipMemcpyHostToDevice);
    std::vector<uint8_t> lut((size_t)round_up(d.N * d.M, 16), 41);
    host_meta_lut(d, lut.data());
    if (hipMemcpy((void*)h->p.meta_lut, lut.data(), lut.size(), hipMemcpyHostToDevice) != hipSuccess) { free_all(h); return fail(CTF_E_HIP, "device initialisation failed"); }
    hipError_t e2 = hipMemset(h->p.status, 0, 4);
    hipError_t e3 = hipMemset(h->p.rec, 0, E * d.RS);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { free_all(h); return fail(CTF_E_HIP, "device initialisation failed"); }
    // first reset (+ _arr = [0..N-1], gridworld_ctf.py:244) and seeds 0/0
    hipError_t e4 = ctf_launch_reset(h->d, h->p, nullptr, 1, nullptr);
    hipError_t e5 = hipMemset(h->seed_scratch, 0, E * 2 * 8);
    hipError_t e6 = ctf_launch_seed(h->d, h->p, h->seed_scratch, h->seed_scratch + E, nullptr);
    if (e6 == hipSuccess) e6 = ctf_launch_rng_refill(h->d, h->p, 0, n_envs, 1, nullptr);
    if (e6 == hipSuccess) e6 = hipMemset(h->p.rngage, 0, E * 2);  // (a seed / state import leaves every ring in place: the ages go back to 0 by themselves, tail_block)
    hipError_t e7 = hipDeviceSynchronize();
    if (e4 != hipSuccess || e5 != hipSuccess || e6 != hipSuccess || e7 != hipSuccess) {
        const hipError_t bad = e4 != hipSuccess ? e4 : e5 != hipSuccess ? e5 : e6 != hipSuccess ? e6 : e7;
        free_all(h);
        return fail(CTF_E_HIP, "first reset/seed failed: %s", hipGetErrorString(bad));
    }
    if (device_id >= 0 && device_id < CTF_MAX_DEVICES) g_obs_bytes[device_id].fetch_add(obs_total(h), std::memory_order_relaxed);
    *out = h;
    return CTF_OK;
}

extern "C" void ctf_destroy(ctf_env* h) {
    if (!h) return;
    if (h->device >= 0 && h->device < CTF_MAX_DEVICES) g_obs_bytes[h->device].fetch_sub(obs_total(h), std::memory_order_relaxed);
    DeviceGuard guard(h->device);
    (void)hipDeviceSynchronize();
    free_all(h);
}

extern "C" int32_t ctf_n_envs(const ctf_env* h) { return h ? h->d.n_envs : 0; }
extern "C" int64_t ctf_obs_bytes_per_env(const ctf_env* h) { return h ? h->d.obs_bytes : 0; }
extern "C" int64_t ctf_meta_elems_per_env(const ctf_env* h) { return h ? (int64_t)h->d.N * h->d.M : 0; }
extern "C" const char* ctf_last_error(void) { return g_err; }
extern "C" int32_t ctf_abi_version(void) { return CTF_ABI_VERSION; }
extern "C" int32_t ctf_sizeof_config(void) { return (int32_t)sizeof(ctf_config); }
extern "C" int32_t ctf_sizeof_state_view(void) { return (int32_t)sizeof(ctf_state_view); }

extern "C" int ctf_seed(ctf_env* h, const uint64_t* py_seeds, const uint64_t* np_seeds, void* stream) {
    if (!h || !py_seeds || !np_seeds) return fail(CTF_E_INVALID, "null argument");
    DeviceGuard guard(h->device);
    const size_t E = (size_t)h->d.n_envs;
    if (h->d.rng_mode == CTF_RNG_MT19937)
        for (size_t e = 0; e < E; e++)
            if (np_seeds[e] >> 32) return fail(CTF_E_INVALID, "np seed %zu must be < 2**32 (np.random.seed's own limit)", e);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->seed_scratch, py_seeds, E * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->seed_scratch + E, np_seeds, E * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctf_launch_seed(h->d, h->p, h->seed_scratch, h->seed_scratch + E, st));
    HIP_TRY(ctf_launch_rng_refill(h->d, h->p, 0, h->d.n_envs, 1, st));  // the blocks after the seeded ones, and all digests
    HIP_TRY(hipStreamSynchronize(st));  // the host arrays are the caller's; do not outlive the call
    return CTF_OK;
}

static int need_mode(const ctf_env* h, int mode, const char* what) {
    if (h->d.rng_mode != mode)
        return fail(CTF_E_INVALID, "%s: the handle runs in %s mode", what, h->d.rng_mode == CTF_RNG_COUNTER ? "counter-RNG" : "MT19937");
    return CTF_OK;
}

extern "C" int ctf_set_rng_state(ctf_env* h, int32_t e, const uint32_t* py, const uint32_t* np_) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (e < 0 || e >= h->d.n_envs) return fail(CTF_E_RANGE, "env index %d", e);
    if (int rc = need_mode(h, CTF_RNG_MT19937, "ctf_set_rng_state")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
    const uint32_t* src[2] = {py, np_};
    uint32_t* dev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; k++) {
        if (!src[k]) continue;
        if (src[k][CTF_MT_N] > CTF_MT_N) return fail(CTF_E_INVALID, "MT position %u > 624", src[k][CTF_MT_N]);
        dev[k] = h->rng_scratch + k * (CTF_MT_N + 1);
        HIP_TRY(hipMemcpy(dev[k], src[k], (CTF_MT_N + 1) * 4, hipMemcpyHostToDevice));
    }
    if (dev[0] || dev[1]) {
        // the two records are not adjacent when only one is given: one launch per generator
        if (dev[0]) HIP_TRY(ctf_launch_import_rng(h->d, h->p, dev[0], nullptr, e, 1, nullptr));
        if (dev[1]) HIP_TRY(ctf_launch_import_rng(h->d, h->p, nullptr, dev[1], e, 1, nullptr));
        // every ring of every env is brought up to date (envs other than e: whatever the last step left for the next launch's tail)
        HIP_TRY(ctf_launch_rng_refill(h->d, h->p, 0, h->d.n_envs, 0, nullptr));
        HIP_TRY(ctf_launch_rng_refill(h->d, h->p, e, 1, 1, nullptr));  // (a stream of env e that was not handed over is simply redone)
        HIP_TRY(hipDeviceSynchronize());
    }
    return CTF_OK;
}

extern "C" int ctf_get_rng_state(ctf_env* h, int32_t e, uint32_t* py, uint32_t* np_) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (e < 0 || e >= h->d.n_envs) return fail(CTF_E_RANGE, "env index %d", e);
    if (int rc = need_mode(h, CTF_RNG_MT19937, "ctf_get_rng_state")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
    uint32_t* dst[2] = {py, np_};
    for (int k = 0; k < 2; k++) {
        if (!dst[k]) continue;
        uint32_t* dev = h->rng_scratch + k * (CTF_MT_N + 1);
        HIP_TRY(ctf_launch_export_rng(h->d, h->p, k == 0 ? dev : nullptr, k == 1 ? dev : nullptr, e, 1, nullptr));
        HIP_TRY(hipMemcpy(dst[k], dev, (CTF_MT_N + 1) * 4, hipMemcpyDeviceToHost));
    }
    return CTF_OK;
}

extern "C" int ctf_set_rng_states(ctf_env* h, const uint32_t* py_dev, const uint32_t* np_dev, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (!py_dev && !np_dev) return CTF_OK;
    if (int rc = need_mode(h, CTF_RNG_MT19937, "ctf_set_rng_states")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_import_rng(h->d, h->p, py_dev, np_dev, 0, h->d.n_envs, (hipStream_t)stream));
    HIP_TRY(ctf_launch_rng_refill(h->d, h->p, 0, h->d.n_envs, 1, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_get_rng_states(ctf_env* h, uint32_t* py_dev, uint32_t* np_dev, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (!py_dev && !np_dev) return CTF_OK;
    if (int rc = need_mode(h, CTF_RNG_MT19937, "ctf_get_rng_states")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_export_rng(h->d, h->p, py_dev, np_dev, 0, h->d.n_envs, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_get_rng_counters(ctf_env* h, uint64_t* counters_dev, void* stream) {
    if (!h || !counters_dev) return fail(CTF_E_INVALID, "null argument");
    if (int rc = need_mode(h, CTF_RNG_COUNTER, "ctf_get_rng_counters")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_get_counters(h->d, h->p, (unsigned long long*)counters_dev, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_set_rng_counters(ctf_env* h, const uint64_t* counters_dev, void* stream) {
    if (!h || !counters_dev) return fail(CTF_E_INVALID, "null argument");
    if (int rc = need_mode(h, CTF_RNG_COUNTER, "ctf_set_rng_counters")) return rc;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_set_counters(h->d, h->p, (const unsigned long long*)counters_dev, (hipStream_t)stream));
    HIP_TRY(ctf_launch_rng_refill(h->d, h->p, 0, h->d.n_envs, 1, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_reset(ctf_env* h, const uint8_t* mask_dev, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_reset(h->d, h->p, mask_dev, 0, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_step(ctf_env* h, const int8_t* actions, float* rw32, double* rw64, uint8_t* done, uint32_t flags, void* stream) {
    if (!h || !actions) return fail(CTF_E_INVALID, "null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_step(h->d, h->p, actions, rw32, rw64, done, flags, 1, (hipStream_t)stream));
    return CTF_OK;
}

static uint32_t resolve_reverse(const ctf_env* h, uint32_t reverse_mask) {
    return reverse_mask == CTF_REVERSE_DEFAULT ? (uint32_t)h->d.default_reverse : (reverse_mask & ((1u << h->d.N) - 1u));
}

extern "C" int ctf_observe(ctf_env* h, uint8_t* obs, uint16_t* meta, uint32_t reverse_mask, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (!obs && !meta) return CTF_OK;
    DeviceGuard guard(h->device);
    h->d.obs_store_nt = store_hint(h);
    HIP_TRY(ctf_launch_observe(h->d, h->p, obs, meta, resolve_reverse(h, reverse_mask), h->n_cus, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int32_t ctf_observe_kernel(const ctf_env* h, const uint8_t* obs) {
    if (!h) return -1;
    return ctf_observe_uses_tiles(h->d, obs) ? 1 : 0;
}

extern "C" int32_t ctf_observe_stores_hinted(const ctf_env* h, const uint8_t* obs) {
    if (!h) return -1;
    return ctf_observe_uses_tiles(h->d, obs) && store_hint(h) ? 1 : 0;
}

extern "C" int ctf_observe_codes(ctf_env* h, uint8_t* codes, uint16_t* meta, uint16_t* selfcells, uint32_t reverse_mask, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (!codes && !meta && !selfcells) return CTF_OK;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_observe_codes(h->d, h->p, codes, meta, selfcells, resolve_reverse(h, reverse_mask), h->n_cus, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_step_observe(ctf_env* h, const int8_t* actions, float* rw32, double* rw64, uint8_t* done, uint8_t* obs,
                                uint16_t* meta, uint32_t reverse_mask, uint32_t flags, void* stream) {
    if (!h || !actions) return fail(CTF_E_INVALID, "null argument");
    DeviceGuard guard(h->device);
    // (The ring regeneration rides at the tail of the step launch.  Running it as a launch of its own on a second stream, beside
    // the render, was built and measured in round 3: the render lost more than the step kernel gained — 189-191 M against 198 M
    // env-steps/s, profiles/r03_side_stream_ablation.md.)
    HIP_TRY(ctf_launch_step(h->d, h->p, actions, rw32, rw64, done, flags, 1, (hipStream_t)stream));
    if (obs || meta) {
        h->d.obs_store_nt = store_hint(h);
        HIP_TRY(ctf_launch_observe(h->d, h->p, obs, meta, resolve_reverse(h, reverse_mask), h->n_cus, (hipStream_t)stream));
    }
    return CTF_OK;
}

extern "C" int ctf_action_mask(const ctf_env* h, uint8_t* mask_host) {
    if (!h || !mask_host) return fail(CTF_E_INVALID, "null argument");
    static const int type_flag[4] = {1, 1, 0, 0};  // AGENT_TYPE_ACTION_MASK, gridworld_ctf.py:218-223
    for (int i = 0; i < h->d.N; i++)
        for (int a = 0; a < CTF_N_ACTIONS; a++) mask_host[i * CTF_N_ACTIONS + a] = (type_flag[h->d.type[i]] && a >= 5) ? 0 : 1;
    return CTF_OK;
}

// rec / grid / metrics bytes of one env -> the scalar part of its host view
static void decode_record(const DevCfg& d, const uint8_t* grid, const uint8_t* rec, const int32_t* metrics, ctf_state_view* out, int32_t misc[4]) {
    memset(out, 0, sizeof(*out));
    memcpy(out->grid, grid, (size_t)d.GG);
    for (int i = 0; i < d.N; i++) {
        memcpy(&out->hp[i], rec + 8 * i, 8);
        out->pos[i][0] = (int8_t)rec[d.off_pos + 2 * i];
        out->pos[i][1] = (int8_t)rec[d.off_pos + 2 * i + 1];
        out->has_flag[i] = rec[d.off_flag + i];
        out->perm[i] = rec[d.off_perm + i];
        int16_t inv;
        memcpy(&inv, rec + d.off_inv + 2 * i, 2);
        out->inventory[i] = inv;
    }
    memcpy(misc, rec + d.off_misc, 16);
    out->step_count = misc[0];
    out->team_captures[0] = misc[1];
    out->team_captures[1] = misc[2];
    out->done = (misc[3] & CTF_F_DONE) ? 1 : 0;
    if (metrics)
        for (int k = 0; k < CTF_N_METRICS; k++)
            for (int i = 0; i < d.N; i++) out->metrics[k][i] = metrics[(size_t)k * d.N + i];
}

// visitation maps = base maps (or zeros + 1 at the start cells while nothing has been folded) + the `count` log entries
// of steps (folded, step_count] (entries[r][i] = agent i's cell after step folded + 1 + r); u8 wrap as in the reference
static void decode_visitation(const DevCfg& d, const int32_t misc[4], std::vector<uint32_t>& v, const uint16_t* entries, int count,
                              ctf_state_view* out) {
    if (misc[3] & CTF_F_BASE_ZERO) {
        std::fill(v.begin(), v.end(), 0u);
        for (int i = 0; i < d.N; i++) v[(size_t)i * d.GS + d.start_pos[i][0] * d.G + d.start_pos[i][1]] = 1;  // reset(): :473
    }
    for (int r = 0; r < count; r++)
        for (int i = 0; i < d.N; i++) {
            const uint16_t cell = entries[(size_t)r * d.N + i];
            if (cell < (uint16_t)d.GG) v[(size_t)i * d.GS + cell]++;
        }
    for (int i = 0; i < d.N; i++)
        for (int k = 0; k < d.GG; k++) out->visitation[i][k] = (uint8_t)(v[(size_t)i * d.GS + k] & 0xFFu);
}

extern "C" int ctf_get_state(ctf_env* h, int32_t e, ctf_state_view* out) {
    if (!h || !out) return fail(CTF_E_INVALID, "null argument");
    if (e < 0 || e >= h->d.n_envs) return fail(CTF_E_RANGE, "env index %d", e);
    DeviceGuard guard(h->device);
    const DevCfg& d = h->d;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint8_t> rec((size_t)d.RS), grid((size_t)d.GS);
    HIP_TRY(hipMemcpy(grid.data(), h->p.grid + (size_t)e * d.GS, (size_t)d.GS, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rec.data(), h->p.rec + (size_t)e * d.RS, (size_t)d.RS, hipMemcpyDeviceToHost));
    std::vector<int32_t> m((size_t)CTF_N_METRICS * d.N);
    if (d.log_metrics)
        HIP_TRY(hipMemcpy(m.data(), h->p.metrics + (size_t)e * CTF_N_METRICS * d.N, m.size() * 4, hipMemcpyDeviceToHost));
    int32_t misc[4];
    decode_record(d, grid.data(), rec.data(), d.log_metrics ? m.data() : nullptr, out, misc);
    if (d.log_metrics) {
        std::vector<uint32_t> v((size_t)d.N * d.GS, 0);
        if (!(misc[3] & CTF_F_BASE_ZERO))
            HIP_TRY(hipMemcpy(v.data(), h->p.vis + (size_t)e * d.N * d.GS, v.size() * 4, hipMemcpyDeviceToHost));
        const int folded = misc[3] >> CTF_F_FOLDED_SHIFT;
        const int count = misc[0] - folded;  // <= CTF_VIS_LOG - 1 entries, slots (folded+1 .. step) mod 512
        std::vector<uint16_t> entries((size_t)(count > 0 ? count : 0) * d.N);
        if (count > 0) {
            const size_t pitch = (size_t)d.n_envs * d.N * 2, width = (size_t)d.N * 2;
            int done_rows = 0;
            while (done_rows < count) {  // at most two runs: the ring may wrap
                const int slot = (folded + 1 + done_rows) & (CTF_VIS_LOG - 1);
                const int rows = (count - done_rows) < (CTF_VIS_LOG - slot) ? (count - done_rows) : (CTF_VIS_LOG - slot);
                HIP_TRY(hipMemcpy2D(entries.data() + (size_t)done_rows * d.N, width,
                                    h->p.vislog + ((size_t)slot * d.n_envs + e) * d.N, pitch, width, (size_t)rows,
                                    hipMemcpyDeviceToHost));
                done_rows += rows;
            }
        }
        decode_visitation(d, misc, v, entries.data(), count > 0 ? count : 0, out);
    }
    return CTF_OK;
}

// ---- ctf_host_step: one env, host memory on both sides ------------------------------------------------------------------------
// Gathers what a host view of env 0 needs (grid, record, counters, visitation base maps and log) next to the step's outputs, so
// that ONE device-to-host copy brings everything back; reads and clears the sticky status word.
// py_out / np_out: also the two generators in the standard form (624 words of the current ring + the position: what k_export_rng
// writes), so that the hand-back needs no launch of its own.
__global__ void __launch_bounds__(256) k_host_pack(DevCfg d, DevPtrs p, uint8_t* io, HostIo L, uint32_t* py_out, uint32_t* np_out) {
    const int t = threadIdx.x;
    {
        uint32_t* dst[2] = {py_out, np_out};
        const uint32_t* src[2] = {p.mt_py, p.mt_np};
        for (int k = 0; k < 2; k++) {
            if (!dst[k]) continue;  // uniform
            const uint32_t rp = p.rngpos[k];
            const uint32_t* in = src[k] + (size_t)CTF_RP_CUR(rp) * CTF_MT_N;
            for (int i = t; i < CTF_MT_N; i += 256) dst[k][i] = in[i];
            if (t == 0) dst[k][CTF_MT_N] = CTF_RP_POS(rp);
        }
    }
    for (int k = t; k < d.GS; k += 256) io[L.grid + k] = p.grid[k];
    for (int k = t; k < d.RS; k += 256) io[L.rec + k] = p.rec[k];
    if (d.log_metrics) {
        uint32_t* m = (uint32_t*)(io + L.metrics);
        for (int k = t; k < CTF_N_METRICS * d.N; k += 256) m[k] = (uint32_t)p.metrics[k];
        uint32_t* v = (uint32_t*)(io + L.vis);
        for (int k = t; k < d.N * d.GS; k += 256) v[k] = p.vis[k];
        uint32_t* lg = (uint32_t*)(io + L.vislog);
        const uint32_t* src = (const uint32_t*)p.vislog;  // n_envs == 1: the ring [512][N] u16 is contiguous, N even or odd: 512 * N * 2 bytes
        for (int k = t; k < CTF_VIS_LOG * d.N / 2; k += 256) lg[k] = src[k];
    }
    if (t == 0) {
        uint32_t* ds = (uint32_t*)(io + L.done_status);
        ds[1] = *p.status;
        *p.status = 0;
    }
}

extern "C" int ctf_host_step(ctf_env* h, const int8_t* actions, const uint32_t* py_in, const uint32_t* np_in, uint32_t reverse_mask,
                             uint32_t flags, double* rewards, int32_t* done, uint32_t* status, ctf_state_view* view, uint32_t* py_out,
                             uint32_t* np_out, uint8_t* obs, uint16_t* meta, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (h->d.n_envs != 1) return fail(CTF_E_INVALID, "ctf_host_step serves a handle of ONE env (this one has %d): batches use ctf_step / ctf_observe", h->d.n_envs);
    const bool rng_io = py_in || np_in || py_out || np_out;
    if (rng_io)
        if (int rc = need_mode(h, CTF_RNG_MT19937, "ctf_host_step with generator states")) return rc;
    for (const uint32_t* s : {py_in, np_in})
        if (s && s[CTF_MT_N] > CTF_MT_N) return fail(CTF_E_INVALID, "MT position %u > 624", s[CTF_MT_N]);
    DeviceGuard guard(h->device);
    const DevCfg& d = h->d;
    const HostIo L = host_io_layout(d);
    if (!h->hio_host) {
        // pinned, mapped, coherent host memory: the kernels below read their inputs from it and write their outputs into it across the
        // bus (a few KB in, ~50 KB out) — no copy operations in the stream; a kernel's stores are visible to the host when the stream
        // has been waited for
        void* dev = nullptr;
        if (hipHostMalloc((void**)&h->hio_host, L.end, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
            return fail(CTF_E_NOMEM, "hipHostMalloc(%zu) failed", L.end);
        if (hipHostGetDevicePointer(&dev, h->hio_host, 0) != hipSuccess) {
            (void)hipHostFree(h->hio_host);
            h->hio_host = nullptr;
            return fail(CTF_E_HIP, "hipHostGetDevicePointer failed");
        }
        h->hio_dev = (uint8_t*)dev;
        memset(h->hio_host, 0, L.end);
    }
    hipStream_t st = (hipStream_t)stream;
    uint8_t *hd = h->hio_dev, *hh = h->hio_host;
    // inputs: actions and the generator states to install (the previous call has waited for the stream: nothing reads the block now)
    if (actions) memcpy(hh + L.actions, actions, (size_t)d.N);
    if (py_in) memcpy(hh + L.py_in, py_in, (CTF_MT_N + 1) * 4);
    if (np_in) memcpy(hh + L.np_in, np_in, (CTF_MT_N + 1) * 4);
    if (py_in || np_in) {
        HIP_TRY(ctf_launch_import_rng(d, h->p, py_in ? (const uint32_t*)(hd + L.py_in) : nullptr,
                                      np_in ? (const uint32_t*)(hd + L.np_in) : nullptr, 0, 1, st));
        HIP_TRY(ctf_launch_rng_refill(d, h->p, 0, 1, 1, st));
    }
    if (actions)
        HIP_TRY(ctf_launch_step(d, h->p, (const int8_t*)(hd + L.actions), nullptr, (double*)(hd + L.rw64), hd + L.done_status, flags,
                                1, st));
    if (obs || meta)
        HIP_TRY(ctf_launch_observe(d, h->p, obs ? hd + L.obs : nullptr, meta ? (uint16_t*)(hd + L.meta) : nullptr,
                                   resolve_reverse(h, reverse_mask), h->n_cus, st));
    hipLaunchKernelGGL(k_host_pack, dim3(1), dim3(256), 0, st, d, h->p, hd, L, py_out ? (uint32_t*)(hd + L.py_out) : nullptr,
                       np_out ? (uint32_t*)(hd + L.np_out) : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // the only wait of the call: every output already lies in host memory
    if (rewards && actions) memcpy(rewards, hh + L.rw64, (size_t)d.N * 8);
    const uint32_t* ds = (const uint32_t*)(hh + L.done_status);
    if (status) *status = ds[1];
    if (py_out) memcpy(py_out, hh + L.py_out, (CTF_MT_N + 1) * 4);
    if (np_out) memcpy(np_out, hh + L.np_out, (CTF_MT_N + 1) * 4);
    if (obs) memcpy(obs, hh + L.obs, (size_t)d.obs_bytes);
    if (meta) memcpy(meta, hh + L.meta, (size_t)d.N * d.M * 2);
    int32_t misc[4];
    ctf_state_view local;
    ctf_state_view* out = view ? view : &local;
    decode_record(d, hh + L.grid, hh + L.rec, d.log_metrics ? (const int32_t*)(hh + L.metrics) : nullptr, out, misc);
    if (done) *done = out->done;
    if (view && d.log_metrics) {
        std::vector<uint32_t> v((size_t)d.N * d.GS);
        memcpy(v.data(), hh + L.vis, v.size() * 4);
        const int folded = misc[3] >> CTF_F_FOLDED_SHIFT;
        const int count = misc[0] - folded;
        std::vector<uint16_t> entries((size_t)(count > 0 ? count : 0) * d.N);
        const uint16_t* ring = (const uint16_t*)(hh + L.vislog);
        for (int r = 0; r < count; r++)
            memcpy(entries.data() + (size_t)r * d.N, ring + (size_t)((folded + 1 + r) & (CTF_VIS_LOG - 1)) * d.N, (size_t)d.N * 2);
        decode_visitation(d, misc, v, entries.data(), count > 0 ? count : 0, out);
    }
    return CTF_OK;
}

extern "C" int ctf_set_state(ctf_env* h, int32_t e, const ctf_state_view* in) {
    if (!h || !in) return fail(CTF_E_INVALID, "null argument");
    if (e < 0 || e >= h->d.n_envs) return fail(CTF_E_RANGE, "env index %d", e);
    DeviceGuard guard(h->device);
    const DevCfg& d = h->d;
    for (int i = 0; i < d.N; i++) {
        if (in->pos[i][0] < 0 || in->pos[i][0] >= d.G || in->pos[i][1] < 0 || in->pos[i][1] >= d.G)
            return fail(CTF_E_INVALID, "pos[%d] outside the grid", i);
        if (in->perm[i] >= d.N) return fail(CTF_E_INVALID, "perm[%d]", i);
        if (in->inventory[i] < 0 || in->inventory[i] > 1000) return fail(CTF_E_INVALID, "inventory[%d]", i);
    }
    for (int k = 0; k < d.GG; k++)
        if (in->grid[k] > 13) return fail(CTF_E_INVALID, "grid[%d]", k);
    if (in->step_count < 0 || in->step_count >= (1 << 28)) return fail(CTF_E_INVALID, "step_count %d", in->step_count);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint8_t> rec((size_t)d.RS, 0), grid((size_t)d.GS, 0);
    memcpy(grid.data(), in->grid, (size_t)d.GG);
    for (int i = 0; i < d.N; i++) {
        memcpy(rec.data() + 8 * i, &in->hp[i], 8);
        rec[d.off_pos + 2 * i] = (uint8_t)in->pos[i][0];
        rec[d.off_pos + 2 * i + 1] = (uint8_t)in->pos[i][1];
        rec[d.off_flag + i] = in->has_flag[i];
        rec[d.off_perm + i] = in->perm[i];
        const int16_t inv = (int16_t)in->inventory[i];
        memcpy(rec.data() + d.off_inv + 2 * i, &inv, 2);
    }
    // the given visitation maps become the base maps; the log is empty (folded up to step_count)
    const int32_t misc[4] = {in->step_count, in->team_captures[0], in->team_captures[1],
                             (in->done ? CTF_F_DONE : 0) | (d.log_metrics ? 0 : CTF_F_BASE_ZERO) | (in->step_count << CTF_F_FOLDED_SHIFT)};
    memcpy(rec.data() + d.off_misc, misc, 16);
    HIP_TRY(hipMemcpy(h->p.grid + (size_t)e * d.GS, grid.data(), (size_t)d.GS, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->p.rec + (size_t)e * d.RS, rec.data(), (size_t)d.RS, hipMemcpyHostToDevice));
    if (d.log_metrics) {
        std::vector<int32_t> m((size_t)CTF_N_METRICS * d.N);
        for (int k = 0; k < CTF_N_METRICS; k++)
            for (int i = 0; i < d.N; i++) m[(size_t)k * d.N + i] = in->metrics[k][i];
        HIP_TRY(hipMemcpy(h->p.metrics + (size_t)e * CTF_N_METRICS * d.N, m.data(), m.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> v((size_t)d.N * d.GS, 0);
        for (int i = 0; i < d.N; i++)
            for (int k = 0; k < d.GG; k++) v[(size_t)i * d.GS + k] = in->visitation[i][k];
        HIP_TRY(hipMemcpy(h->p.vis + (size_t)e * d.N * d.GS, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    }
    return CTF_OK;
}

extern "C" int ctf_export_counters(ctf_env* h, int32_t* metrics_dev, int32_t* captures_dev, int32_t* steps_dev, void* stream) {
    if (!h) return fail(CTF_E_INVALID, "null handle");
    if (!metrics_dev && !captures_dev && !steps_dev) return CTF_OK;
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_export_counters(h->d, h->p, metrics_dev, captures_dev, steps_dev, (hipStream_t)stream));
    return CTF_OK;
}

extern "C" int ctf_status(ctf_env* h, uint32_t* out_bits, void* stream) {
    if (!h || !out_bits) return fail(CTF_E_INVALID, "null argument");
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(out_bits, h->p.status, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(h->p.status, 0, 4, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CTF_OK;
}

extern "C" int ctf_random_actions(ctf_env* h, int8_t* actions, uint64_t seed, uint32_t step, uint32_t env_offset, void* stream) {
    if (!h || !actions) return fail(CTF_E_INVALID, "null argument");
    DeviceGuard guard(h->device);
    HIP_TRY(ctf_launch_random_actions(h->d, actions, seed, step, env_offset, (hipStream_t)stream));
    return CTF_OK;
}
