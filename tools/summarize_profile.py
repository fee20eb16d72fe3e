#!/usr/bin/env python3
"""Condense one tools/profile.sh output directory into the files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof_<tag> <round-tag> [workload-key]

writes profiles/<round-tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, our kernels first),
profiles/<round-tag>_pmc_summary.json (per-kernel mean of every counter of every --pmc pass, with the mean launch
duration of that pass beside it) and updates profiles/traffic.json (HBM bytes per k_observe launch = WRITE_SIZE +
2 x FETCH_SIZE, KiB units, the gfx950 correction of MI355X_MICROARCH.md) which bench.py reads for roofline.traffic.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)  # a directory reused by a later run: the newest files count
    return hits[-1] if hits else None


def is_ours(name):
    return name.startswith("void k_") or name.startswith("k_")


def pmc_pass(path, tag, out):
    # rows: one per (dispatch, counter)
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    launch = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if not is_ours(k):
                continue
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            launch[k] = dict(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"], grid=r["Grid_Size"],
                             wg=r["Workgroup_Size"])
    for k, cs in vals.items():
        d = out.setdefault(k, {})
        d["launch"] = launch[k]
        for c, v in cs.items():
            d[c] = sum(v) / len(v)
        d["duration_ns@" + tag] = sum(dur[k].values()) / len(dur[k])
        d["launches@" + tag] = len(dur[k])


def main():
    src, tag = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else "arena_65536"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    stats = one(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats, newline="")))
        head, body = rows[0], rows[1:]
        body.sort(key=lambda r: (not is_ours(r[0]),))  # stable: ours first, rocprofv3's order (by total time) inside
        with open(os.path.join(root, tag + "_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows([head] + body)
    # provenance: the bench line the kernel-trace pass printed (its own HIP-event timings, to hold against the trace's averages) and
    # the exact commands — every file of a tag comes from ONE invocation of the profiling script
    tlog = os.path.join(src, "trace.log")
    if os.path.exists(tlog):
        lines = [l for l in open(tlog, errors="replace") if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(root, tag + "_bench_under_rocprof.json"), "w").write(lines[-1])
    # the same launches bench.py timed: the last `windows x steps` launches of the step and render kernels before the 12 sampled
    # steps at the end of the process, from the kernel trace — the average to hold against the line's kernels_ms and ms_per_step
    timed_note = ""
    ktrace = one(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    bpath = os.path.join(root, tag + "_bench_under_rocprof.json")
    if ktrace and os.path.exists(bpath) and "policy" not in tag:
        line = json.loads(open(bpath).read())
        n_timed = int(line["steps"]) * len(line.get("windows_ms_per_step", [0]))
        n_sample = 12  # bench.py: SAMPLE_STEPS
        per = defaultdict(list)
        with open(ktrace, newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "k_step<" in k or ("k_observe" in k and "codes" not in k):
                    per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        timed = {}
        for v in per.values():
            v.sort()
        # the render delimits the region (nothing launches it after the sampled steps); k_step also runs in what follows them
        renders = [k for k in per if "k_observe" in k]
        render = max(renders, key=lambda k: len(per[k])) if renders else None
        if render and len(per[render]) > n_timed + n_sample:
            rv = per[render]
            t_lo, t_hi = rv[len(rv) - n_sample - n_timed - 1][1], rv[len(rv) - n_sample - 1][1]
            for k, v in per.items():
                w = [(s_, e) for s_, e in v if t_lo <= s_ < t_hi]
                if not w:
                    continue
                timed[k] = dict(launches=len(w), avg_ns=sum(e - s_ for s_, e in w) / len(w), all_launches=len(v),
                                all_avg_ns=sum(e - s_ for s_, e in v) / len(v))
            timed["_region"] = dict(launches=n_timed, avg_ns=(t_hi - t_lo) / n_timed, all_launches=n_timed, all_avg_ns=(t_hi - t_lo) / n_timed)
        if timed:
            json.dump(dict(source=tag, note="rocprofv3 --kernel-trace: the launches of bench.py's timed windows only (the last "
                           f"{n_timed} before the {n_sample} sampled steps that end the process); all_avg_ns = every launch of the process, placement search included",
                           bench_kernels_ms=line.get("kernels_ms"), bench_ms_per_step=line.get("ms_per_step"), kernels=timed),
                      open(os.path.join(root, tag + "_timed_region_kernels.json"), "w"), indent=1, sort_keys=True)
            timed_note = ("  " + tag + "_timed_region_kernels.json: the trace's averages over the timed windows' launches alone: "
                          + "; ".join(f"{k.split('(')[0].replace('void ', '')} {v['avg_ns'] / 1e6:.4f} ms x {v['launches']}" for k, v in sorted(timed.items()))
                          + " (_region: first timed launch's predecessor's end to the last one's end, per step; the windows' host-side joins included).")
    script = "tools/profile_policy.sh" if "policy" in tag else "tools/profile.sh"
    with open(os.path.join(root, tag + "_PROVENANCE.txt"), "w") as f:
        f.write(f"{tag}_kernel_stats.csv, {tag}_pmc_summary.json, {tag}_bench_under_rocprof.json: one invocation of `bash {script} {tag}` on one MI355X box "
                f"(BENCH_ARGS={os.environ.get('PROFILE_BENCH_ARGS', 'see the script')}); passes found: "
                + ", ".join(p for p in ("trace", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3") if os.path.isdir(os.path.join(src, p)))
                + ".  kernel_stats averages run over ALL launches of the process (stagger phase, warm-up and secondaries included); the bench line's "
                  "kernels_ms are HIP events around the two launches of 12 steps run AFTER the timed windows, net of an empty event pair."
                + timed_note + "\n")
    out = {}
    for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
        path = one(os.path.join(src, p, "**", "*_counter_collection.csv"))
        if path:
            pmc_pass(path, p, out)
    json.dump(out, open(os.path.join(root, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    # the render of this run: the tile kernel when it ran, else the wave-per-env kernel (never k_observe_codes)
    cands = [(k, v) for k, v in out.items() if "k_observe" in k and "codes" not in k and "WRITE_SIZE" in v and "FETCH_SIZE" in v]
    cands.sort(key=lambda kv: -kv[1]["WRITE_SIZE"] * kv[1].get("launches@pmc_write", 1))
    obs = cands[0][1] if cands else None
    obs_name = ("k_observe_tiles" if "tiles" in cands[0][0] else "k_observe") if cands else None
    if obs:
        tpath = os.path.join(root, "traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        w, r = obs["WRITE_SIZE"] * 1024.0, obs["FETCH_SIZE"] * 1024.0 * 2.0
        traffic[key] = {
            "kernel": obs_name, "k_observe_hbm_bytes_per_launch": w + r, "write_bytes": w, "fetch_bytes_corrected_x2": r, "source": tag,
            "note": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes (tools/profile.sh), KiB units x1024; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read; our "
                    "reads are 4 B/lane so the read side, 1% of the total, is approximate)"}
        json.dump(traffic, open(tpath, "w"), indent=1)
    for k, v in out.items():
        print(k, {c: round(x) for c, x in v.items() if c.startswith("duration")})


if __name__ == "__main__":
    main()
