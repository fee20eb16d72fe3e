#!/usr/bin/env python3
"""Per-INSTANCE split of a counter for the fast and the slow buffer of tools/placement_pmc (rocprofv3 --output-format json keeps one
record per counter instance: 16 L2 channels x 8 XCDs for the TCC_* counters).  usage: placement_dims.py <rocprofv3 output dir>"""
import glob, json, sys, collections
import numpy as np

for f in glob.glob(sys.argv[1] + "/**/*results.json", recursive=True):
    d = json.load(open(f))["rocprofiler-sdk-tool"][0]
    names = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name", "?") for k in d["kernel_symbols"]}
    cname = {}
    for c in d.get("counters", []):
        h = c.get("id", {}).get("handle") if isinstance(c.get("id"), dict) else c.get("id")
        cname[h] = c.get("name", "?")
    per = collections.defaultdict(list)
    for rec in d["callback_records"]["counter_collection"]:
        k = names.get(rec["dispatch_data"]["dispatch_info"]["kernel_id"], "?")
        if "k_fill_tagged" not in k:
            continue
        tag = "fast" if ("<0>" in k or "ILi0E" in k) else ("slow" if ("<1>" in k or "ILi1E" in k) else None)
        if tag is None:
            continue
        by = collections.defaultdict(list)
        for r in rec["records"]:
            by[r["counter_id"]["handle"]].append(r["value"])
        for h, vals in by.items():
            per[(cname.get(h, str(h)), tag)].append(np.array(vals))
    for (c, tag), rows in sorted(per.items()):
        n = min(len(r) for r in rows)
        m = np.mean([r[:n] for r in rows], axis=0)  # mean over launches, per instance
        print(f"{c:40s} {tag}: launches {len(rows)} instances {n} total {m.sum():.0f} per-instance min {m.min():.0f} max {m.max():.0f} "
              f"std/mean {m.std() / max(m.mean(), 1e-9):.4f}")
        if n in (128, 16 * 8):
            x = m.reshape(8, 16)  # (the order of the instances is the tool's: XCD-major is an assumption, the spread is not)
            print("    sums of the 8 groups of 16:", " ".join(f"{v:.0f}" for v in x.sum(1)))
            print("    sums of the 16 groups of 8:", " ".join(f"{v:.0f}" for v in x.sum(0)))
