#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts_profile.sh into small, committed summaries under profiles/.

    python tools/summarize_profile.py gpurun_out/<tag> profiles/<tag>

Writes <prefix>_kernel_stats.csv (per kernel: calls, total/avg duration, % of GPU time) and pmc_summary.json
(per kernel: FETCH/WRITE bytes per launch with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts wide
coalesced reads at 1/2, so it is doubled; units are KiB as rocprofv3 reports them; plus SQ counters per launch).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("ardae::(anonymous namespace)::", "").replace("ardae::wide::", "").replace("ardae::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:120]


def find(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None      # the newest one: gpurun merges into an existing directory, older runs' files stay there


def main(src, dst_prefix):
    os.makedirs(os.path.dirname(dst_prefix) or ".", exist_ok=True)
    tr = find(os.path.join(src, "trace"), "*kernel_trace.csv")
    agg = collections.OrderedDict()
    if tr:
        for r in csv.DictReader(open(tr)):
            k = short(r["Kernel_Name"])
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            a = agg.setdefault(k, [0, 0])
            a[0] += 1; a[1] += d
        tot = sum(v[1] for v in agg.values())
        with open(dst_prefix + "_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
            for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, c, f"{t/1e3:.1f}", f"{t/1e3/c:.2f}", f"{100*t/tot:.2f}"])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        cc = find(os.path.join(src, sub), "*counter_collection.csv")
        if not cc:
            continue
        for r in csv.DictReader(open(cc)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in pmc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches_sampled"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            # KiB per launch; FETCH_SIZE x2 (gfx950 reports 64 B per 128-B request for wide coalesced streams)
            e["hbm_bytes_per_launch"] = 1024.0 * (2.0 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0))
        out[k] = e
    # bench.py looks kernels up by the template-argument form it prints; add those aliases
    for k, (c, t) in agg.items():
        e = out.setdefault(k, {})
        e["trace_calls"] = c
        e["trace_avg_us"] = t / 1e3 / c
    for k in list(out):
        m = re.match(r"(linear_kernel)<(.*)>", k)
        if m:
            out[f"{m.group(1)}<{m.group(2)}>"] = out[k]
        m = re.match(r"linear_wide_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>", k)
        if m:   # bench.py prints the two flags as 0 / 1
            a = list(m.groups())
            out["linear_wide_kernel<%s, %s, %s, %s, %d, %d>" % (a[0], a[1], a[2], a[3], a[4] == "true", a[5] == "true")] = out[k]
        if k.startswith("wgrad_wide_kernel<"):
            out["wgrad_wide_kernel<256x256>" if "4, 4, 2" in k else "wgrad_wide_kernel<256x32>"] = out[k]
    # (the kernel-trace pass's average durations - trace_avg_us - travel in the same hash-stamped file: bench.py's roofline.frac quotes them)
    # which build the counters belong to: bench.py quotes them only for a library built from the same kernel sources
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import subprocess
    from bench import kernel_source_hash
    try:
        git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
        dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "pytorch-ardae-vae_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
    except OSError:
        git, dirty = "?", False
    out["_meta"] = {"git": git + ("+uncommitted kernel changes" if dirty else ""), "kernel_source_sha256": kernel_source_hash(), "source": src}
    with open(os.path.join(os.path.dirname(dst_prefix) or ".", "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst_prefix + "_kernel_stats.csv", "and pmc_summary.json with", len(out), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
