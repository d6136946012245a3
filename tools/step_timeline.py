#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV (graph replay or eager launches).

    python tools/step_timeline.py <dir or kernel_trace.csv> [--step -2] [--out profiles/<tag>_timeline.txt]

A step ends with its `step_state_advance_kernel` launch (the last launch of `ArdaeEngine`'s plan: the device step state is
advanced for the NEXT step); the bench's two `bernoulli` launches behind it belong to the next step.  For every kernel of the chosen step: start offset, duration,
queue, and the idle gap on the critical path (time since the latest end of any kernel that started before it).  The summary
gives the union of busy time, the idle time inside the step and the per-family totals - the numbers DESIGN section 6 quotes.
"""
import argparse
import collections
import csv
import glob
import os
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("ardae::(anonymous namespace)::", "").replace("ardae::wide::", "").replace("ardae::", "")
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--step", type=int, default=-2, help="which step of the trace (negative: from the end)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--marker", default="step_state_advance_kernel")
    a = ap.parse_args()
    path = a.src
    if os.path.isdir(path):
        fs = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
        path = fs[-1]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    # the model optimiser's advance closes a step (the cDAE's own advance, when it has an Adam block, sits inside it)
    marks = [i for i, r in enumerate(rows) if r[2].startswith(a.marker)]
    # keep only marks that are followed by a long stretch (first advance of a step): steps = gaps between consecutive first marks
    starts = []
    for i in marks:
        if not starts or rows[i][0] - rows[starts[-1]][0] > 200_000:   # > 0.2 ms apart
            starts.append(i)
    k = a.step if a.step >= 0 else len(starts) + a.step
    lo, hi = starts[k] + 1, starts[k + 1] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    out = []
    busy_end = t0
    idle = 0
    fam = collections.OrderedDict()
    for s, e, n, q, wg in step:
        gap = s - busy_end
        if gap > 0:
            idle += gap
        out.append(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:8.2f}  gap {max(gap, 0) / 1e3:6.2f}{'*' if gap < 0 else ' '} q{q} wg{wg:<6d} {n[:110]}")
        busy_end = max(busy_end, e)
        f = fam.setdefault(n.split("<", 1)[0], [0, 0])
        f[0] += 1
        f[1] += e - s
    span = rows[hi][0] - t0
    lines = [f"# {path}", f"# step {k} of {len(starts) - 1}: {len(step)} launches, span {span / 1e3:.1f} us (start of this step to start of the next), "
             f"idle on the device inside the step {idle / 1e3:.1f} us ('*' = launched while an earlier kernel was still running)"]
    lines += out
    lines.append("# per family: launches, summed duration (us)")
    for n, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"# {c:4d} {t / 1e3:9.1f}  {n}")
    # spans of all complete steps, for the spread
    spans = [(rows[starts[i + 1]][0] - rows[starts[i]][0]) / 1e3 for i in range(len(starts) - 1)]
    lines.append("# step spans (us): " + " ".join(f"{s:.0f}" for s in spans[-12:]))
    text = "\n".join(lines)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
