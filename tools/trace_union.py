#!/usr/bin/env python3
"""Roofline figure of a concurrent-launch schedule FROM THE rocprofv3 KERNEL TRACE (not from in-process HIP events).

    python tools/trace_union.py <dir with *_kernel_trace.csv> <bench line .json> <out .json> [kernel substring]

`bench.py`'s default schedule launches the posterior kernel twice per step on two streams; the launches overlap one
another, so `bytes per launch / average launch duration` (what `--stats` gives) is not the rate the kernel sustained.
This script takes the start / end time stamps of every dispatch of the kernel in the timed region of the profiled run
(the last `steps x launches_per_step` dispatches of the schedule's grid size before the bench's un-overlapped
re-measurements, identified by order: the first `warmup x launches_per_step` are the untimed steps), forms the union
of their intervals and writes

    achieved = algorithmic bytes of those launches / union time,   frac = achieved / 8 TB/s

next to the bench line's own `roofline` (HIP events inside the same process), so that `roofline.frac` follows from
tracked rocprof data.  Also written: the mean launch duration (what *_kernel_stats.csv averages) and the wall span."""
import csv
import glob
import json
import os
import sys


def union_ms(spans):
    spans = sorted(spans)
    tot, a0, b0 = 0.0, spans[0][0], spans[0][1]
    for a, b in spans[1:]:
        if a > b0:
            tot += b0 - a0
            a0, b0 = a, b
        else:
            b0 = max(b0, b)
    return (tot + b0 - a0) / 1e6


def main():
    trace_dir, bench_json, out_json = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "posterior_step_kernel<float, 3, 4, 0, 1, false"
    line = [l for l in open(bench_json) if l.startswith("{")][-1]
    bench = json.loads(line)
    rf = bench["roofline"]
    per_step, steps, warm = int(rf["launches_per_step"]), int(bench["steps"]), int(bench["warmup"])
    sizes = rf.get("instances_per_launch_by_part") or rf.get("queries_per_launch_by_part") or \
        [int(rf.get("instances_per_launch", rf.get("queries_per_launch", 0)))]
    sizes = set(int(v) for v in sizes)
    files = sorted(glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getsize)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"] and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) in sizes:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    want = (warm + steps) * per_step
    if len(rows) < want:
        raise SystemExit("trace holds %d dispatches of %r at %s workgroups, the run made >= %d" % (len(rows), needle, sorted(sizes), want))
    timed = rows[warm * per_step:want]                  # dispatch order = start order: untimed steps first, re-measurements last
    u = union_ms(timed)
    unit = rf["unit"]
    per_launch = rf.get("algorithmic_bytes_per_launch", rf.get("algorithmic_flops_per_launch"))
    scale = 1e9 if unit == "GB/s" else 1e12
    achieved = per_launch * len(timed) / (u * 1e-3) / scale
    out = dict(source="rocprofv3 --kernel-trace time stamps of %d dispatches (%d timed steps x %d launches) of %s, %s workgroups each"
                      % (len(timed), steps, per_step, needle, sorted(sizes)),
               union_busy_ms=u, union_busy_ms_per_step=u / steps,
               mean_launch_ms=sum(b - a for a, b in timed) / len(timed) / 1e6,
               span_ms=(max(b for _, b in timed) - min(a for a, _ in timed)) / 1e6,
               algorithmic_units_per_launch=per_launch, unit=unit, achieved=achieved, peak=rf["peak"], frac=achieved / rf["peak"],
               bench_line_roofline=dict(achieved=rf["achieved"], frac=rf["frac"], kernel_ms=rf["kernel_ms"],
                                        kernel_busy_ms_per_step=rf["kernel_busy_ms_per_step"]),
               frac_trace_over_events=achieved / rf["peak"] / rf["frac"],
               bench=dict(value=bench["value"], ms_per_step=bench["ms_per_step"], steps=steps, warmup=warm))
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("union_busy_ms_per_step", "mean_launch_ms", "achieved", "frac", "frac_trace_over_events")}))


if __name__ == "__main__":
    main()
