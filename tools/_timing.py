"""Device timing for the tools: the MI355X lowers its clocks within a few ms of idleness and needs ~15 ms of load to raise
them again (tools/probe_ramp.py), so a measurement that starts from an idle device and lasts a few ms reads 5-10 % slow.
`timeit` first runs the function for at least `warm_ms` of device time, then times `reps` calls between two HIP events."""
import time

import torch


def timeit(fn, reps=20, warm_ms=40.0, min_warm=3):
    torch.cuda.synchronize()
    t0, k = time.perf_counter(), 0
    while k < min_warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
