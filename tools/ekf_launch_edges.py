#!/usr/bin/env python3
"""EKF headline, K = 20 timed steps in one launch: does what precedes the launch on the stream change its duration?  
A: warm-up launch, host synchronisation (what bench.py's barrier + synchronize brackets do), timed launch.
B: warm-up launch and timed launch back to back on the stream, no host synchronisation between them.
C: as A, but the host sleeps 50 ms after the synchronisation (an idle device before the timed launch).
Durations from HIP events on the launch stream; the three variants alternate on ONE handle chain (same timesteps re-run on fresh handles)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

L, B, K, W, T0 = 50, 65536, 20, 5, 644
os.environ["SLAM_RUN_CHUNK"] = "0"
lm, cmds = make_scenario(1234, L, T0 + K + 8)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
res = {"A": [], "B": [], "C": []}
for rep in range(3):
    for var in ("A", "B", "C"):
        f = S.BatchedEKF(B, L, device=0).readParams()
        f.set_stream(stream.cuda_stream)
        f.set_map(lm); f.set_seed(2025); f.init(0.0, 0.0, 0.0)
        with torch.cuda.stream(stream):
            f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
            f.run_sim(cmds[1:T0 - W]); f.set_run_chunk(K)
            f.sync()
            f.run_sim(cmds[T0 - W:T0])
            if var != "B":
                f.sync(); torch.cuda.synchronize(dev)
            if var == "C":
                time.sleep(0.05)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); f.run_sim(cmds[T0:T0 + K]); e1.record(stream)
            f.sync(); torch.cuda.synchronize(dev)
        res[var].append(e0.elapsed_time(e1))
        f.close()
for var, what in (("A", "warm-up, host sync, timed launch (bench.py)"), ("B", "warm-up and timed launch back to back"), ("C", "warm-up, host sync, 50 ms idle, timed launch")):
    v = np.array(res[var])
    print(f"{var}: {what:48s} {v.mean():7.3f} ms per 20-step launch (runs: {' '.join(f'{x:.3f}' for x in v)})  -> {B * K / v.mean() / 1e3:.2f} M steps/s")
