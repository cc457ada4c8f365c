#!/usr/bin/env python3
"""Throughput of the HBM-streamed size classes (ekf_big_kernel.hip: EKF beyond 200 landmarks; ukf_big_kernel.hip: UKF beyond 50): steady state
with every landmark mapped, device-generated messages.  usage: gpu_big_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

for kind, L, B, T in (("ekf", 400, 256, 20), ("ekf", 400, 1024, 20), ("ekf", 1000, 256, 6), ("ekf", 200, 2048, 40), ("ukf", 100, 256, 6), ("ukf", 100, 1024, 6), ("ukf", 200, 256, 3), ("ukf", 50, 4096, 20)):
    lm, cmds = make_scenario(1234, L, 12 + T)
    f = (S.BatchedEKF(B, L) if kind == "ekf" else S.BatchedUKF(B, L)).readParams(); f.set_map(lm); f.set_seed(1); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    f.run_sim(cmds[1:12]); f.sync()
    f.k_histogram(reset=True)
    t0 = time.perf_counter(); f.run_sim(cmds[12:12 + T]); f.sync(); dt = time.perf_counter() - t0
    h = f.k_histogram().astype(float)
    name = f.kernel_info()["name"] if kind == "ekf" else ("ukf_big_*" if L > 50 else "ukf_*")
    print(f"{kind} L={L} (n={(3 if kind == 'ekf' else 4) + 2 * L}) batch {B}: {dt / T * 1e3:9.3f} ms per batch step, {B * T / dt:10.1f} steps/s, mean detections {(h * np.arange(8)).sum() / max(h.sum(), 1):.2f}, "
          f"landmarks mapped {f.landmark_counts().mean():.0f}, flagged {(f.status() != 0).sum()}  [{name}]", flush=True)
    f.close()
