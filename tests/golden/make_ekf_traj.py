#!/usr/bin/env python3
"""Independent numpy transliteration of EKF::update (reference ekf_ws/src/localization_pkg/src/ekf.cpp:37-179) run over the
golden measurement streams of the imported reference simulator -> tests/golden/ekf_traj_*.npz.

Purpose: the C++ filter of the reference cannot be built here (filter.h:8-40 needs ROS / Eigen / yaml-cpp / GTSAM), so the
oracle's EKF arithmetic (oracle/slam_oracle.cpp) has no reference binary to be pinned against.  This script is a SECOND,
differently structured statement of the same lines: dense numpy matrices exactly as the reference asks Eigen for them
(F_x P F_x^T + F_v V F_v^T, (K H) P, Y p_temp Y^T with a full Y), numpy's inverse of S, float32 casts where the reference
declares `float`.  tests/test_oracle.py::test_ekf_oracle_matches_numpy_transliteration runs the oracle (MODE_DENSE |
MATH_LIBM) over the same streams and requires agreement to 1e-10 at every checkpoint.  It removes the single-author risk of
the oracle; it does NOT make parity "pinned" (only the reference binary could).

Run from the repo root:  python tests/golden/make_ekf_traj.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
PI = 3.14159265358979323846            # filter.h:42
TWO_PI = 2 * PI


class NumpyEKF:
    """ekf.cpp:4-21 (constructor), :29-34 (init), :37-179 (update); readCommonParams quirk filter.h:116-117."""

    def __init__(self, W_00=0.01, W_11=0.01, v_d=0.0, v_th=0.0, w_r=0.0, w_b=0.0):
        self.x_t = np.zeros(3)
        self.P_t = np.diag([0.01 * 0.01, 0.01 * 0.01, 0.005 * 0.005])           # ekf.cpp:11-14
        self.H_w = np.eye(2)                                                      # :20
        self.V = np.eye(2); self.V[0, 0] = W_00; self.V[1, 1] = W_11              # filter.h:116-117 (W_00, W_11 land in V)
        self.W = np.eye(2)                                                        # ... and W stays the identity
        self.v_d, self.v_th, self.w_r, self.w_b = f32(v_d), f32(v_th), f32(w_r), f32(w_b)   # floats (filter.h:84-89)
        self.M = 0
        self.lm_IDs = []
        self.timestep = 0

    def init(self, x_0, y_0, yaw_0):
        self.x_t = np.array([float(f32(x_0)), float(f32(y_0)), float(f32(yaw_0))])

    def update(self, fwd, ang, lm_meas):
        self.timestep += 1                                                        # :39
        d_d, d_th = f32(fwd), f32(ang)                                            # :43-44 (float)
        n = 3 + 2 * self.M
        th = self.x_t[2]
        F_x = np.eye(n)                                                           # :47
        F_x[0, 2] = float(f32(-1) * d_d) * math.sin(th)                           # :48  (-1*d_d is a float product)
        F_x[1, 2] = float(d_d) * math.cos(th)                                     # :49
        F_v = np.zeros((n, 2))                                                    # :51
        F_v[0, 0] = math.cos(th); F_v[1, 0] = math.sin(th); F_v[2, 1] = 1         # :52-54
        x_pred = self.x_t.copy()                                                  # :56
        x_pred[0] = self.x_t[0] + float(d_d + self.v_d) * math.cos(th)            # :57  (d_d + v_d is a float sum)
        x_pred[1] = self.x_t[1] + float(d_d + self.v_d) * math.sin(th)            # :58
        x_pred[2] = math.remainder(th + float(d_th) + float(self.v_th), TWO_PI)   # :59
        P_pred = F_x @ self.P_t @ F_x.T + F_v @ self.V @ F_v.T                    # :61
        lm_meas = np.asarray(lm_meas, dtype=np.float32).reshape(-1)               # :64
        num_landmarks = len(lm_meas) // 3                                         # :65
        if num_landmarks < 1:                                                     # :67-71
            self.x_t, self.P_t = x_pred, P_pred
            return
        for l in range(num_landmarks):                                            # :73
            r, b = f32(lm_meas[3 * l + 1]), f32(lm_meas[3 * l + 2])               # :75-76 (float)
            i = -1
            ident = int(lm_meas[3 * l])                                           # :101 (landmark_id_is_known)
            for j in range(self.M):                                               # :103-108
                if self.lm_IDs[j] == ident:
                    i = j
                    break
            if i != -1:
                i = i * 2 + 3                                                     # :113
                dx = self.x_t[i] - x_pred[0]                                      # landmark from x_t, vehicle from x_pred
                dy = self.x_t[i + 1] - x_pred[1]
                dist = f32(math.sqrt(dx ** 2 + dy ** 2))                          # :115 (float)
                dd, d2 = float(dist), float(dist * dist)                          # dist*dist is a float product
                n = 3 + 2 * self.M
                H_x = np.zeros((2, n))                                            # :117
                H_x[0, 0] = -dx / dd; H_x[0, 1] = -dy / dd                         # :118-119
                H_x[1, 0] = dy / d2; H_x[1, 1] = -dx / d2; H_x[1, 2] = -1          # :120-122
                H_x[0, i] = dx / dd; H_x[0, i + 1] = dy / dd                       # :123-124
                H_x[1, i] = -dy / d2; H_x[1, i + 1] = dx / d2                      # :125-126
                ang_f = f32(math.remainder(math.atan2(dy, dx) - x_pred[2], TWO_PI))   # :129 (float)
                nu = np.array([float(r - dist - self.w_r), float(b - ang_f - self.w_b)])   # :130-131 (float arithmetic)
                S = H_x @ P_pred @ H_x.T + self.H_w @ self.W @ self.H_w.T         # :133
                K = P_pred @ H_x.T @ np.linalg.inv(S)                             # :135
                x_pred = x_pred + K @ nu                                          # :138
                x_pred[2] = math.remainder(x_pred[2], TWO_PI)                     # :139
                P_pred = P_pred - (K @ H_x) @ P_pred                              # :140  (Eigen: (K*H)*P)
            else:
                self.M += 1                                                       # :144
                n = 3 + 2 * self.M
                phi = x_pred[2] + float(b)
                x_pred = np.concatenate([x_pred, [x_pred[0] + float(r) * math.cos(phi),      # :146-148
                                                  x_pred[1] + float(r) * math.sin(phi)]])
                self.lm_IDs.append(ident)                                         # :150
                Y = np.eye(n)                                                     # :153
                Y[n - 2, n - 2] = math.cos(phi); Y[n - 2, n - 1] = -float(r) * math.sin(phi)      # :155-156
                Y[n - 1, n - 2] = math.sin(phi); Y[n - 1, n - 1] = float(r) * math.cos(phi)       # :157-158
                Y[n - 2, 0] = 1; Y[n - 2, 1] = 0; Y[n - 2, 2] = -float(r) * math.sin(phi)         # :160-162
                Y[n - 1, 0] = 0; Y[n - 1, 1] = 1; Y[n - 1, 2] = float(r) * math.cos(phi)          # :163-165
                p_temp = np.zeros((n, n))                                         # :168
                p_temp[:n - 2, :n - 2] = P_pred                                   # :169
                p_temp[n - 2:, n - 2:] = self.W                                   # :170
                P_pred = Y @ p_temp @ Y.T                                         # :172
        self.x_t, self.P_t = x_pred, P_pred                                       # :176-177


def run(fixture, every=20):
    g = np.load(os.path.join(HERE, fixture))
    T = int(g["T"])
    e = NumpyEKF(); e.init(0.0, 0.0, 0.0)
    L = int(g["L"])
    n_max = 3 + 2 * L
    ts, xs, dg, Ms = [], [], [], []
    for t in range(T):
        k = int(g["meas_count"][t])
        e.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
        if t % every == every - 1 or t == T - 1:
            n = 3 + 2 * e.M
            x = np.zeros(n_max); x[:n] = e.x_t
            d = np.zeros(n_max); d[:n] = np.diag(e.P_t)
            ts.append(t + 1); xs.append(x); dg.append(d); Ms.append(e.M)
    out = os.path.join(HERE, "ekf_traj_" + fixture.replace("sim_", ""))
    np.savez_compressed(out, fixture=fixture, steps=np.array(ts), x=np.array(xs), diagP=np.array(dg), M=np.array(Ms),
                        ids=np.array(e.lm_IDs), P_final=e.P_t)
    print(out, "T", T, "M", e.M, "final pose", e.x_t[:3])


if __name__ == "__main__":
    for fx in ("sim_seed0_L20_T1000.npz", "sim_seed1_L20_T400.npz", "sim_seed2_L50_T1000.npz", "sim_seed1234_L50_T400.npz"):
        run(fx)
