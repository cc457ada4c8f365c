import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
f = S.BatchedUKFLoc(3).readParams()
print("created", flush=True)
f.init(0, 0, 0); print("init ok", flush=True)
try:
    f.update((0.1, 0.0), []); print("update ok (unexpected)", flush=True)
except S.SlamError as e:
    print("raised:", e, flush=True)
