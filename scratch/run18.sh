run() { echo "--- $*"; env "$@" python bench.py --filter pgs --iterative --batch 256 --no-cpu-baseline --no-parity-check 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*" | tr '\n' ' '; echo; }
run SLAM_PGS_GROUPS=1
run SLAM_PGS_GROUPS=2
run SLAM_PGS_GROUPS=4
run SLAM_PGS_GROUPS=8
run SLAM_PGS_GROUPS=4 SLAM_PGS_GROUP_PRIO=0
cat > /tmp/two.py <<'PY'
import os, sys, threading, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
H = int(sys.argv[1]); B = 256 // H
L, N = 200, 1000
lm, cmds = make_scenario(1234, L, N - 1)
hs = []
for k in range(H):
    pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams(solve_graph_every_iteration=True)
    pg.set_map(lm); pg.set_seed(2025); pg.set_instance_offset(k * B); pg.init(0.0, 0.0, 0.0)
    hs.append(pg)
t0 = time.time()
th = [threading.Thread(target=lambda p=p: p.run_sim_every_iteration(cmds)) for p in hs]
[t.start() for t in th]; [t.join() for t in th]
for p in hs: p.sync()
dt = time.time() - t0
print(f"{H} handles x {B} graphs on their own streams / host threads: {dt:.2f} s -> {256 * (N - 1) / dt:.0f} graph-ticks/s")
PY
python /tmp/two.py 1 2>&1 | tail -1
python /tmp/two.py 2 2>&1 | tail -1
python /tmp/two.py 4 2>&1 | tail -1
python /tmp/two.py 8 2>&1 | tail -1
