import sys, numpy as np, time
sys.path.insert(0,'.')
import live_ekf_slam_amd as S
from oracle import oracle as O
from live_ekf_slam_amd.scenario import make_scenario
for L,T,B in ((20,1000,8),(50,400,4)):
    lm,cmds=make_scenario(77,L,T)
    f=S.BatchedUKF(B,L).readParams(); f.set_map(lm); f.set_seed(3); f.init(0,0,0); f.run_sim(cmds)
    r=O.run_ukf_batch(lm,cmds,B,L,seed=3,nthreads=8)
    ok=np.array_equal(f.landmark_counts(),r['M']) and np.array_equal(f.error_stats(),r['avg_err']) and np.array_equal(f.status(),r['flags'])
    for b in range(B):
        n=4+2*r['M'][b]; s=f.get_state(b)
        ok=ok and np.array_equal(s['x'],r['x'][b,:n]) and np.array_equal(s['P'].ravel(),r['P'][b,:n*n])
    print(L,T,B,'bit-identical' if ok else 'MISMATCH','flags',np.unique(f.status()),'avg_err',f.error_stats().mean())
