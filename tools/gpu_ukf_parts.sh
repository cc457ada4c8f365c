for P in 2 3 4 2 3 4; do for L in 20 50; do echo -n "parts=$P L=$L: "; SLAM_UKF_PARTS=$P python bench.py --filter ukf --batch 4096 --landmarks $L --steps $([ $L = 20 ] && echo 100 || echo 30) --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
