# round 5, session t: the driver's command three times (the UKF secondary leg of session s read 1.94 M on the wall clock against 5.3 M by device events)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5t/driver_$i.json
  python3 -c "
import json
d=json.loads(open('gpurun_out/r5t/driver_$i.json').read()); print($i, d['value'], d['roofline']['frac'], d['config']['secondary_digest'], [ (l['name'][:10], l.get('ms_per_step'), l.get('leg_seconds')) for l in d['secondary']])"
done
