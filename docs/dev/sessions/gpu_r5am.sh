# round 5, session am: the driver's round-end sequence on the force-rebuilt final tree (GPU suite, smoke, bench command)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5am
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5am/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5am/pytest.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r5am/pytest.log | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5am/driver.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r5am/driver.json').read()); print(d['value'], d['roofline']['frac'], d['config']['secondary_digest'], len(json.dumps(d)))"
