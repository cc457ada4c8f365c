# round 5, session u: long-message path (launch pairs LDS kernel + streamed kernel per instance; UKF / UKF_LOC / device buffers / generator)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_parity_ukf_gpu.py -x -q -m gpu -k "long or over_long or any_length or loc" > gpurun_out/r5u/new_tests.txt 2>&1
tail -5 gpurun_out/r5u/new_tests.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5u/driver.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r5u/driver.json').read()); print(d['value'], d['roofline']['frac'], d['config']['secondary_digest'])"
timeout 400 python3 tools/gpu_soak_adversarial.py 300 51 both > gpurun_out/r5u/soak_adv.txt 2>&1; tail -3 gpurun_out/r5u/soak_adv.txt
