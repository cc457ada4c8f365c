for v in "x=1" "KP=100"; do echo "== $v"; timeout 200 python scratch/repro2.py $v 2>&1 | grep -v amdgpu.ids | tail -3; done
python tools/gpu_soak_pgs.py 240 11 wide 2>&1 | tail -3
python tools/gpu_soak_pgs.py 120 13 wide 2>&1 | tail -3
python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -2
