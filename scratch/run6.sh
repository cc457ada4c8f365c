python tools/ekf_launch_edges.py 2>&1 | tail -4
python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -5
