python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -8
