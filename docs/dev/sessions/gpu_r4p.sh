#!/bin/bash
# round 4, session p: the whole GPU suite + smoke on the tree with the pipelined Cholesky and the streamed size classes
mkdir -p gpurun_out/r4p
timeout 2400 python -m pytest tests -q -m gpu -rs > gpurun_out/r4p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4p/pytest.log
tail -6 gpurun_out/r4p/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
