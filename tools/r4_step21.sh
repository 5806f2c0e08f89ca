#!/bin/bash
# final: the whole GPU suite on the final code
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
