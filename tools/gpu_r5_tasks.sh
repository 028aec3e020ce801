#!/bin/bash
# the ddd / exdet task tests alone (tests/test_gpu_tasks.py) -- one short GPU-box session
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/tasks; mkdir -p $O
timeout 145 python -m pytest tests/test_gpu_tasks.py -m gpu -q --timeout 120 -p no:cacheprovider > $O/pytest_tasks.log 2>&1
echo "rc $?" >> $O/pytest_tasks.log
tail -60 $O/pytest_tasks.log
