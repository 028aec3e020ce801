#!/bin/bash
# the ddd / exdet task tests (tests/test_gpu_tasks.py) + the exct decode goldens -- one short GPU-box session
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/tasks; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_tasks.py tests/test_gpu_exct.py -m gpu -q --timeout 90 -p no:cacheprovider > $O/pytest_tasks.log 2>&1
echo "rc $?" >> $O/pytest_tasks.log
tail -60 $O/pytest_tasks.log
