#!/bin/bash
# the ddd / exdet task tests (tests/test_gpu_tasks.py) + the exct decode goldens -- one short GPU-box session
# usage: bash tools/gpu_r5_tasks.sh [pytest -k expression]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/tasks; mkdir -p $O
timeout ${TASKS_TIMEOUT:-100} python -m pytest tests/test_gpu_tasks.py tests/test_gpu_exct.py -m gpu -q --timeout 90 -p no:cacheprovider ${1:+-k "$1"} > $O/pytest_tasks.log 2>&1
echo "rc $?" >> $O/pytest_tasks.log
tail -60 $O/pytest_tasks.log
