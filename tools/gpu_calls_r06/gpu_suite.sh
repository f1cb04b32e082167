#!/bin/bash
# Round 6: the driver's GPU suite with per-test durations (gpurun_out/r06_final/pytest_gpu_full_suite.log)
mkdir -p gpurun_out/r06_final
python -c "import __graft_entry__ as g; g.build()" || exit 1
( time timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests/ -x -q -m gpu --durations=40 ${PYTEST_EXTRA} ) > gpurun_out/r06_final/pytest_gpu_full_suite.log 2>&1
tail -60 gpurun_out/r06_final/pytest_gpu_full_suite.log
