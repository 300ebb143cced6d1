#!/bin/bash
# the GPU suite (optionally a -k subset) and smoke on a gpurun box: scripts/grun.sh 2000 'bash scripts/r4_suite.sh [-k expr]'
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --tb=short -rf "$@" > gpurun_out/suite_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/suite_tests.log | cut -c1-300 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
