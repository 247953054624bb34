#!/bin/bash
# round 5, GPU call 17 (closing): the whole GPU suite and smoke on the final library
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/e17_pytest.log 2>&1; tail -3 $O/e17_pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
