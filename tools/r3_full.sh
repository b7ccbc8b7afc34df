#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3_full; mkdir -p $O; rm -f $O/*
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed\|error" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
