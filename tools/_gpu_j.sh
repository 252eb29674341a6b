#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j; mkdir -p $O
tools/micro/_build/urem24 > $O/urem24.txt 2>&1; tail -8 $O/urem24.txt
export MK_DEBUG=1
timeout 900 python tools/_diag_one.py 8000000 60 100 2000 > $O/diag_one.txt 2>&1; grep -c OK $O/diag_one.txt; grep DIFF $O/diag_one.txt | cut -c1-200
