#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
export MK_DEBUG=1
timeout 900 python tools/_diag_var.py 8000000 60 1500 2000 > $O/diag_var.txt 2>&1; grep -v "^\[prefilter\] chunk" $O/diag_var.txt | tail -30 | cut -c1-600
