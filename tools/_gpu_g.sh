#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
export MK_DEBUG=1
timeout 900 python tools/_diag_ref.py 8000000 60 1500 2000 > $O/diag_ref.txt 2>&1; tail -12 $O/diag_ref.txt | cut -c1-1200
