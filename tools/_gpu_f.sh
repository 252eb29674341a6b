#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
export MK_DEBUG=1
timeout 600 python tools/_diag60m.py 11800000 300 400 500 > $O/diag_11m_long.txt 2>&1; tail -12 $O/diag_11m_long.txt | cut -c1-900
timeout 900 python tools/_diag60m.py 60000000 300 30 120 > $O/diag_60m.txt 2>&1; tail -12 $O/diag_60m.txt | cut -c1-900
