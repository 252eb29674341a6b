#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04n; mkdir -p $O
timeout 600 python tools/wide_classes_experiment.py 60000000 4000 > $O/classes_60m.txt 2>&1; grep "^{'classes" $O/classes_60m.txt | cut -c1-330
timeout 600 python tools/wide_classes_experiment.py 11800000 40000 > $O/classes_11m.txt 2>&1; grep "^{'classes" $O/classes_11m.txt | cut -c1-330
