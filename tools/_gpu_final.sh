#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04final; mkdir -p $O
MK_DEBUG=1 timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
rm -rf /tmp/pytest-of-root
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
QUICK=1 bash tools/collect_profiles.sh r04 > $O/collect.log 2>&1; tail -30 $O/collect.log | cut -c1-1500
