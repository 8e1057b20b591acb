#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_validate
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log
python -m pytest tests -m gpu -q -s > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
tail -2 $out/smoke.log; tail -3 $out/gputests.log
