#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04_validate
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log
python -m pytest tests -m gpu -q -s > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
tail -2 $out/smoke.log; tail -3 $out/gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err; echo "bench rc $?"; tail -c 1800 $out/bench_driver_form.json
