#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_g
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
timeout 600 python3 examples/train_ddpgfd.py --envs 1024 --steps 360 > $out/example.log 2>&1; echo "example rc $?" >> $out/example.log
timeout 600 python3 examples/train_ddpgfd.py --envs 1024 --steps 360 --free-running > $out/example_free.log 2>&1; echo "example rc $?" >> $out/example_free.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
tail -4 $out/pytest_gpu.log; grep -E "^FAILED|^ERROR" $out/pytest_gpu.log | head; tail -5 $out/example.log; tail -5 $out/example_free.log; tail -2 $out/smoke.log
