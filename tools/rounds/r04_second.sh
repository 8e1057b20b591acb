#!/bin/bash
# round 4, second measurement batch: shared support skew 1e-6, plane tie rule (fp64), captured-record pool
out=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
python -m tests.studies.long_horizon > $out/long_horizon.txt 2>&1
python bench.py > $out/bench_default.log 2>&1
python bench.py --no-cpu-baseline --mode sim > $out/bench_sim.log 2>&1
python bench.py --no-cpu-baseline --rollout lockstep > $out/bench_lockstep.log 2>&1
python bench.py --no-cpu-baseline --config 5 > $out/bench_cfg5.log 2>&1
python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.log 2>&1
# the skew at 1e-5 in kernels AND oracle (the oracle is rebuilt on the box for this leg only, then restored)
(cd oracle && gcc -O2 -fPIC -shared -DKO_SUPPORT_SKEW_OVERRIDE=1e-5 -o libko_oracle.so ko_model.c ko_physics.c ko_env.c -lm)
KS_LIB=$GRAFT_REPO_ROOT/kinovagrasping_amd/libkinova_sim_skew5.so python -m tests.studies.long_horizon > $out/long_horizon_skew1e-5.txt 2>&1
make -C oracle -s -B
tail -3 $out/gputests.log
