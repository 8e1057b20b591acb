#!/bin/bash
# which rocprofv3 counter-collection forms survive (kernel filter x HIP graphs): the ones that segfault are why tools/pmc_run.sh ddpg uses --eager
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_probe
rocprofv3 --kernel-trace --kernel-include-regex k_env_step --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_probe/a -- python3 bench.py --mode sim --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_probe/a.log 2>&1; echo "sim+regex rc $?"
rocprofv3 --kernel-trace --kernel-include-regex k_env_step --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_probe/b -- python3 bench.py --eager --steps 10 --warmup 2 --no-cpu-baseline --pretrain-updates 40 --steady-steps 0 > gpurun_out/pmc_probe/b.log 2>&1; echo "eager+regex rc $?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_probe/c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --pretrain-updates 40 --steady-steps 0 > gpurun_out/pmc_probe/c.log 2>&1; echo "graphs no-regex rc $?"
ls gpurun_out/pmc_probe/*/*/ 2>/dev/null | head; rm -rf gpurun_out/pmc_probe/a gpurun_out/pmc_probe/b gpurun_out/pmc_probe/c
