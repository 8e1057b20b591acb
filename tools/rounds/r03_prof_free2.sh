#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_b
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/pmc_run.sh free gpurun_out/pmc_free > $out/pmc_free_summary.txt 2>&1
PMC_PRETRAIN=300 timeout 1500 bash tools/pmc_run.sh free gpurun_out/pmc_free300 > $out/pmc_free300_summary.txt 2>&1
tail -34 $out/pmc_free_summary.txt; tail -5 $out/pmc_free300_summary.txt
