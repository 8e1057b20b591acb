#!/bin/bash
# does the rocprofv3 kernel trace agree with the bench line's HIP-event figure?  (pacing kernels on / off)
cd /tmp && export TMPDIR=/tmp
for lead in 8 -1; do
  export KS_ASYNC_LEAD=$lead
  rm -rf /tmp/prof_gap; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gap -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steady-steps 60 > /tmp/prof_gap.log 2>&1
  echo "KS_ASYNC_LEAD=$lead:"; python3 $GRAFT_REPO_ROOT/tools/rollout_trace_join.py /tmp/prof_gap/g_kernel_trace.csv /tmp/prof_gap.log | sed -n 2,3p
done
