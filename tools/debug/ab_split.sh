#!/bin/bash
# experiment (round 2): the learner kernels one wave / split over 2 / 4 waves (KS_MLP_SPLIT), alternating runs of the bench on one box
for val in 0 unset 2 4 0 unset; do
  if [ $val = unset ]; then unset KS_MLP_SPLIT; else export KS_MLP_SPLIT=$val; fi
  python bench.py --no-cpu-baseline --steady-updates 600 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split=$val', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['steady_state']['value'], 'update alone ms', d['mfma']['update_ms_alone'], d['nonfinite_envs'])"
done
