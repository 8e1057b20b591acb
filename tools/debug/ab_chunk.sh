#!/bin/bash
# free-running rollout: env-steps per launch (A/B on one box)
cd $GRAFT_REPO_ROOT
show() { grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], (d.get('steady_state') or {}).get('value'), d['config']['launch'][:90], d['config']['free_running'])"; }
for rep in 1 2; do
for c in 10 30 60; do
  echo "chunk $c default:"; python3 bench.py --no-cpu-baseline --chunk $c 2>/dev/null | show
  echo "chunk $c driver form:"; python3 bench.py --no-cpu-baseline --chunk $c --steps 20 --warmup 5 2>/dev/null | show
done; done
