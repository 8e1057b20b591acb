mkdir -p gpurun_out/r06s2
o=gpurun_out/r06s2/ab2.txt; : > $o
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), d['ms_per_step'], d.get('timed_windows_ms_per_step'), round(d['steady_state']['value']/1e6,3))"; }
for i in 1 2; do
echo "phase=2 driver: $(KS_ROLLOUT_PHASE_DEAL=2 run --steps 20 --warmup 5)" >> $o
echo "phase=0 driver: $(KS_ROLLOUT_PHASE_DEAL=0 run --steps 20 --warmup 5)" >> $o
echo "phase=1 driver: $(run --steps 20 --warmup 5)" >> $o
done
echo "phase=0 default: $(KS_ROLLOUT_PHASE_DEAL=0 run)" >> $o
echo "phase=1 default: $(run)" >> $o
echo "phase=2 default: $(KS_ROLLOUT_PHASE_DEAL=2 run)" >> $o
echo "phase=0 default: $(KS_ROLLOUT_PHASE_DEAL=0 run)" >> $o
echo "phase=1 default: $(run)" >> $o
echo "phase=2 default: $(KS_ROLLOUT_PHASE_DEAL=2 run)" >> $o
python -m pytest tests/test_gpu_async.py -m gpu -x -q > gpurun_out/r06s2/async.txt 2>&1; tail -3 gpurun_out/r06s2/async.txt >> $o
cat $o
