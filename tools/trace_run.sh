# Dev tool: kernel trace of the default bench on the GPU box -> gpurun_out/trace_*.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/tr.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_steps_summary.py /tmp/tr/tr_kernel_trace.csv > gpurun_out/trace_summary.txt 2>&1
python3 tools/trace_step.py /tmp/tr/tr_kernel_trace.csv > gpurun_out/trace_step.txt 2>&1
python3 tools/trace_critical_path.py /tmp/tr/tr_kernel_trace.csv > gpurun_out/trace_critical.txt 2>&1
