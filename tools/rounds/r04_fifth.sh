#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/debug/overlap_probe.py 4096 cubes > $out/overlap_4096_cubes.txt 2>&1
python tools/debug/overlap_probe.py 4096 mixed 3 > $out/overlap_4096_mixed3.txt 2>&1
python tools/debug/overlap_probe.py 4080 mixed 3 > $out/overlap_4080_mixed3.txt 2>&1
for f in overlap_4096_cubes overlap_4096_mixed3 overlap_4080_mixed3; do echo "== $f"; grep "done" $out/$f.txt; done
