#!/bin/bash
# Dev tool: the kernel source (host lane-check build, one lane = the serial algorithm) under sanitizers on the CPU.
#   1. ASan + UBSan: the CPU lane tests with a sanitized libks_lanecheck.so
#   2. MSan (uninitialised reads): a closing grasp + lift driver, fp32 and fp64, cube and vase
# GPU sanitizers are not available on the pool; this covers everything that is not DPP / LDS specific.
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
from kinovagrasping_amd import scenarios
open('/tmp/ks_cube.blob', 'wb').write(scenarios.model_blob("CubeS"))
open('/tmp/ks_vase.blob', 'wb').write(scenarios.model_blob("Vase2B"))
from kinovagrasping_amd import model_compiler as mc
with open('/tmp/ks_mg_args', 'w') as f:
    for sh in ("BottleS", "BowlS"):
        open(f'/tmp/ks_{sh}.blob', 'wb').write(scenarios.model_blob(sh))
        g = mc.read_blob(scenarios.model_blob(sh))["geom_pos"][8]
        f.write(f"/tmp/ks_{sh}.blob {-g[0]} {-g[1]} 0.0\n")
PY
(cd tools/sanitize && /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=memory -fsanitize-memory-track-origins -fno-omit-frame-pointer -o /tmp/ks_msan msan_driver.cpp)
/tmp/ks_msan /tmp/ks_cube.blob | tail -2
/tmp/ks_msan /tmp/ks_vase.blob | tail -2
# the multi-geom capacities (-DKS_MULTI_GEOM, what libkinova_sim_mg.so is built with) on a bottle and a bowl placed in the hand
(cd tools/sanitize && /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -DKS_MULTI_GEOM -fsanitize=memory -fsanitize-memory-track-origins -fno-omit-frame-pointer -o /tmp/ks_msan_mg msan_driver.cpp)
while read -r line; do /tmp/ks_msan_mg $line | tail -2; done < /tmp/ks_mg_args
cp tests/native/libks_lanecheck.so /tmp/ks_lc_backup.so
(cd tests/native && g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o libks_lanecheck.so ks_lanecheck.cpp)
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_kernel_source_cpu.py -x -q | tail -2
cp /tmp/ks_lc_backup.so tests/native/libks_lanecheck.so
cp tests/native/libks_lanecheck_mg.so /tmp/ks_lc_mg_backup.so
(cd tests/native && g++ -O1 -g -std=c++17 -fPIC -shared -DKS_MULTI_GEOM -fsanitize=address,undefined -fno-omit-frame-pointer -o libks_lanecheck_mg.so ks_lanecheck.cpp)
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_multi_geom_cpu.py -x -q -k "kernel_source or env_step" | tail -2
cp /tmp/ks_lc_mg_backup.so tests/native/libks_lanecheck_mg.so
echo "sanitizers: clean"
