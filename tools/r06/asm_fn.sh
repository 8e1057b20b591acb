#!/bin/bash
# device assembly of ks_api.hip with the given -D flags; prints instruction-class counts of the function whose mangled name contains $1
pat=$1; shift
cd "$(dirname "$0")/../../kinovagrasping_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" --cuda-device-only -S -o /tmp/ks_api_dev.s ks_api.hip 2>&1 | grep -v "warning\|^$" | tail -3
a=$(grep -n "Begin function .*$pat" /tmp/ks_api_dev.s | head -1 | cut -d: -f1)
n=$(tail -n +$a /tmp/ks_api_dev.s | grep -n "^.Lfunc_end" | head -1 | cut -d: -f1)
tail -n +$a /tmp/ks_api_dev.s | head -n $n > /tmp/fn.s
echo "lines $(wc -l < /tmp/fn.s) flat $(grep -c 'flat_load\|flat_store' /tmp/fn.s) scratch $(grep -c 'scratch_' /tmp/fn.s) ds $(grep -c 'ds_read\|ds_load\|ds_write\|ds_store' /tmp/fn.s) global $(grep -c 'global_load\|global_store' /tmp/fn.s) waitcnt $(grep -c s_waitcnt /tmp/fn.s) f64 $(grep -c '_f64' /tmp/fn.s)"
grep "$pat.*num_vgpr\|$pat.*private_seg_size" /tmp/ks_api_dev.s | head -4
