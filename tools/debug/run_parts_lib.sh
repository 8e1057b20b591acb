#!/bin/bash
# bash tools/debug/run_parts_lib.sh LIBNAME [chunk]: async_parts.py with kinovagrasping_amd/libkinova_sim_LIBNAME.so
cd $GRAFT_REPO_ROOT
KS_LIB=$PWD/kinovagrasping_amd/libkinova_sim_$1.so python3 tools/debug/async_parts.py ${2:-30}
