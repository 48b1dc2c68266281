#!/bin/bash
# cycle stamps of the one-launch batch plan's workgroups: gpurun -- bash profiles/scripts/prof_plan.sh   (rebuilds the default library afterwards)
cd "${GRAFT_REPO_ROOT:-.}"
C=signnet_basisnet_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSN_PROFILE -c $C/plan.hip -o $C/plan.o && hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
python profiles/scripts/prof_plan.py 2>&1 | grep -v amdgpu.ids
python -m signnet_basisnet_amd.build --force > /dev/null 2>&1
