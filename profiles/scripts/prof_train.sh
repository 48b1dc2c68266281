#!/bin/bash
# cycles per phase of the training link kernels: gpurun -- bash profiles/scripts/prof_train.sh   (rebuilds the default library afterwards)
cd "${GRAFT_REPO_ROOT:-.}"
C=signnet_basisnet_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSN_PROFILE -c $C/train.hip -o $C/train.o && hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
python profiles/scripts/prof_train.py 2>&1 | grep -v amdgpu.ids | tail -5
python -m signnet_basisnet_amd.build --force > /dev/null 2>&1
