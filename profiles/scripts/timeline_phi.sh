#!/bin/bash
cd "$GRAFT_REPO_ROOT"
C=signnet_basisnet_amd/csrc
for flags in "$@"; do
  fl="$flags"; [ "$fl" = "base" ] && fl=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSN_TIMELINE $fl -c $C/fused_phi.hip -o $C/fused_phi.o || { echo "build failed: $flags"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o signnet_basisnet_amd/libsignnet_hip.so $C/*.o
  echo "=== TIMELINE $flags"
  python profiles/scripts/timeline_phi.py 2>&1 | grep -v amdgpu.ids
done
