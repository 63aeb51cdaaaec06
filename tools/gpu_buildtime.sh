#!/bin/bash
# Stage times of rtx_scene_create (RTX_BUILD_TIMING) for the sponza stand-in, device-side and host-side hierarchy build, cold + warm.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for hb in device host; do
  echo "== RTX_BUILD_HIERARCHY=$hb"
  RTX_BUILD_HIERARCHY=$hb RTX_BUILD_TIMING=1 timeout -k 10 120 python tools/gpu_buildtime.py 2>&1 | grep -v amdgpu.ids || exit 1
done
