#!/bin/bash
R=$GRAFT_REPO_ROOT
cp $R/raytracer_amd/csrc/librt_hip.so /tmp/tree.so
cp $R/tools/ab/librt_hip_noslack.so $R/raytracer_amd/csrc/librt_hip.so
timeout -k 10 200 python3 tools/c5_check.py render /tmp/a.npz || exit 1
cp /tmp/tree.so $R/raytracer_amd/csrc/librt_hip.so
timeout -k 10 200 python3 tools/c5_check.py render /tmp/b.npz || exit 1
timeout -k 10 1000 python3 -u tools/c5_check.py compare /tmp/a.npz /tmp/b.npz
