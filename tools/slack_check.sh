#!/bin/bash
# does the far-origin test fail without the per-ray slack (library of the previous commit) and pass with it?
R=$GRAFT_REPO_ROOT
cp $R/raytracer_amd/csrc/librt_hip.so /tmp/tree.so
cp $R/tools/ab/librt_hip_noslack.so $R/raytracer_amd/csrc/librt_hip.so
timeout -k 10 200 python -u -m pytest tests/test_gpu_bvh.py -m gpu -q -k far_origins --timeout 120 --timeout-method thread 2>&1 | tail -3
cp /tmp/tree.so $R/raytracer_amd/csrc/librt_hip.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_bvh.py -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -3
