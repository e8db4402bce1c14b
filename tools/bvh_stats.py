"""tools/bvh_stats.py <c3|c5> <out.bin>: the primitive boxes of a bench scene for tools/bvh_stats.cpp"""
import sys, numpy as np
sys.path.insert(0, '.')
from raytracer_amd import scenes, abi
wl, out = sys.argv[1], sys.argv[2]
m = scenes.heightfield_mesh(224) if wl == 'c3' else scenes.many_spheres(1_000_000)
tri = m[m["geom"] == abi.GEOM_TRIANGLE]; sph = m[m["geom"] == abi.GEOM_SPHERE]
boxes = []
if len(tri):
    v = tri["g"][:, :9].reshape(-1, 3, 3); boxes.append(np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1))
if len(sph):
    c = sph["g"][:, :3]; r = np.abs(sph["g"][:, 3:4]); boxes.append(np.concatenate([c - r, c + r], axis=1))
np.concatenate(boxes).astype(np.float32).tofile(out)
