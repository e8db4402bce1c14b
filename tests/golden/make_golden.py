#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.

  oracle_c1_64x48.npz      oracle render: scene_spheres, 64x48, 1 spp, maxTraceDepth 5, pinhole
  oracle_c2_48x36.npz      oracle render: scene_spheres, 48x36, 4 spp, maxTraceDepth 4, jitter, seed 7
  oracle_ref_40x30_hi.npz  oracle render: 40x30, 32768 spp, maxTraceDepth 4, pinhole, seed 1001 (the high-spp
                           reference of the PSNR check; ~1.5 minutes on 8 cores)
  display_100x80.npy       the reference's own render display.jpg (800x640) box-filtered 8x8 to 100x80 RGB
                           u8 -- DATA from /root/reference (a soft visual fixture)

The first three pin the oracle against accidental change (they are produced BY the oracle: the
reference cannot be run here, see DESIGN.md "Oracle"); GPU tests compare the HIP path with them too,
so the GPU box needs neither /root/reference nor a long CPU render.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import oracle_lib as O  # noqa: E402
from raytracer_amd import host  # noqa: E402


def save(name, r, **meta):
    np.savez_compressed(os.path.join(HERE, name), rgba=r["rgba"], depth=r["depth"], ids=r["ids"],
                        max_depth=np.float32(r["max_depth"]), rays=np.uint64(r["rays"]), **meta)


def main(which):
    cam = host.make_camera()
    sc = host.SceneDefinitions.scene_spheres
    if "c1" in which:
        save("oracle_c1_64x48.npz", O.render(sc, cam, host.make_params(64, 48, 1, 5, jitter=0, seed=1)))
    if "c2" in which:
        save("oracle_c2_48x36.npz", O.render(sc, cam, host.make_params(48, 36, 4, 4, jitter=1, seed=7)))
    if "hi" in which:
        r = O.render(sc, cam, host.make_params(40, 30, 32768, 4, jitter=0, seed=1001))
        np.savez_compressed(os.path.join(HERE, "oracle_ref_40x30_hi.npz"), rgba=r["rgba"], rays=np.uint64(r["rays"]))
    if "display" in which:
        from PIL import Image
        im = np.asarray(Image.open("/root/reference/display.jpg").convert("RGB"), dtype=np.float64)   # [640, 800, 3]
        small = im.reshape(80, 8, 100, 8, 3).mean(axis=(1, 3))
        np.save(os.path.join(HERE, "display_100x80.npy"), np.round(small).astype(np.uint8))


if __name__ == "__main__":
    main(sys.argv[1:] or ["c1", "c2", "hi", "display"])
