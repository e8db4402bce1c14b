"""Comparisons with the reference's OWN outputs: the JPEGs its authors rendered with the F# renderer and checked in
(/root/reference/*.jpg, written by Scene.saveFrameBuffer / Scene.saveDepthBuffer, Scene.fs:123-144).  They are the
only reference-produced data that exist (the reference has no tests and cannot be run here), committed as decoded
pixel data under tests/golden/ref_* by tests/golden/make_golden.py.

Thresholds (measured with the oracle in the build container, recorded here so a regression is visible):
  depth.jpg        800x640 luminance, full resolution : 48.3 dB, miss mask (white) agreement 99.90 %   -> >= 45 dB, >= 99.8 %
  16_4_sphere.jpg  16 spp, maxTraceDepth 4            : 40.1 dB after an 8x8 box filter                   -> >= 38 dB
  display.jpg      (16 spp, maxTraceDepth 5)          : 40.3 dB                                           -> >= 38 dB
  16_6_sphere.jpg  16 spp, maxTraceDepth 6            : 40.7 dB                                           -> >= 38 dB
  4_6_sphere.jpg   4 spp,  maxTraceDepth 6            : 34.1 dB                                           -> >= 32 dB
The residual is Monte-Carlo noise (the reference used Henzai.Core's RNG, this build Philox) plus JPEG loss; the
box filter averages 64 pixels of either.  Pinhole primary rays: the reference's sub-pixel offset is < 1/W of a pixel
(Camera.fs:8-17), invisible at this resolution.
"""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (samplesPerPixel, maxTraceDepth, PSNR floor in dB)
COLOUR_FIXTURES = {
    "16_4_sphere": (16, 4, 38.0),
    "display": (16, 5, 38.0),
    "16_6_sphere": (16, 6, 38.0),
    "4_6_sphere": (4, 6, 32.0),
}
DEPTH_PSNR_MIN, DEPTH_MASK_MIN = 45.0, 0.998


def rgba32_pack(v):
    """ImageSharp 1.0.0-beta0004 `Rgba32(Vector4)` (Scene.fs:128,140): v*255 + 0.5, clamped to [0,255], truncated."""
    v = np.asarray(v, dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        s = v * np.float32(255.0) + np.float32(0.5)
    s = np.where(np.isnan(s), np.float32(0.0), s)
    return np.clip(s, 0.0, 255.0).astype(np.uint8)


def frame_image(rgba):
    """Scene.saveFrameBuffer (Scene.fs:123-132) up to the encoder: [W,H,4] f32 -> [H,W,4] u8 (image.Item(px,py))."""
    return rgba32_pack(rgba).transpose(1, 0, 2)


def depth_image(depth, max_depth):
    """Scene.saveDepthBuffer (Scene.fs:134-144): Vector4(Vector3(depth/maxFrameBufferDepth), 1) -> Rgba32; a miss is
    Single.MaxValue / max = huge -> white."""
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        g = np.asarray(depth, dtype=np.float32) / np.float32(max_depth)
    rgba = np.stack([g, g, g, np.ones_like(g)], axis=-1)
    return rgba32_pack(rgba).transpose(1, 0, 2)


def psnr255(a, b):
    mse = np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)
    return 10.0 * np.log10(255.0 ** 2 / mse)


def box8(img):
    """[640,800,C] -> [80,100,C] means of 8x8 blocks."""
    h, w = img.shape[:2]
    return np.asarray(img, dtype=np.float64).reshape(h // 8, 8, w // 8, 8, -1).mean(axis=(1, 3))


def check_depth_image(img_u8):
    """img_u8 [640,800,4] from depth_image / Scene.depthBufferImage against the reference's depth.jpg."""
    want = np.load(os.path.join(G, "ref_depth_800x640.npz"))["luminance"]
    assert img_u8.shape == (640, 800, 4) and np.all(img_u8[:, :, 3] == 255)
    assert np.array_equal(img_u8[:, :, 0], img_u8[:, :, 1]) and np.array_equal(img_u8[:, :, 0], img_u8[:, :, 2])
    got = img_u8[:, :, 0]
    p = psnr255(got, want)
    agree = float(np.mean((want >= 248) == (got == 255)))       # white = miss (Scene.fs:29,139)
    assert p >= DEPTH_PSNR_MIN and agree >= DEPTH_MASK_MIN, (p, agree)
    return p, agree


def check_colour_image(name, img_u8):
    """img_u8 [640,800,4] from frame_image / Scene.frameBufferImage against the reference's <name>.jpg."""
    want = np.load(os.path.join(G, "ref_%s_100x80.npy" % name))
    assert img_u8.shape == (640, 800, 4) and np.all(img_u8[:, :, 3] == 255)
    got = box8(img_u8[:, :, :3])
    p = psnr255(got, want)
    corr = float(np.corrcoef(got.ravel(), want.ravel())[0, 1])
    assert p >= COLOUR_FIXTURES[name][2] and corr > 0.995, (name, p, corr)
    return p, corr
