"""GPU: the HIP render loop, through the C ABI and the image stage of the host mirror
(`Scene.frameBufferImage` / `Scene.depthBufferImage` = Scene.saveFrameBuffer / saveDepthBuffer up to the encoder,
Scene.fs:123-144), against the JPEGs the reference's authors rendered with the F# renderer (tests/refimg.py).
BASELINE.json configs[1] names two of them: 16_4_sphere.jpg and depth.jpg."""
import numpy as np
import pytest

import refimg
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


def test_depth_buffer_image_matches_reference_depth_jpg():
    s = Scene(width=800, height=640, samplesPerPixel=1, maxTraceDepth=0, jitter=0, seed=1)
    s.renderScene()
    img = s.depthBufferImage()
    refimg.check_depth_image(img)
    # the product's image stage equals the restatement of Rgba32(Vector4) in the test helper
    assert np.array_equal(img, refimg.depth_image(s.depthBuffer, s.maxFrameBufferDepth))


@pytest.mark.parametrize("name", sorted(refimg.COLOUR_FIXTURES))
def test_frame_buffer_image_matches_reference_colour_jpg(name):
    spp, depth, _ = refimg.COLOUR_FIXTURES[name]
    s = Scene(width=800, height=640, samplesPerPixel=spp, maxTraceDepth=depth, jitter=0, seed=1)
    s.renderScene()
    img = s.frameBufferImage()
    refimg.check_colour_image(name, img)
    assert np.array_equal(img, refimg.frame_image(s.frameBuffer))


def test_jittered_primary_rays_match_the_reference_images_too():
    """The reference's own primary rays carry a sub-pixel offset (Camera.fs:8-17); with it on, the images agree as well."""
    s = Scene(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4, jitter=1, seed=3)
    s.renderScene()
    refimg.check_colour_image("16_4_sphere", s.frameBufferImage())
    d = Scene(width=800, height=640, samplesPerPixel=1, maxTraceDepth=0, jitter=1, seed=3)
    d.renderScene()
    refimg.check_depth_image(d.depthBufferImage())
