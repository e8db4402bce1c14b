// The reference's Program.main for the render path, written against the C++ host mirror: build the Scene with the
// default scene_spheres, renderScene(), write frameBuffer / depthBuffer / objectIds / maxFrameBufferDepth / rayCount.
// usage: render_scene <width> <height> <spp> <maxTraceDepth> <jitter> <out.bin> [shares]
// shares > 1: that many scene handles on device 0, the frame split into interleaved columns (Scene::devices, rt_render_multi);
// the frame is rendered twice on the same handles (the scene is uploaded once).
#include <cstdio>
#include <cstdlib>

#include "rt_host.hpp"

int main(int argc, char** argv)
{
    if (argc != 7 && argc != 8) { fprintf(stderr, "usage: render_scene w h spp depth jitter out.bin [shares]\n"); return 2; }
    raytracer::Scene scene;
    scene.width = atoi(argv[1]); scene.height = atoi(argv[2]); scene.samplesPerPixel = atoi(argv[3]);
    scene.maxTraceDepth = atoi(argv[4]); scene.jitter = atoi(argv[5]);
    if (argc == 8) scene.devices.assign((size_t)atoi(argv[7]), 0);
    try {
        scene.renderScene();
        scene.renderScene();
    } catch (const std::exception& e) {
        fprintf(stderr, "renderScene failed: %s\n", e.what());
        return 1;
    }
    FILE* f = fopen(argv[6], "wb");
    if (!f) return 3;
    fwrite(scene.frameBuffer.data(), sizeof(float), scene.frameBuffer.size(), f);
    fwrite(scene.depthBuffer.data(), sizeof(float), scene.depthBuffer.size(), f);
    fwrite(scene.objectIds.data(), sizeof(uint64_t), scene.objectIds.size(), f);
    fwrite(&scene.maxFrameBufferDepth, sizeof(float), 1, f);
    fwrite(&scene.rayCount, sizeof(uint64_t), 1, f);
    fclose(f);
    return 0;
}
