// saveFrameBuffer / saveDepthBuffer of the C++ host mirror (Scene.fs:123-144) on buffers handed in by the test:
// usage: save_images <in.bin> <frame.ppm> <depth.ppm>
// in.bin: int32 width, int32 height, float maxFrameBufferDepth, frameBuffer [w][h][4] f32, depthBuffer [w][h] f32
#include <cstdio>
#include <cstdlib>

#include "rt_host.hpp"

int main(int argc, char** argv)
{
    if (argc != 4) { fprintf(stderr, "usage: save_images in.bin frame.ppm depth.ppm\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    raytracer::Scene scene;
    int32_t wh[2];
    if (fread(wh, sizeof(int32_t), 2, f) != 2 || fread(&scene.maxFrameBufferDepth, sizeof(float), 1, f) != 1) return 4;
    scene.width = wh[0]; scene.height = wh[1];
    const size_t n = (size_t)scene.width * scene.height;
    scene.frameBuffer.resize(n * 4); scene.depthBuffer.resize(n);
    if (fread(scene.frameBuffer.data(), sizeof(float), n * 4, f) != n * 4 || fread(scene.depthBuffer.data(), sizeof(float), n, f) != n) return 4;
    fclose(f);
    try {
        scene.saveFrameBuffer(argv[2]);
        scene.saveDepthBuffer(argv[3]);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
