// Writes what the C++ host mirror (include/rt_host.hpp) hands to the C ABI: every SceneDefinitions list flattened
// to rt_surface records, then the default camera.  tests/test_host_cpp.py compares the bytes with the Python mirror.
#include <cstdio>

#include "rt_host.hpp"

int main()
{
    const raytracer::SceneDefinitions sd;
    const std::vector<raytracer::Surface>* lists[] = { &sd.lights, &sd.lights_simple, &sd.spheres, &sd.spheres_submission,
                                                       &sd.spheres_simple, &sd.spheres_caustics, &sd.spheres_scene_2, &sd.planes,
                                                       &sd.planes_submission, &sd.planes_simple, &sd.planes_scene_2, &sd.scene_spheres };
    for (const auto* l : lists) {
        const std::vector<rt_surface> flat = raytracer::flatten(*l);
        fwrite(flat.data(), sizeof(rt_surface), flat.size(), stdout);
    }
    const rt_camera cam = raytracer::make_camera();
    fwrite(&cam, sizeof cam, 1, stdout);
    const raytracer::Hitable tri = raytracer::Triangle({ 0, 0, 0 }, { 1, 0, 0 }, { 0, 1, 0 });
    const rt_surface ts = raytracer::Lambertian(77, tri, raytracer::Material(raytracer::Rgba32::Red())).flatten();
    fwrite(&ts, sizeof ts, 1, stdout);
    return 0;
}
