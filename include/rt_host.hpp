// rt_host.hpp — C++ host-side mirror of the reference's scene API for the render path, above the C ABI
// (include/rt_abi.h).  The reference is compiled F# (no .NET toolchain in this image), so the compiled-language
// host is C++: same names, same argument meaning, same error behaviour as the F# for everything renderScene
// touches.  Header-only; link with librt_hip.so only if Scene::renderScene is used.
//
//   Material (Material.fs:9-18)            raytracer::Material
//   Sphere / Plane (Geometry/*.fs)         raytracer::Sphere, raytracer::Plane (+ Triangle, this build's own)
//   NoSurface/Lambertian/Metal/Dielectric  raytracer::Surface factories (Surface.fs:44,60,85,112)
//   SceneDefinitions.fs                    raytracer::SceneDefinitions (module-initialisation order, so IDs match)
//   WorldToCamera / CameraToWorld          raytracer::WorldToCamera / CameraToWorld (Camera.fs:41-45)
//   Scene (Scene.fs:16-144)                raytracer::Scene: renderScene(), frameBuffer, depthBuffer,
//                                          maxFrameBufferDepth
//
// All arithmetic is float32 with one rounding per operation in System.Numerics' order: compile with
// -ffp-contract=off (tests/test_host_cpp.py checks the flattened scene bit for bit against the Python mirror).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_abi.h"

namespace raytracer {

// ---- System.Numerics subset -------------------------------------------------------------------------------
struct Vector3 {
    float X = 0, Y = 0, Z = 0;
    Vector3() = default;
    Vector3(float x, float y, float z) : X(x), Y(y), Z(z) {}
    static float Dot(Vector3 a, Vector3 b) { return (a.X * b.X + a.Y * b.Y) + a.Z * b.Z; }
    static Vector3 Cross(Vector3 a, Vector3 b) { return { a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X }; }
    float Length() const { return std::sqrt(Dot(*this, *this)); }
    static Vector3 Normalize(Vector3 a) { return a / a.Length(); }
    friend Vector3 operator-(Vector3 a, Vector3 b) { return { a.X - b.X, a.Y - b.Y, a.Z - b.Z }; }
    friend Vector3 operator*(Vector3 a, float s) { return { a.X * s, a.Y * s, a.Z * s }; }
    friend Vector3 operator*(float s, Vector3 a) { return a * s; }
    friend Vector3 operator/(Vector3 a, float s) { const float inv = 1.0f / s; return a * inv; }   // multiplies by the reciprocal
    Vector3 operator-() const { return { -X, -Y, -Z }; }
};

struct NumericsPlane {      // System.Numerics.Plane
    Vector3 Normal; float D = 0;
    NumericsPlane() = default;
    NumericsPlane(Vector3 n, float d) : Normal(n), D(d) {}
    static NumericsPlane CreateFromVertices(Vector3 p1, Vector3 p2, Vector3 p3)
    {
        const Vector3 n = Vector3::Normalize(Vector3::Cross(p2 - p1, p3 - p1));
        return { n, -Vector3::Dot(n, p1) };
    }
};

struct Matrix4x4 {
    float m[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };   // M11..M44, row-major
    static Matrix4x4 CreateLookAt(Vector3 position, Vector3 target, Vector3 up)
    {
        const Vector3 z = Vector3::Normalize(position - target);
        const Vector3 x = Vector3::Normalize(Vector3::Cross(up, z));
        const Vector3 y = Vector3::Cross(z, x);
        Matrix4x4 r;
        const float v[16] = { x.X, y.X, z.X, 0, x.Y, y.Y, z.Y, 0, x.Z, y.Z, z.Z, 0,
                              -Vector3::Dot(x, position), -Vector3::Dot(y, position), -Vector3::Dot(z, position), 1 };
        std::memcpy(r.m, v, sizeof v);
        return r;
    }
    // Matrix4x4.Invert: cofactor expansion in float32
    static bool Invert(const Matrix4x4& M, Matrix4x4& out)
    {
        const float a = M.m[0], b = M.m[1], c = M.m[2], d = M.m[3], e = M.m[4], f = M.m[5], g = M.m[6], h = M.m[7];
        const float i = M.m[8], j = M.m[9], k = M.m[10], l = M.m[11], mm = M.m[12], n = M.m[13], o = M.m[14], p = M.m[15];
        const float kp_lo = k * p - l * o, jp_ln = j * p - l * n, jo_kn = j * o - k * n;
        const float ip_lm = i * p - l * mm, io_km = i * o - k * mm, in_jm = i * n - j * mm;
        const float a11 = +(f * kp_lo - g * jp_ln + h * jo_kn), a12 = -(e * kp_lo - g * ip_lm + h * io_km);
        const float a13 = +(e * jp_ln - f * ip_lm + h * in_jm), a14 = -(e * jo_kn - f * io_km + g * in_jm);
        const float det = a * a11 + b * a12 + c * a13 + d * a14;
        if (std::fabs(det) < std::numeric_limits<float>::denorm_min()) return false;
        const float inv = 1.0f / det;
        float* r = out.m;
        r[0] = a11 * inv; r[4] = a12 * inv; r[8] = a13 * inv; r[12] = a14 * inv;
        r[1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv;
        r[5] = +(a * kp_lo - c * ip_lm + d * io_km) * inv;
        r[9] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv;
        r[13] = +(a * jo_kn - b * io_km + c * in_jm) * inv;
        const float gp_ho = g * p - h * o, fp_hn = f * p - h * n, fo_gn = f * o - g * n;
        const float ep_hm = e * p - h * mm, eo_gm = e * o - g * mm, en_fm = e * n - f * mm;
        r[2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv;
        r[6] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv;
        r[10] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv;
        r[14] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv;
        const float gl_hk = g * l - h * k, fl_hj = f * l - h * j, fk_gj = f * k - g * j;
        const float el_hi = e * l - h * i, ek_gi = e * k - g * i, ej_fi = e * j - f * i;
        r[3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv;
        r[7] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv;
        r[11] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv;
        r[15] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv;
        return true;
    }
};

inline Vector3 SurfaceNormal(float a, float b, float c) { return Vector3::Normalize({ a, b, c }); }   // Numerics.fs:28
inline Matrix4x4 WorldToCamera(Vector3 position, Vector3 target, Vector3 up) { return Matrix4x4::CreateLookAt(position, target, up); }   // Camera.fs:41
inline Matrix4x4 CameraToWorld(const Matrix4x4& worldToCamera)                                        // Camera.fs:43-45
{
    Matrix4x4 inv;
    if (!Matrix4x4::Invert(worldToCamera, inv)) throw std::runtime_error("Matrix4x4 Invert Failed");
    return inv;
}

// ---- Material.fs (ImageSharp named colours, bytes / 255) ------------------------------------------------------
struct Rgba32 {
    uint8_t R, G, B;
    Vector3 ToVector3() const { return { R / 255.0f, G / 255.0f, B / 255.0f }; }
    static constexpr Rgba32 White() { return { 255, 255, 255 }; }
    static constexpr Rgba32 Red() { return { 255, 0, 0 }; }
    static constexpr Rgba32 DarkGreen() { return { 0, 100, 0 }; }
    static constexpr Rgba32 RoyalBlue() { return { 65, 105, 225 }; }
    static constexpr Rgba32 AntiqueWhite() { return { 250, 235, 215 }; }
    static constexpr Rgba32 IndianRed() { return { 205, 92, 92 }; }
};

struct Material {           // Material.fs:9-18
    Vector3 Albedo, Emmitance;
    explicit Material(Vector3 color) : Albedo(color) {}
    explicit Material(Rgba32 color) : Albedo(color.ToVector3()) {}
    Material(Rgba32 color, float albedoFactor, Rgba32 emittance, float emittingFactor)
        : Albedo(albedoFactor * color.ToVector3()), Emmitance(emittingFactor * emittance.ToVector3()) {}
};

// ---- Geometry -------------------------------------------------------------------------------------------------
struct Hitable {            // Geometry/Core.fs:26: what a Surface needs of its geometry is its flat form
    int32_t geom = RT_GEOM_SPHERE;
    float g[12] = { 0 };
};
inline Hitable Sphere(Vector3 center, float radius)                                                   // Sphere.fs:8
{
    Hitable h; h.geom = RT_GEOM_SPHERE;
    h.g[0] = center.X; h.g[1] = center.Y; h.g[2] = center.Z; h.g[3] = radius;
    return h;
}
inline Hitable Plane(NumericsPlane plane)                                                             // Plane.fs:12, (plane, None, None, None)
{
    Hitable h; h.geom = RT_GEOM_PLANE;
    h.g[0] = plane.Normal.X; h.g[1] = plane.Normal.Y; h.g[2] = plane.Normal.Z; h.g[3] = plane.D;
    return h;
}
inline Hitable Plane(NumericsPlane plane, Vector3 center, float width, float height)                  // Plane.fs:12, bounded
{
    Hitable h = Plane(plane);
    h.g[4] = 1.0f; h.g[5] = center.X; h.g[6] = center.Y; h.g[7] = center.Z; h.g[8] = width; h.g[9] = height;
    return h;
}
inline Hitable Triangle(Vector3 v0, Vector3 v1, Vector3 v2)       // not in the reference; see RT_GEOM_TRIANGLE
{
    Hitable h; h.geom = RT_GEOM_TRIANGLE;
    const float v[9] = { v0.X, v0.Y, v0.Z, v1.X, v1.Y, v1.Z, v2.X, v2.Y, v2.Z };
    std::memcpy(h.g, v, sizeof v);
    return h;
}

// ---- Surface.fs -------------------------------------------------------------------------------------------------
struct Surface {
    uint64_t ID; Hitable Geometry; raytracer::Material Material;
    int32_t kind; float fuzz = 0, ior = 0; int32_t sampleCount = 0;
    rt_surface flatten() const
    {
        rt_surface s;
        std::memset(&s, 0, sizeof s);
        s.kind = kind; s.geom = Geometry.geom; s.id = ID;
        s.albedo[0] = Material.Albedo.X; s.albedo[1] = Material.Albedo.Y; s.albedo[2] = Material.Albedo.Z;
        s.emission[0] = Material.Emmitance.X; s.emission[1] = Material.Emmitance.Y; s.emission[2] = Material.Emmitance.Z;
        s.fuzz = fuzz; s.ior = ior; s.sample_count = sampleCount;
        std::memcpy(s.g, Geometry.g, sizeof s.g);
        return s;
    }
};
inline Surface NoSurface(uint64_t id, Hitable g, Material m) { return { id, g, m, RT_KIND_NOSURFACE }; }                       // Surface.fs:44
inline Surface Lambertian(uint64_t id, Hitable g, Material m) { return { id, g, m, RT_KIND_LAMBERTIAN }; }                     // Surface.fs:60
inline Surface Metal(uint64_t id, Hitable g, Material m, float fuzz) { return { id, g, m, RT_KIND_METAL, fuzz }; }             // Surface.fs:85
inline Surface Dielectric(uint64_t id, Hitable g, Material m, float refractiveIndex) { return { id, g, m, RT_KIND_DIELECTRIC, 0.0f, refractiveIndex }; }   // Surface.fs:112

// ---- SceneDefinitions.fs: every list, built in module-initialisation order so that the IDs match ----------------
struct SceneDefinitions {
    std::vector<Surface> lights, lights_simple, spheres, spheres_submission, spheres_simple, spheres_caustics,
        spheres_scene_2, planes, planes_submission, planes_simple, planes_scene_2, scene_spheres;

    SceneDefinitions()
    {
        uint64_t id = 1;
        auto nid = [&]() { return id++; };                            // assignIDAndIncrement, SceneDefinitions.fs:16-22
        const Rgba32 White = Rgba32::White();
        const Vector3 n111 = SurfaceNormal(-1.0f, -1.0f, 1.0f);
        const Material light(White, 0.0f, White, 1.0f);
        lights = { NoSurface(nid(), Plane(NumericsPlane(n111, 20.0f), n111 * -20.0f, 40.0f, 13.0f), light),
                   NoSurface(nid(), Plane(NumericsPlane(SurfaceNormal(1.0f, 0.0f, 0.0f), 15.0f), { -15.0f, 5.0f, -10.0f }, 20.0f, 10.0f), light) };
        lights_simple = { NoSurface(nid(), Sphere({ -15.0f, 5.0f, -10.0f }, 5.0f), light),
                          NoSurface(nid(), Sphere({ -5.0f, 11.0f, -13.0f }, 5.0f), light) };
        spheres = { Dielectric(nid(), Sphere({ 4.0f, -2.0f, -12.0f }, 2.0f), Material(White), 1.3f),
                    Metal(nid(), Sphere({ 3.0f, -1.0f, -19.0f }, 3.5f), Material(Rgba32::RoyalBlue()), 0.7f),
                    Metal(nid(), Sphere({ -1.5f, -4.5f, -7.0f }, 1.5f), Material(White), 0.0f),
                    Lambertian(nid(), Sphere({ 2.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::Red())),
                    Lambertian(nid(), Sphere({ 6.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::DarkGreen())),
                    Dielectric(nid(), Sphere({ 2.5f, -3.0f, -3.0f }, 1.5f), Material(White), 1.5f),
                    Dielectric(nid(), Sphere({ -5.1f, 2.0f, -11.0f }, 5.0f), Material(White), 1.5f) };
        spheres_submission = { Lambertian(nid(), Sphere({ 4.0f, -1.0f, -15.0f }, 3.5f), Material(Rgba32::RoyalBlue())),
                               Metal(nid(), Sphere({ -1.5f, -4.5f, -7.0f }, 1.5f), Material(White), 0.0f),
                               Lambertian(nid(), Sphere({ 2.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::Red())),
                               Lambertian(nid(), Sphere({ 6.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::DarkGreen())),
                               Dielectric(nid(), Sphere({ 2.5f, -3.0f, -3.0f }, 1.5f), Material(White), 1.5f),
                               Dielectric(nid(), Sphere({ -5.1f, 2.0f, -11.0f }, 5.0f), Material(White), 1.5f) };
        spheres_simple = { Lambertian(nid(), Sphere({ 4.0f, -1.0f, -15.0f }, 3.5f), Material(Rgba32::RoyalBlue())) };
        spheres_caustics = { Dielectric(nid(), Sphere({ -5.1f, -2.0f, -11.0f }, 3.0f), Material(White), 1.5f) };
        spheres_scene_2 = { Lambertian(nid(), Sphere({ 4.0f, -1.0f, -15.0f }, 3.5f), Material(Rgba32::RoyalBlue())),
                            Lambertian(nid(), Sphere({ 2.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::Red())),
                            Lambertian(nid(), Sphere({ 6.0f, -4.5f, -7.0f }, 1.5f), Material(Rgba32::DarkGreen())),
                            Metal(nid(), Sphere({ -1.5f, -4.5f, -7.0f }, 1.5f), Material(White), 0.0f),
                            Dielectric(nid(), Sphere({ -5.1f, -2.0f, -11.0f }, 3.0f), Material(White), 1.5f),
                            Dielectric(nid(), Sphere({ 2.5f, -3.0f, -3.0f }, 1.5f), Material(White), 1.5f) };
        auto floor = [&]() { return Lambertian(nid(), Plane(NumericsPlane::CreateFromVertices({ -1.0f, -6.0f, 0.0f }, { 1.0f, -6.0f, 0.0f }, { 0.0f, -6.0f, -1.0f })),
                                               Material(Vector3(1.0f, 1.0f, 1.0f))); };
        auto back = [&]() { return Lambertian(nid(), Plane(NumericsPlane(SurfaceNormal(0.0f, 0.0f, -1.0f), 17.0f), { 0.0f, 0.0f, 17.0f }, 30.0f, 10.0f), Material(White)); };
        auto red = [&]() { return Lambertian(nid(), Plane(NumericsPlane(SurfaceNormal(0.0f, 0.0f, 1.0f), 30.0f), { -5.0f, 10.0f, -30.0f }, 25.0f, 20.0f), Material(Rgba32::IndianRed())); };
        auto mirror = [&](float fuzz) { return Metal(nid(), Plane(NumericsPlane(SurfaceNormal(-1.0f, 0.0f, 1.0f), 15.0f), { 15.0f, 3.0f, -15.0f }, 10.0f, 6.0f), Material(Rgba32::AntiqueWhite()), fuzz); };
        planes = { floor(), back(), red(), mirror(0.01f) };
        planes_submission = { floor(), back(), red(), mirror(0.0f) };
        planes_simple = { floor() };
        planes_scene_2 = { floor(), back(), mirror(0.0f), red() };
        scene_spheres = spheres_scene_2;                                  // SceneDefinitions.fs:119
        scene_spheres.insert(scene_spheres.end(), planes_scene_2.begin(), planes_scene_2.end());
        scene_spheres.insert(scene_spheres.end(), lights.begin(), lights.end());
    }
};

inline std::vector<rt_surface> flatten(const std::vector<Surface>& surfaces)     // `scene_spheres |> Array.ofList`, Scene.fs:33
{
    std::vector<rt_surface> out;
    out.reserve(surfaces.size());
    for (const Surface& s : surfaces) out.push_back(s.flatten());
    return out;
}

inline rt_camera make_camera(Vector3 origin = { -3.0f, 6.0f, 15.0f }, Vector3 lookAt = { -1.0f, 1.0f, -10.0f },
                             float fov = 3.14159265358979323846f / 4.0f)            // Scene.fs:36-43
{
    rt_camera cam;
    const Matrix4x4 cw = CameraToWorld(WorldToCamera(origin, lookAt, { 0.0f, 1.0f, 0.0f }));
    std::memcpy(cam.cam_to_world, cw.m, sizeof cw.m);
    cam.look_at[0] = lookAt.X; cam.look_at[1] = lookAt.Y; cam.look_at[2] = lookAt.Z;
    cam.fov = fov;
    return cam;
}

// ---- Scene.fs:16-144 ---------------------------------------------------------------------------------------------
class Scene {
public:
    int width = 800, height = 640, samplesPerPixel = 1, maxTraceDepth = 5;      // Scene.fs:18-20,31
    Vector3 backgroundColor{ 0.0f, 0.0f, 0.0f };                                  // Scene.fs:32
    Vector3 cameraOriginWS{ -3.0f, 6.0f, 15.0f }, lookAt{ -1.0f, 1.0f, -10.0f };   // Scene.fs:36-37
    uint64_t seed = 1; int jitter = 1, device = 0;
    std::vector<int> devices;                // empty = {device}; several = one share of interleaved columns per device (rt_render_multi)
    int estimator = RT_ESTIMATOR_BRANCHING;  // RT_ESTIMATOR_SINGLE_PATH: one child per hit, Dielectric.Scatter (Surface.fs:155-166)
    int lambertianSamples = 0;               // 0 = Surface.fs:63 (4)
    std::vector<Surface> surfaces;                                                // Scene.fs:33
    std::vector<float> frameBuffer;        // [width][height] Vector4, Scene.fs:28
    std::vector<float> depthBuffer;        // [width][height], Scene.fs:29
    std::vector<uint64_t> objectIds;       // [width][height] Surface.ID of the primary hit
    float maxFrameBufferDepth = 0.0f;      // Scene.fs:30
    uint64_t rayCount = 0;

    explicit Scene(std::vector<Surface> s) : surfaces(std::move(s)) {}
    Scene() : surfaces(SceneDefinitions().scene_spheres) {}
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() { release(); }

    // `scene_spheres |> Array.ofList` (Scene.fs:33) happens ONCE: the flattened scene (and, for large scenes, the BVH
    // built inside rt_scene_create) is uploaded to the HBM of every device used and kept until the surfaces are
    // replaced (call release()) or the object dies; renderScene() only renders.
    void upload()
    {
        if (!handles_.empty()) return;
        const std::vector<rt_surface> flat = flatten(surfaces);
        const std::vector<int> devs = devices.empty() ? std::vector<int>{ device } : devices;
        for (int d : devs) {
            rt_scene* sc = nullptr;
            if (rt_scene_create(flat.data(), (int64_t)flat.size(), d, &sc) != RT_OK) {
                const std::string err = rt_last_error();
                release();
                throw std::runtime_error(err);
            }
            handles_.push_back(sc);
        }
    }
    void release()
    {
        for (rt_scene* sc : handles_) rt_scene_destroy(sc);
        handles_.clear();
    }

    // Scene.fs:117-121.  Throws like the reference's failwith on any error of the library.
    void renderScene()
    {
        upload();
        frameBuffer.assign((size_t)width * height * 4, 0.0f);
        depthBuffer.assign((size_t)width * height, std::numeric_limits<float>::max());
        objectIds.assign((size_t)width * height, 0);
        const rt_camera cam = make_camera(cameraOriginWS, lookAt);
        rt_params p;
        std::memset(&p, 0, sizeof p);
        p.width = width; p.height = height; p.spp = samplesPerPixel; p.max_trace_depth = maxTraceDepth;
        p.background[0] = backgroundColor.X; p.background[1] = backgroundColor.Y; p.background[2] = backgroundColor.Z;
        p.jitter = jitter; p.seed = seed; p.x_begin = 0; p.x_end = width; p.x_step = 1;
        p.estimator = estimator; p.lambertian_samples = lambertianSamples;
        const int rc = handles_.size() == 1
            ? rt_render(handles_[0], &cam, &p, frameBuffer.data(), depthBuffer.data(), objectIds.data(), &maxFrameBufferDepth, &rayCount)
            : rt_render_multi(handles_.data(), (int32_t)handles_.size(), &cam, &p, frameBuffer.data(), depthBuffer.data(), objectIds.data(),
                              &maxFrameBufferDepth, &rayCount);
        if (rc != RT_OK) throw std::runtime_error(rt_last_error());
    }

    // ---- Scene.fs:123-144: the image stage.  ImageSharp's JPEG encoder is replaced by binary PPM (P6); everything up
    // to the encoder is the reference's: image.Item(px,py) <- Rgba32(frameBuffer.[px,py]) and, for the depth image,
    // Rgba32(Vector4(Vector3(depth / maxFrameBufferDepth), 1)).
    // ImageSharp `Rgba32(Vector4)` packs a channel as v*255 + 0.5, clamped to [0,255], truncated to a byte.
    static uint8_t packChannel(float v)
    {
        float s = v * 255.0f + 0.5f;
        if (s != s) s = 0.0f;
        s = s < 0.0f ? 0.0f : (s > 255.0f ? 255.0f : s);
        return (uint8_t)s;
    }
    // [height][width][4] bytes, row py / column px: what saveFrameBuffer hands to the encoder (Scene.fs:125-130)
    std::vector<uint8_t> frameBufferImage() const
    {
        std::vector<uint8_t> img((size_t)width * height * 4);
        for (int px = 0; px < width; px++)
            for (int py = 0; py < height; py++)
                for (int c = 0; c < 4; c++)
                    img[((size_t)py * width + px) * 4 + c] = packChannel(frameBuffer[((size_t)px * height + py) * 4 + c]);
        return img;
    }
    // Scene.fs:136-142: depth / maxFrameBufferDepth in the three colour channels, alpha 1; a miss holds Single.MaxValue
    // (Scene.fs:29) and saturates to white
    std::vector<uint8_t> depthBufferImage() const
    {
        std::vector<uint8_t> img((size_t)width * height * 4);
        for (int px = 0; px < width; px++)
            for (int py = 0; py < height; py++) {
                const uint8_t g = packChannel(depthBuffer[(size_t)px * height + py] / maxFrameBufferDepth);
                uint8_t* o = &img[((size_t)py * width + px) * 4];
                o[0] = g; o[1] = g; o[2] = g; o[3] = packChannel(1.0f);
            }
        return img;
    }
    void saveFrameBuffer(const std::string& path = "sphere.ppm") const { writePpm(path, frameBufferImage()); }     // Scene.fs:123-132
    void saveDepthBuffer(const std::string& path = "depth.ppm") const { writePpm(path, depthBufferImage()); }      // Scene.fs:134-144

private:
    std::vector<rt_scene*> handles_;       // one per device, created by upload()

    void writePpm(const std::string& path, const std::vector<uint8_t>& rgba) const
    {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        std::fprintf(f, "P6\n%d %d\n255\n", width, height);
        std::vector<uint8_t> rgb((size_t)width * height * 3);
        for (size_t i = 0; i < (size_t)width * height; i++) { rgb[3 * i] = rgba[4 * i]; rgb[3 * i + 1] = rgba[4 * i + 1]; rgb[3 * i + 2] = rgba[4 * i + 2]; }
        const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
        std::fclose(f);
        if (!ok) throw std::runtime_error("short write to " + path);
    }
};

}  // namespace raytracer
