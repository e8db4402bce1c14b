/* rt_oracle.c — CPU restatement of geoeo/Raytracer's render loop.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle and the timed CPU baseline ("port").  Nothing in the product
 * (raytracer_amd/, include/) may call, link or import it; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do.
 *
 * HOW IT IS PINNED.  The reference (F#, .NET Core 2.0, raytracer/raytracer.fsproj:4) cannot be built
 * or run here (no .NET toolchain; its Henzai.Core project dependency, raytracer.fsproj:8, is not in
 * the tree) and it ships no tests or golden vectors for this path.  It does hold five images that its own
 * saveFrameBuffer / saveDepthBuffer (Scene.fs:123-144) wrote: depth.jpg, display.jpg, 16_4_sphere.jpg,
 * 16_6_sphere.jpg, 4_6_sphere.jpg.  tests/golden/make_golden.py turned them into data-only fixtures and
 *   - tests/test_golden.py holds THIS FILE to them on the CPU: the 800x640 pinhole depth buffer through the
 *     image stage against depth.jpg (>= 45 dB, hit/miss mask >= 99.8 %: measured 48.0 dB, 99.91 %), renders at
 *     the sample counts / trace depths the file names state against the four colour JPEGs (>= 38 dB on 8x8
 *     box-filtered RGB: measured 40.1-40.7 dB);
 *   - tests/test_gpu_reference_images.py holds the HIP frames to the same images.
 * That is a SOFT pin: lossy JPEGs with unrecorded parameters catch a wrong camera, root rule, rectangle test,
 * cull or estimator, not ulp-level choices.  Wherever a test says "bit-exact", it means bit-exact against this
 * restatement, which fills the gaps the images cannot see with stated assumptions (PARITY UNPINNED for these):
 *   - System.Numerics (.NET Core 2.0.5, hardware-accelerated path): Dot = (x*x'+y*y')+z*z';
 *     Length = sqrt(Dot(v,v)); `Vector3 / float` multiplies by 1.0f/float; Normalize =
 *     v / Length(v) through that operator; TransformNormal / Vector4.Transform /
 *     Matrix4x4.Multiply sum left to right; MathF.Round(x,d) = rint(x*10^d)/10^d for |x|<1e8.
 *   - MathF.Pow(x,2) = x*x (what a correctly rounded powf returns); MathF.Pow(x,5) =
 *     (float)(x^5 evaluated in double); MathF.Cos/Tan = this libm's cosf/tanf.
 *   - Henzai.Core (missing): RandomInUnitHemisphere = scratchapixel uniformSampleHemisphere
 *     (y up, y = r1 = cos theta; Scene.fs:1 and Surface.fs:111 cite those lessons),
 *     CreateCoordinateSystemAroundNormal = scratchapixel createCoordinateSystem,
 *     RandomInUnitSphere = rejection sampling in [-1,1]^3.  Random numbers come from the
 *     build's counter-based generator (Philox4x32-10, Random123 KATs in tests/test_oracle_kat.py), keyed by
 *     (seed; pixel-sample index, ray-tree node index), so the GPU can reproduce every draw.
 * Hand-derived known answers of the reference's formulas (tests/test_oracle_kat.py) and committed renders of this
 * file (tests/golden/oracle_*.npz) guard it against its own regressions.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#include "../include/rt_abi.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_FLAG_LITERAL_PLANE   1u /* recompute the Rodrigues matrix per test, Plane.fs:22-23 */
#define RTO_FLAG_LITERAL_LEAVES  2u /* scatter at the last level even when background is 0     */

/* ------------------------------------------------------------------------------------------
 * System.Numerics restatement
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* Vector3 operator/(Vector3, float): reciprocal then three multiplies */
static inline v3 v3_divs(v3 a, float s) { float inv = 1.0f / s; return V3(a.x * inv, a.y * inv, a.z * inv); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { return v3_divs(a, v3_length(a)); }

/* Matrix4x4, row-major M11..M44 */
typedef struct { float m[16]; } m4;

static m4 m4_identity(void)
{
    m4 r; memset(&r, 0, sizeof r);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
static m4 m4_add(m4 a, m4 b) { m4 r; for (int i = 0; i < 16; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
static m4 m4_scale(m4 a, float s) { m4 r; for (int i = 0; i < 16; i++) r.m[i] = a.m[i] * s; return r; }
static m4 m4_mul(m4 a, m4 b)
{
    m4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[4 * i + j] = ((a.m[4 * i + 0] * b.m[0 + j] + a.m[4 * i + 1] * b.m[4 + j])
                              + a.m[4 * i + 2] * b.m[8 + j]) + a.m[4 * i + 3] * b.m[12 + j];
    return r;
}
/* Vector4.Transform(Vector4(v, w), M).xyz : row vector times matrix */
static v3 m4_transform_v4(v3 v, float w, const m4 *M)
{
    const float *m = M->m;
    return V3(((v.x * m[0] + v.y * m[4]) + v.z * m[8]) + w * m[12],
              ((v.x * m[1] + v.y * m[5]) + v.z * m[9]) + w * m[13],
              ((v.x * m[2] + v.y * m[6]) + v.z * m[10]) + w * m[14]);
}
/* Vector4.Transform(Vector3 position, M).xyz : w = 1 is not multiplied, M4x is added */
static v3 m4_transform_pos(v3 v, const m4 *M)
{
    const float *m = M->m;
    return V3(((v.x * m[0] + v.y * m[4]) + v.z * m[8]) + m[12],
              ((v.x * m[1] + v.y * m[5]) + v.z * m[9]) + m[13],
              ((v.x * m[2] + v.y * m[6]) + v.z * m[10]) + m[14]);
}
/* Vector3.TransformNormal(n, M) */
static v3 m4_transform_normal(v3 v, const m4 *M)
{
    const float *m = M->m;
    return V3((v.x * m[0] + v.y * m[4]) + v.z * m[8],
              (v.x * m[1] + v.y * m[5]) + v.z * m[9],
              (v.x * m[2] + v.y * m[6]) + v.z * m[10]);
}

/* ------------------------------------------------------------------------------------------
 * raytracer/Numerics.fs
 * ---------------------------------------------------------------------------------------- */
/* Numerics.fs:9  Round = MathF.Round(num, digits), midpoint to even */
float rto_round(float x, int digits)
{
    static const float p10[7] = { 1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f };
    if (fabsf(x) < 1e8f) {
        float p = p10[digits];
        x = x * p;
        x = rintf(x);
        x = x / p;
    }
    return x;
}
/* Numerics.fs:12-16 NormalizedOrFail (the warning print is not restated) */
static v3 normalized_or_fail(v3 v)
{
    if (rto_round(v3_length(v), 5) == 1.0f) return v;
    return v3_normalize(v);
}
/* Numerics.fs:31 */
static inline float square(float b) { return b * b; }
/* Numerics.fs:33-37 */
static m4 rotation_of(const m4 *M)
{
    m4 r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[4 * i + j] = M->m[4 * i + j];
    return r;
}
/* Numerics.fs:45-49 */
static m4 change_of_base(v3 nt, v3 n, v3 nb)
{
    m4 r; memset(&r, 0, sizeof r);
    r.m[0] = nt.x; r.m[1] = nt.y; r.m[2] = nt.z;
    r.m[4] = n.x;  r.m[5] = n.y;  r.m[6] = n.z;
    r.m[8] = nb.x; r.m[9] = nb.y; r.m[10] = nb.z;
    return r;
}
/* Numerics.fs:51-55 */
static m4 skew_symmetric(v3 v)
{
    m4 r; memset(&r, 0, sizeof r);
    r.m[1] = -v.z; r.m[2] = v.y;
    r.m[4] = v.z;  r.m[6] = -v.x;
    r.m[8] = -v.y; r.m[9] = v.x;
    return r;
}
/* Numerics.fs:59,63-76  (cos of |omega| = sin(theta), as written; R is not orthonormal) */
static m4 rotation_between_unit_vectors(v3 a, v3 b)
{
    v3 omega = v3_cross(a, b);
    m4 ox = skew_symmetric(omega);
    m4 ox2 = m4_mul(ox, ox);
    float angle = sqrtf(v3_dot(omega, omega));
    float c = cosf(angle);
    return m4_add(m4_add(m4_identity(), ox), m4_scale(ox2, 1.0f / (1.0f + c)));
}

/* ------------------------------------------------------------------------------------------
 * Counter-based RNG (replaces Henzai.Core RandomSampling; see header)
 * ---------------------------------------------------------------------------------------- */
void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* 24 random bits -> [0,1), exact in f32 */
static inline float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

/* cos and sin of 2*pi*u for u in [0,1): octant reduction (exact) + cephes minimax polynomials,
 * every operation a single f32 rounding in the order written, so that the GPU restatement is
 * bit-identical.  Stands in for MathF.Cos/Sin inside the (missing) Henzai sampler. */
void rto_sincos_2pi(float u, float *s_out, float *c_out)
{
    float t = u * 8.0f;                 /* exact */
    int k = (int)t;                     /* octant 0..7 */
    int j = (k + 1) & ~1;               /* nearest even octant boundary */
    float f = t - (float)j;             /* exact, in [-1,1) */
    float a = f * 0.78539816339744830962f;
    float z = a * a;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * a + a;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch ((j >> 1) & 3) {
    case 0:  *s_out = sp;  *c_out = cp;  break;
    case 1:  *s_out = cp;  *c_out = -sp; break;
    case 2:  *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp;  break;
    }
}

/* RandomSampling.RandomInUnitHemisphere_Sync (call sites Surface.fs:72,99): uniform over the
 * y-up unit hemisphere, .Y = cos(theta) (Surface.fs:73). */
v3 rto_hemisphere(uint64_t seed, uint64_t ps, uint64_t node)
{
    uint32_t ctr[4] = { (uint32_t)ps, (uint32_t)(ps >> 32), (uint32_t)node, (uint32_t)(node >> 32) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    rto_philox4x32_10(ctr, key, w);
    float r1 = u01(w[0]);
    float r2 = u01(w[1]);
    float sin_theta = sqrtf(1.0f - r1 * r1);
    float s, c;
    rto_sincos_2pi(r2, &s, &c);
    return V3(sin_theta * c, r1, sin_theta * s);
}
/* RandomSampling.RandomInUnitSphere_Sync (call site Camera.fs:9): rejection sampling, at most
 * 32 tries keyed by (pixel, attempt), then the zero vector. */
v3 rto_in_unit_sphere(uint64_t seed, uint64_t pixel)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (uint32_t attempt = 0; attempt < 32; attempt++) {
        uint32_t ctr[4] = { (uint32_t)pixel, (uint32_t)(pixel >> 32), attempt, 0xFFFFFFFFu };
        uint32_t w[4];
        rto_philox4x32_10(ctr, key, w);
        v3 p = V3(2.0f * u01(w[0]) - 1.0f, 2.0f * u01(w[1]) - 1.0f, 2.0f * u01(w[2]) - 1.0f);
        if (v3_dot(p, p) < 1.0f) return p;
    }
    return V3(0.0f, 0.0f, 0.0f);
}
/* Henzai.Core Geometry.CreateCoordinateSystemAroundNormal (call sites Surface.fs:76,102) */
void rto_frame(v3 n, v3 *nt, v3 *nb)
{
    if (fabsf(n.x) > fabsf(n.y))
        *nt = v3_divs(V3(n.z, 0.0f, -n.x), sqrtf(n.x * n.x + n.z * n.z));
    else
        *nt = v3_divs(V3(0.0f, -n.z, n.y), sqrtf(n.y * n.y + n.z * n.z));
    *nb = v3_cross(n, *nt);
}

/* ------------------------------------------------------------------------------------------
 * raytracer/Geometry
 * ---------------------------------------------------------------------------------------- */
#define TMIN 0.001f   /* Core.fs:37 */
#define TMAX 500.0f   /* Core.fs:38 */

/* Core.fs:15-24 */
typedef struct { v3 o, d; } ray_t;
static ray_t make_ray(v3 o, v3 d) { ray_t r; r.o = o; r.d = normalized_or_fail(d); return r; }
/* Core.fs:59 */
static inline v3 point_for_ray(const ray_t *r, float t) { return v3_add(r->o, v3_scale(r->d, t)); }

typedef struct {
    int kind, geom, nsamples;
    uint64_t id;
    v3 albedo, emission;
    float fuzz, ior;
    /* sphere */
    v3 center; float radius;
    /* plane: System.Numerics.Plane(normal, d) + options; n = Normalize(plane.Normal) (Plane.fs:15) */
    v3 pn_raw; float pd; v3 pn;
    int has_center; v3 pcenter; float pw, ph;
    /* hoisted rectangle constants (identical values to the per-test recomputation) */
    m4 R; v3 kern; v3 newB;
    /* triangle (NOT in the reference: Geometry/AABB.fs is empty and there is no triangle type; this is the
       build's own Hitable, see triangle_intersect) */
    v3 tv0, te1, te2, tn;
} osurf;

/* Sphere.fs:18-27 */
static void sphere_intersections(const osurf *s, const ray_t *r, int *has, float *i1, float *i2)
{
    v3 ctr = v3_sub(r->o, s->center);
    float B = 2.0f * v3_dot(ctr, r->d);
    float C = v3_dot(ctr, ctr) - square(s->radius);
    float disc = B * B - 4.0f * C;
    if (disc < 0.0f) { *has = 0; *i1 = 0.0f; *i2 = 0.0f; }
    else if (rto_round(disc, 3) == 0.0f) { *has = 0; *i1 = -B / 2.0f; *i2 = -FLT_MAX; }
    else { *has = 1; *i1 = (-B + sqrtf(disc)) / 2.0f; *i2 = (-B - sqrtf(disc)) / 2.0f; }
}
/* Sphere.fs:29-36 */
static void sphere_intersect(const osurf *s, const ray_t *r, int *has, float *t)
{
    float i1, i2;
    sphere_intersections(s, r, has, &i1, &i2);
    if (i1 >= TMIN && i2 >= TMIN) *t = (i1 < i2) ? i1 : i2;
    else if (i1 < 0.0f) *t = i2;
    else *t = i1;
}
/* Plane.fs:19-30, the per-plane part: R, kern, newB */
static void plane_rect_constants(const osurf *s, m4 *R, v3 *kern, v3 *newB)
{
    *R = rotation_between_unit_vectors(s->pn, V3(0.0f, 0.0f, -1.0f));
    *kern = (s->pd > 0.0f) ? v3_scale(s->pn, -1.0f * s->pd) : v3_scale(s->pn, s->pd);
    v3 b = v3_sub(s->pcenter, *kern);
    *newB = v3_add(*kern, m4_transform_v4(b, 0.0f, R));
}
/* Plane.fs:19-34 */
static int plane_point_in_rectangle(const osurf *s, v3 p, unsigned flags)
{
    float wOff = s->pw / 2.0f, hOff = s->ph / 2.0f;
    m4 Rl; v3 kernl, newBl;
    const m4 *R = &s->R; v3 kern = s->kern, newB = s->newB;
    if (flags & RTO_FLAG_LITERAL_PLANE) {
        plane_rect_constants(s, &Rl, &kernl, &newBl);
        R = &Rl; kern = kernl; newB = newBl;
    }
    v3 v = v3_sub(p, kern);
    v3 newP = v3_add(kern, m4_transform_v4(v, 0.0f, R));
    return newP.x <= newB.x + wOff && newP.x >= newB.x - wOff &&
           newP.y <= newB.y + hOff && newP.y >= newB.y - hOff;
}
/* Plane.fs:36-40 ; Plane.DotNormal uses the stored (un-normalised) plane.Normal */
static void plane_intersect(const osurf *s, const ray_t *r, int *has, float *t)
{
    float num = -s->pd - v3_dot(s->pn_raw, r->o);
    float den = v3_dot(s->pn_raw, r->d);
    if (fabsf(den) < TMIN) { *has = 0; *t = 0.0f; }
    else { *has = 1; *t = num / den; }
}
/* Triangle as a Hitable (Geometry/Core.fs:26-43 contract): build-defined, no reference behaviour to
 * match.  Moeller-Trumbore in f32, every operation in the order written:
 *   p = D x e2; det = e1.p; |det| < 1e-12 -> no hit; inv = 1/det; tv = O - v0; u = (tv.p)*inv;
 *   q = tv x e1; v = (D.q)*inv; hit iff u >= 0, v >= 0, u + v <= 1; t = (e2.q)*inv.
 * Acceptable like a sphere: has && t > TMin (no TMax).  Normal = Normalize(e1 x e2), never flipped. */
/* cross and dot product with explicit fused multiply-adds in a fixed order: the triangle Hitable is this
   build's own (the reference has none) and is DEFINED this way; rt_bvh.h cross_fma / dot_fma mirror it */
static v3 tri_cross(v3 a, v3 b)
{
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static float tri_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static void triangle_intersect(const osurf *s, const ray_t *r, int *has, float *t)
{
    v3 p = tri_cross(r->d, s->te2);
    float det = tri_dot(s->te1, p);
    float inv = 1.0f / det;
    v3 tv = v3_sub(r->o, s->tv0);
    float u = tri_dot(tv, p) * inv;
    v3 q = tri_cross(tv, s->te1);
    float v = tri_dot(r->d, q) * inv;
    *t = tri_dot(s->te2, q) * inv;
    *has = !(fabsf(det) < 1e-12f) && u >= 0.0f && v >= 0.0f && (u + v) <= 1.0f;
}
/* Hitable.Intersect dispatch */
static void geom_intersect(const osurf *s, const ray_t *r, int *has, float *t)
{
    if (s->geom == RT_GEOM_SPHERE) sphere_intersect(s, r, has, t);
    else if (s->geom == RT_GEOM_TRIANGLE) triangle_intersect(s, r, has, t);
    else plane_intersect(s, r, has, t);
}
/* Hitable.IntersectionAcceptable: Sphere.fs:42-43, Plane.fs:45-49 */
static int geom_acceptable(const osurf *s, int has, float t, float dot_view, v3 p, unsigned flags)
{
    if (s->geom == RT_GEOM_PLANE) {
        int general = has && t > TMIN && t <= (TMAX / dot_view);
        if (s->has_center) return general && plane_point_in_rectangle(s, p, flags);
        return general;
    }
    return has && t > TMIN;
}
/* Hitable.NormalForSurfacePoint: Sphere.fs:37-38, Plane.fs:50-51 */
static v3 geom_normal(const osurf *s, v3 p)
{
    if (s->geom == RT_GEOM_SPHERE) return v3_normalize(v3_sub(p, s->center));
    if (s->geom == RT_GEOM_TRIANGLE) return s->tn;
    return s->pn;
}

/* ------------------------------------------------------------------------------------------
 * raytracer/Surface.fs
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const osurf *surf; int n;
    uint64_t seed;
    int max_trace_depth;
    v3 background;
    unsigned flags;
    int estimator;       /* RT_ESTIMATOR_*: 0 = the reference's branching renderScene, 1 = single path */
    uint64_t rays;
} octx;

/* Surface.fs:49-58 */
static int find_closest_intersection(octx *c, const ray_t *r, int *bMin, float *tMin)
{
    int vMin = -1;       /* the NoSurface(0, NotHitable) sentinel */
    *bMin = 0; *tMin = FLT_MAX;
    c->rays++;
    for (int i = 0; i < c->n; i++) {
        int b; float t;
        geom_intersect(&c->surf[i], r, &b, &t);
        if (geom_acceptable(&c->surf[i], b, t, 1.0f, point_for_ray(r, t), c->flags) && t < *tMin) {
            *bMin = b; *tMin = t; vMin = i;
        }
    }
    return vMin;
}

/* MathF.Pow(x, 5.0f), see header */
static inline float pow5(float x) { double d = (double)x; double d2 = d * d; return (float)(d2 * d2 * d); }

/* Surface.fs:119-121 */
static float schlick(float cos_i, float eta_i, float eta_t)
{
    float q = (eta_t - eta_i) / (eta_t + eta_i);
    float R0 = q * q;
    return R0 + (1.0f - R0) * pow5(1.0f - cos_i);
}
/* Surface.fs:89-90 and 122-123 */
static v3 reflect(v3 d, v3 n) { return v3_sub(d, v3_scale(n, 2.0f * v3_dot(d, n))); }

typedef struct { ray_t ray; v3 shading; } sample_t;

/* Surface.fs:66-83 + ComputeSample/MCComputeBRDF Surface.fs:28-29 */
static sample_t lambertian_sample(const octx *c, const osurf *s, const ray_t *in, float t,
                                  uint64_t ps, uint64_t child_node)
{
    v3 p = v3_add(in->o, v3_scale(in->d, t));
    v3 normal = geom_normal(s, p);
    v3 rn = rto_hemisphere(c->seed, ps, child_node);
    float cos_i = rn.y;
    v3 nt, nb;
    rto_frame(normal, &nt, &nb);
    m4 cob = change_of_base(nt, normal, nb);
    v3 out = v3_normalize(m4_transform_pos(rn, &cob));
    sample_t r;
    r.ray = make_ray(p, out);
    float pdf = 1.0f / (2.0f * 3.14159265358979323846f);      /* Surface.fs:64 */
    v3 brdf = v3_divs(s->albedo, 3.14159265358979323846f);    /* Surface.fs:65 */
    r.shading = v3_scale(brdf, cos_i / pdf);                   /* Surface.fs:28 */
    return r;
}
/* Surface.fs:93-109 */
static sample_t metal_sample(const octx *c, const osurf *s, const ray_t *in, float t,
                             uint64_t ps, uint64_t child_node)
{
    v3 p = v3_add(in->o, v3_scale(in->d, t));
    v3 normal = v3_normalize(geom_normal(s, p));
    v3 rn = rto_hemisphere(c->seed, ps, child_node);
    v3 nt, nb;
    rto_frame(normal, &nt, &nb);
    m4 cob = change_of_base(nt, normal, nb);
    v3 ns = m4_transform_pos(rn, &cob);
    float fuzz = fmaxf(fminf(1.0f, s->fuzz), 0.0f);           /* Surface.fs:88 */
    v3 mod = v3_normalize(v3_add(v3_scale(normal, 1.0f - fuzz), v3_scale(ns, fuzz)));
    v3 out = v3_normalize(reflect(in->d, mod));
    sample_t r;
    r.ray = make_ray(p, out);
    r.shading = v3_scale(s->albedo, 1.0f / 1.0f);              /* BRDF*(cos/PDF), cos = PDF = 1 */
    return r;
}
/* Surface.fs:124-153 + 168-181 ; returns the sample count */
static int dielectric_samples(const osurf *s, const ray_t *in, float t, sample_t out[2])
{
    v3 p = v3_add(in->o, v3_scale(in->d, t));
    v3 normal = v3_normalize(geom_normal(s, p));
    v3 refl = v3_normalize(reflect(in->d, normal));
    float eta_i, eta_t; v3 fn;
    if (v3_dot(in->d, normal) > 0.0f) { eta_i = s->ior; eta_t = 1.0f; fn = v3_neg(normal); }
    else { eta_i = 1.0f; eta_t = s->ior; fn = normal; }
    float cos_i = v3_dot(in->d, v3_neg(fn));
    float eta = eta_i / eta_t;
    float disc = 1.0f - square(eta) * (1.0f - square(cos_i));
    int refracted = 0; v3 refr = V3(0.0f, 0.0f, 0.0f);
    if (disc > 0.0f) {
        v3 q = v3_sub(v3_scale(v3_add(in->d, v3_scale(fn, cos_i)), eta), v3_scale(fn, sqrtf(disc)));
        refr = v3_normalize(q);
        refracted = 1;
    }
    float F = refracted ? schlick(cos_i, eta_i, eta_t) : 1.0f;
    v3 refl_sh = v3_scale(s->albedo, F);
    out[0].ray = make_ray(p, refl);
    if (rto_round(F, 3) == 1.0f) { out[0].shading = refl_sh; return 1; }
    v3 refr_sh = v3_scale(s->albedo, 1.0f - F);
    out[0].shading = v3_scale(refl_sh, 2.0f);
    out[1].ray = make_ray(p, refr);
    out[1].shading = v3_scale(refr_sh, 2.0f);
    return 2;
}

/* RandomSampling.RandomFloat_Sync (call site Surface.fs:157): uniform in [0,1); the third word of the Philox
 * block keyed by (pixel-sample, child node) -- the block whose first two words a hemisphere draw would use. */
float rto_random_float(uint64_t seed, uint64_t ps, uint64_t node)
{
    uint32_t ctr[4] = { (uint32_t)ps, (uint32_t)(ps >> 32), (uint32_t)node, (uint32_t)(node >> 32) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t w[4];
    rto_philox4x32_10(ctr, key, w);
    return u01(w[2]);
}
/* Dielectric.Scatter (Surface.fs:155-166) through the default GenerateSamples (Surface.fs:37-41) with
 * SampleCount 1: the stochastic pick the reference defines but never reaches (its GenerateSamples override
 * shadows it).  shading = ComputeSample = BRDF * (cos / PDF) = Albedo * (1 / 1)  (Surface.fs:28,35-36,162,166). */
static sample_t dielectric_scatter(const octx *c, const osurf *s, const ray_t *in, float t,
                                   uint64_t ps, uint64_t child_node)
{
    v3 p = v3_add(in->o, v3_scale(in->d, t));
    v3 normal = v3_normalize(geom_normal(s, p));
    v3 refl = v3_normalize(reflect(in->d, normal));
    float eta_i, eta_t; v3 fn;
    if (v3_dot(in->d, normal) > 0.0f) { eta_i = s->ior; eta_t = 1.0f; fn = v3_neg(normal); }
    else { eta_i = 1.0f; eta_t = s->ior; fn = normal; }
    float cos_i = v3_dot(in->d, v3_neg(fn));
    float eta = eta_i / eta_t;
    float disc = 1.0f - square(eta) * (1.0f - square(cos_i));
    int refracted = 0; v3 refr = V3(0.0f, 0.0f, 0.0f);
    if (disc > 0.0f) {
        v3 q = v3_sub(v3_scale(v3_add(in->d, v3_scale(fn, cos_i)), eta), v3_scale(fn, sqrtf(disc)));
        refr = v3_normalize(q);
        refracted = 1;
    }
    float reflect_prob = refracted ? schlick(cos_i, eta_i, eta_t) : 1.0f;
    float random_float = rto_random_float(c->seed, ps, child_node);
    sample_t r;
    if (random_float <= reflect_prob) r.ray = make_ray(p, refl);
    else r.ray = make_ray(p, refr);          /* "refraction has to have been successful" (Surface.fs:163) */
    r.shading = v3_scale(s->albedo, 1.0f / 1.0f);
    return r;
}

/* Surface.GenerateSamples: Surface.fs:37-41 (default), :47 (NoSurface), :168-181 (Dielectric).
 * Child i of tree node `node` is node 4*node+i+1; its hemisphere draw is keyed by that id. */
static int generate_samples(const octx *c, const osurf *s, const ray_t *in, float t,
                            uint64_t ps, uint64_t node, sample_t out[4])
{
    if (c->estimator == RT_ESTIMATOR_SINGLE_PATH) {
        /* one child per hit, through the default GenerateSamples with SampleCount 1 (notes.txt:21-54 with one
           ray per level); Dielectric through its stochastic Scatter */
        switch (s->kind) {
        case RT_KIND_LAMBERTIAN: out[0] = lambertian_sample(c, s, in, t, ps, 4 * node + 1); return 1;
        case RT_KIND_METAL:      out[0] = metal_sample(c, s, in, t, ps, 4 * node + 1); return 1;
        case RT_KIND_DIELECTRIC: out[0] = dielectric_scatter(c, s, in, t, ps, 4 * node + 1); return 1;
        default:                 return 0;
        }
    }
    switch (s->kind) {
    case RT_KIND_LAMBERTIAN:
        for (int i = 0; i < s->nsamples; i++) out[i] = lambertian_sample(c, s, in, t, ps, 4 * node + i + 1);
        return s->nsamples;
    case RT_KIND_METAL:
        for (int i = 0; i < s->nsamples; i++) out[i] = metal_sample(c, s, in, t, ps, 4 * node + i + 1);
        return s->nsamples;
    case RT_KIND_DIELECTRIC:
        return dielectric_samples(s, in, t, out);
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * raytracer/Scene.fs
 * ---------------------------------------------------------------------------------------- */
static inline int v3_is_zero(v3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

/* Scene.fs:51-72 ; `prev` is previousTraceDepth */
static v3 ray_trace(octx *c, int prev, const ray_t *r, uint64_t ps, uint64_t node)
{
    if (prev > c->max_trace_depth) return c->background;
    int b; float t;
    int idx = find_closest_intersection(c, r, &b, &t);
    if (idx >= 0 && geom_acceptable(&c->surf[idx], b, t, 1.0f, point_for_ray(r, t), c->flags)) {
        const osurf *s = &c->surf[idx];
        /* children would be called with prev+1; when that exceeds maxTraceDepth they all
           return backgroundColor without intersecting (Scene.fs:52-54) */
        int last = (prev + 1 > c->max_trace_depth);
        if (last && v3_is_zero(c->background) && !(c->flags & RTO_FLAG_LITERAL_LEAVES))
            return s->emission;      /* emitted + (sum shading*0)/n == emitted */
        sample_t smp[4];
        int n = generate_samples(c, s, r, t, ps, node, smp);
        if (n == 0) return s->emission;
        v3 total = V3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < n; i++)
            total = v3_add(total, v3_mul(smp[i].shading, ray_trace(c, prev + 1, &smp[i].ray, ps, 4 * node + i + 1)));
        return v3_add(s->emission, v3_divs(total, (float)n));
    }
    return c->background;
}

static void convert_surfaces(const rt_surface *in, int64_t n, int lamb_default, osurf *out)
{
    for (int64_t i = 0; i < n; i++) {
        const rt_surface *a = &in[i]; osurf *s = &out[i];
        memset(s, 0, sizeof *s);
        s->kind = a->kind; s->geom = a->geom; s->id = a->id;
        s->albedo = V3(a->albedo[0], a->albedo[1], a->albedo[2]);
        s->emission = V3(a->emission[0], a->emission[1], a->emission[2]);
        s->fuzz = a->fuzz; s->ior = a->ior;
        switch (a->kind) {
        case RT_KIND_LAMBERTIAN: s->nsamples = a->sample_count > 0 ? a->sample_count : lamb_default; break;
        case RT_KIND_METAL:      s->nsamples = a->sample_count > 0 ? a->sample_count : 1; break;
        case RT_KIND_DIELECTRIC: s->nsamples = 2; break;
        default:                 s->nsamples = 0; break;
        }
        if (a->geom == RT_GEOM_SPHERE) {
            s->center = V3(a->g[0], a->g[1], a->g[2]); s->radius = a->g[3];
        } else if (a->geom == RT_GEOM_PLANE) {
            s->pn_raw = V3(a->g[0], a->g[1], a->g[2]); s->pd = a->g[3];
            s->pn = v3_normalize(s->pn_raw);
            s->has_center = a->g[4] != 0.0f;
            s->pcenter = V3(a->g[5], a->g[6], a->g[7]); s->pw = a->g[8]; s->ph = a->g[9];
            if (s->has_center) plane_rect_constants(s, &s->R, &s->kern, &s->newB);
        } else if (a->geom == RT_GEOM_TRIANGLE) {
            v3 v1 = V3(a->g[3], a->g[4], a->g[5]), v2 = V3(a->g[6], a->g[7], a->g[8]);
            s->tv0 = V3(a->g[0], a->g[1], a->g[2]);
            s->te1 = v3_sub(v1, s->tv0); s->te2 = v3_sub(v2, s->tv0);
            s->tn = v3_normalize(v3_cross(s->te1, s->te2));
        }
    }
}

static int check_args(const rt_surface *surf, int64_t n, const rt_camera *cam, const rt_params *p)
{
    if (!surf || n < 0 || !cam || !p) return RT_E_INVALID;
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0) return RT_E_INVALID;
    if (p->max_trace_depth < 0 || p->max_trace_depth > 28) return RT_E_INVALID;
    if (p->lambertian_samples < 0 || p->lambertian_samples > 4) return RT_E_INVALID;
    if (p->x_step <= 0 || p->x_begin < 0 || p->x_end > p->width || p->x_begin > p->x_end) return RT_E_INVALID;
    if (p->estimator != RT_ESTIMATOR_BRANCHING && p->estimator != RT_ESTIMATOR_SINGLE_PATH) return RT_E_INVALID;
    for (int64_t i = 0; i < n; i++) {
        if (surf[i].kind < 0 || surf[i].kind > 3 || surf[i].geom < 0 || surf[i].geom > 2) return RT_E_INVALID;
        if (surf[i].sample_count < 0 || surf[i].sample_count > 4) return RT_E_INVALID;
    }
    return RT_OK;
}

/* Optional per-pixel ray-count map (analysis only): element k*height+py receives the number of
 * findClosestIntersection calls of that pixel.  NULL (default) disables it. */
static uint32_t *g_raymap = 0;
void rto_set_raymap(uint32_t *map) { g_raymap = map; }

/* Scene.renderScene (Scene.fs:117-121) with renderPass (Scene.fs:97-114), rayTraceBase
 * (Scene.fs:75-95) and the Camera functions (Camera.fs:8-39) inlined per pixel.
 * Same arguments as rt_render plus oracle flags and a thread count (0 = all cores; the
 * reference itself is serial, pixels are independent so threads do not change results). */
int rto_render(const rt_surface *surfaces, int64_t n, const rt_camera *cam, const rt_params *p,
               unsigned flags, int nthreads,
               float *rgba, float *depth, uint64_t *object_id, float *max_depth, uint64_t *ray_count)
{
    int rc = check_args(surfaces, n, cam, p);
    if (rc) return rc;
    if (!rgba || !depth) return RT_E_INVALID;
    osurf *surf = (osurf *)malloc(sizeof(osurf) * (size_t)(n > 0 ? n : 1));
    if (!surf) return RT_E_INVALID;
    convert_surfaces(surfaces, n, p->lambertian_samples > 0 ? p->lambertian_samples : 4, surf);

    const int W = p->width, H = p->height, spp = p->spp;
    const int ncols = (p->x_end - p->x_begin + p->x_step - 1) / p->x_step;
    m4 cameraWS; memcpy(cameraWS.m, cam->cam_to_world, sizeof cameraWS.m);
    const m4 rot = rotation_of(&cameraWS);                                /* Scene.fs:100 */
    const v3 origin = V3(cameraWS.m[12], cameraWS.m[13], cameraWS.m[14]); /* .Translation  */
    const v3 look_n = v3_normalize(V3(cam->look_at[0], cam->look_at[1], cam->look_at[2]));
    const float fw = (float)W, fh = (float)H;
    const float aspect = fw / fh;                                         /* Camera.fs:27  */
    const float tan_half = tanf(cam->fov / 2.0f);                         /* Camera.fs:32  */
    const v3 background = V3(p->background[0], p->background[1], p->background[2]);

    uint64_t total_rays = 0; float maxd = 0.0f;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total_rays) reduction(max : maxd)
    for (int k = 0; k < ncols; k++) {
        const int px = p->x_begin + k * p->x_step;
        octx c; c.surf = surf; c.n = (int)n; c.seed = p->seed; c.max_trace_depth = p->max_trace_depth;
        c.background = background; c.flags = flags; c.rays = 0; c.estimator = p->estimator;
        for (int py = 0; py < H; py++) {
            const uint64_t pixel = (uint64_t)px * (uint64_t)H + (uint64_t)py;
            /* Camera.fs:8-17 pertrube, :20-22 PixelToNDC */
            v3 rv = p->jitter ? rto_in_unit_sphere(p->seed, pixel) : V3(0.0f, 0.0f, 0.0f);
            float xoff = rv.x / fw, yoff = rv.y / fh;
            float pxp = ((float)px + 0.5f) + xoff, pyp = ((float)py + 0.5f) + yoff;
            float xn = pxp / fw, yn = pyp / fh;
            /* Camera.fs:25 NDCToScreen, :31-32 ScreenToCamera, :38-39 RayDirection */
            float xs = 2.0f * xn - 1.0f, ys = 1.0f - 2.0f * yn;
            float xc = xs * aspect * tan_half, yc = ys * tan_half;
            v3 dirCS = v3_normalize(V3(xc, yc, -1.0f));
            v3 dirWS = v3_normalize(m4_transform_normal(dirCS, &rot));     /* Scene.fs:101 */
            ray_t ray = make_ray(origin, dirWS);                            /* Scene.fs:102 */

            const size_t o = (size_t)k * (size_t)H + (size_t)py;
            const uint64_t rays_before = c.rays;
            depth[o] = FLT_MAX;                                             /* Scene.fs:29 */
            if (object_id) object_id[o] = 0;
            v3 sum = V3(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < spp; s++) {
                const uint64_t ps = pixel * (uint64_t)spp + (uint64_t)s;
                /* rayTraceBase, Scene.fs:75-95 ; the root of the ray tree is node 0 */
                v3 col;
                float dot_view = v3_dot(look_n, ray.d);                     /* Scene.fs:76 */
                int b; float t;
                int idx = find_closest_intersection(&c, &ray, &b, &t);
                if (idx >= 0 && geom_acceptable(&surf[idx], b, t, dot_view, point_for_ray(&ray, t), flags)) {
                    if (s == 0) {                                           /* Scene.fs:80 */
                        depth[o] = t;
                        if (t > maxd) maxd = t;
                        if (object_id) object_id[o] = surf[idx].id;
                    }
                    sample_t smp[4];
                    int ns = generate_samples(&c, &surf[idx], &ray, t, ps, 0, smp);
                    if (ns == 0) col = surf[idx].emission;
                    else {
                        v3 total = V3(0.0f, 0.0f, 0.0f);
                        for (int i = 0; i < ns; i++)
                            total = v3_add(total, v3_mul(smp[i].shading, ray_trace(&c, 0, &smp[i].ray, ps, (uint64_t)i + 1)));
                        col = v3_add(surf[idx].emission, v3_divs(total, (float)ns));
                    }
                } else col = background;
                /* Scene.fs:109 Array.reduce (+) in sample order */
                sum = (s == 0) ? col : v3_add(sum, col);
            }
            v3 avg = v3_divs(sum, (float)spp);
            rgba[4 * o + 0] = sqrtf(avg.x);                                 /* Scene.fs:114 */
            rgba[4 * o + 1] = sqrtf(avg.y);
            rgba[4 * o + 2] = sqrtf(avg.z);
            rgba[4 * o + 3] = 1.0f;
            if (g_raymap) g_raymap[o] = (uint32_t)(c.rays - rays_before);
        }
        total_rays += c.rays;
    }
    free(surf);
    if (max_depth) *max_depth = maxd;
    if (ray_count) *ray_count = total_rays;
    return RT_OK;
}

/* ------------------------------------------------------------------------------------------
 * Camera.fs:41-45 — Matrix4x4.CreateLookAt and Matrix4x4.Invert restated (the oracle's own
 * copy; tests use it to check the host mirror's camera).
 * ---------------------------------------------------------------------------------------- */
void rto_create_look_at(const float pos[3], const float target[3], const float up[3], float out[16])
{
    v3 p = V3(pos[0], pos[1], pos[2]);
    v3 z = v3_normalize(v3_sub(p, V3(target[0], target[1], target[2])));
    v3 x = v3_normalize(v3_cross(V3(up[0], up[1], up[2]), z));
    v3 y = v3_cross(z, x);
    float r[16] = { x.x, y.x, z.x, 0.0f,
                    x.y, y.y, z.y, 0.0f,
                    x.z, y.z, z.z, 0.0f,
                    -v3_dot(x, p), -v3_dot(y, p), -v3_dot(z, p), 1.0f };
    memcpy(out, r, sizeof r);
}
/* cofactor expansion along the first column blocks, as System.Numerics does */
int rto_invert(const float M[16], float out[16])
{
    float a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7];
    float i = M[8], j = M[9], k = M[10], l = M[11], m = M[12], n = M[13], o = M[14], p = M[15];
    float kp_lo = k * p - l * o, jp_ln = j * p - l * n, jo_kn = j * o - k * n;
    float ip_lm = i * p - l * m, io_km = i * o - k * m, in_jm = i * n - j * m;
    float a11 = +(f * kp_lo - g * jp_ln + h * jo_kn);
    float a12 = -(e * kp_lo - g * ip_lm + h * io_km);
    float a13 = +(e * jp_ln - f * ip_lm + h * in_jm);
    float a14 = -(e * jo_kn - f * io_km + g * in_jm);
    float det = a * a11 + b * a12 + c * a13 + d * a14;
    if (fabsf(det) < 1.401298464e-45f) return 0;
    float inv = 1.0f / det;
    out[0] = a11 * inv; out[4] = a12 * inv; out[8] = a13 * inv; out[12] = a14 * inv;
    out[1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv;
    out[5] = +(a * kp_lo - c * ip_lm + d * io_km) * inv;
    out[9] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv;
    out[13] = +(a * jo_kn - b * io_km + c * in_jm) * inv;
    float gp_ho = g * p - h * o, fp_hn = f * p - h * n, fo_gn = f * o - g * n;
    float ep_hm = e * p - h * m, eo_gm = e * o - g * m, en_fm = e * n - f * m;
    out[2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv;
    out[6] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv;
    out[10] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv;
    out[14] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv;
    float gl_hk = g * l - h * k, fl_hj = f * l - h * j, fk_gj = f * k - g * j;
    float el_hi = e * l - h * i, ek_gi = e * k - g * i, ej_fi = e * j - f * i;
    out[3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv;
    out[7] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv;
    out[11] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv;
    out[15] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Unit-level entry points for the known-answer tests (tests/test_oracle_kat.py)
 * ---------------------------------------------------------------------------------------- */
static void one_surface(const rt_surface *a, osurf *s) { convert_surfaces(a, 1, 4, s); }

/* Hitable.Intersect + IntersectionAcceptable for one surface */
void rto_intersect(const rt_surface *a, const float o[3], const float d[3], float dot_view,
                   int *has, float *t, int *acceptable)
{
    osurf s; one_surface(a, &s);
    ray_t r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    geom_intersect(&s, &r, has, t);
    *acceptable = geom_acceptable(&s, *has, *t, dot_view, point_for_ray(&r, *t), 0);
}
/* findClosestIntersection (Surface.fs:49-58) over a whole surface array, for a batch of rays */
void rto_find_closest(const rt_surface *surfaces, int64_t n, int64_t nrays, const float *o, const float *d,
                      float *t_out, int32_t *idx_out)
{
    osurf *surf = (osurf *)malloc(sizeof(osurf) * (size_t)(n > 0 ? n : 1));
    convert_surfaces(surfaces, n, 4, surf);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < nrays; i++) {
        octx c; memset(&c, 0, sizeof c); c.surf = surf; c.n = (int)n;
        ray_t r; r.o = V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.d = V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        int b; float t;
        idx_out[i] = find_closest_intersection(&c, &r, &b, &t);
        t_out[i] = t;
    }
    free(surf);
}
/* Surface.GenerateSamples for one surface hit at parameter t */
int rto_generate_samples(const rt_surface *a, const float o[3], const float d[3], float t,
                         uint64_t seed, uint64_t ps, uint64_t node, float out_dirs[12], float out_shading[12])
{
    osurf s; one_surface(a, &s);
    octx c; memset(&c, 0, sizeof c); c.seed = seed;
    ray_t r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    sample_t smp[4];
    int n = generate_samples(&c, &s, &r, t, ps, node, smp);
    for (int i = 0; i < n; i++) {
        out_dirs[3 * i] = smp[i].ray.d.x; out_dirs[3 * i + 1] = smp[i].ray.d.y; out_dirs[3 * i + 2] = smp[i].ray.d.z;
        out_shading[3 * i] = smp[i].shading.x; out_shading[3 * i + 1] = smp[i].shading.y; out_shading[3 * i + 2] = smp[i].shading.z;
    }
    return n;
}
/* the same with the single-path estimator (one child; Dielectric through its stochastic Scatter) */
int rto_generate_samples_single_path(const rt_surface *a, const float o[3], const float d[3], float t,
                                     uint64_t seed, uint64_t ps, uint64_t node, float out_dir[3], float out_shading[3])
{
    osurf s; one_surface(a, &s);
    octx c; memset(&c, 0, sizeof c); c.seed = seed; c.estimator = RT_ESTIMATOR_SINGLE_PATH;
    ray_t r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    sample_t smp[4];
    int n = generate_samples(&c, &s, &r, t, ps, node, smp);
    if (n > 0) {
        out_dir[0] = smp[0].ray.d.x; out_dir[1] = smp[0].ray.d.y; out_dir[2] = smp[0].ray.d.z;
        out_shading[0] = smp[0].shading.x; out_shading[1] = smp[0].shading.y; out_shading[2] = smp[0].shading.z;
    }
    return n;
}
void rto_hemisphere_sample(uint64_t seed, uint64_t ps, uint64_t node, float out[3])
{
    v3 h = rto_hemisphere(seed, ps, node); out[0] = h.x; out[1] = h.y; out[2] = h.z;
}
void rto_unit_sphere_sample(uint64_t seed, uint64_t pixel, float out[3])
{
    v3 h = rto_in_unit_sphere(seed, pixel); out[0] = h.x; out[1] = h.y; out[2] = h.z;
}
void rto_frame_around(const float n[3], float nt[3], float nb[3])
{
    v3 t, b; rto_frame(V3(n[0], n[1], n[2]), &t, &b);
    nt[0] = t.x; nt[1] = t.y; nt[2] = t.z; nb[0] = b.x; nb[1] = b.y; nb[2] = b.z;
}
void rto_plane_rotation(const float n[3], float out[16])
{
    m4 R = rotation_between_unit_vectors(V3(n[0], n[1], n[2]), V3(0.0f, 0.0f, -1.0f));
    memcpy(out, R.m, sizeof R.m);
}
float rto_pow5(float x) { return pow5(x); }
