// rt_device.h — device-side numerics and scene layout of the MI355X render loop.
//
// Every arithmetic step is a single IEEE f32 operation in the order System.Numerics performs it
// (DESIGN.md "Numerics contract"); the translation unit is compiled with -ffp-contract=off and
// hipcc's default correctly rounded f32 divide/sqrt, so a ray computed here has bit-identical
// t / hit point / child direction to the CPU restatement in oracle/ (which is NOT included or
// linked: this file is an independent implementation of the same reference lines).
//
// Reference lines are relative to /root/reference/raytracer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr float kTMin = 0.001f;   // Geometry/Core.fs:37
constexpr float kTMax = 500.0f;   // Geometry/Core.fs:38
constexpr float kFltMax = 3.402823466e+38f;

// geometry types of the flattened scene (bounded and unbounded planes are split so the
// closest-hit loop branches on a wave-uniform scalar)
constexpr int kGeomSphere = 0, kGeomPlaneInf = 1, kGeomPlaneRect = 2, kGeomTriangle = 3;
// surface kinds == RT_KIND_*
constexpr int kKindNone = 0, kKindLambert = 1, kKindMetal = 2, kKindDielectric = 3;

// One surface's geometry, read with SCALAR loads (the loop index is wave-uniform).  80 bytes.
//   sphere : a = center.xyz, radius^2                                  (Sphere.fs:18-27)
//   plane  : a = plane.Normal (as stored).xyz, plane.D                  (Plane.fs:36-40)
//   rect   : b = first column of R (R11,R21,R31),  x upper bound  (newB.x + w/2)
//            c = second column of R (R12,R22,R32), x lower bound  (newB.x - w/2)
//            d = kern.xyz,                          y upper bound  (newB.y + h/2)
//            ymin =                                 y lower bound  (newB.y - h/2)   (Plane.fs:19-34)
struct alignas(16) GeomRec {
    float a[4], b[4], c[4], d[4];
    float ymin;
    int32_t type;
    int32_t index;        // position in the caller's surface array (what the closest hit reports)
    int32_t pad;
};
static_assert(sizeof(GeomRec) == 80, "GeomRec layout");

// One surface's shading data, gathered per lane by hit index.  64 bytes = four 16-byte loads.
struct alignas(16) ShadeRec {
    float albedo[3];      // Material.Albedo
    float param;          // Metal: clamp(fuzz,0,1) (Surface.fs:88); Dielectric: refractive index
    float emission[3];    // Material.Emmitance
    uint32_t info;        // kind | nsamples << 2 | geom << 5 | (Lambertian with the default SampleCount) << 7
    float nc[3];          // sphere: center; plane: Normalize(plane.Normal) (Plane.fs:15)
    uint32_t pad0;
    uint64_t id;          // Surface.ID
    uint64_t pad1;
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec layout");

struct f3 { float x, y, z; };

#define RT_DEV __device__ __forceinline__

RT_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
// Vector3.Dot: (x*x' + y*y') + z*z'
RT_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// the value of the first active lane, as a wave-uniform (scalar) value
RT_DEV unsigned long long uniform64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// The XCD (0..7) this wave runs on.  HIP promises nothing about workgroup placement; the hardware register says it.
RT_DEV uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

// Lanes set in a ballot mask, as a 32-bit SCALAR value.  (__popcll of a ballot is a 64-bit value for the compiler, which
// then compares two such counts with a 64-bit VECTOR compare — there is no scalar one — and from there keeps everything
// computed from the comparison, the wave-uniform traversal statistics included, in vector registers.)
RT_DEV int popc_mask(unsigned long long m)
{
    if (__builtin_constant_p(m)) return __builtin_popcountll(m);
    m = uniform64(m);       // (a no-op for a mask in scalar registers; the compiler may hold a wave-uniform mask in vector ones)
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}

// ---- correctly rounded f32 sqrt and divide, normal-range operands --------------------------------
// hipcc's IEEE lowerings of sqrtf and '/' are v_sqrt_f32 / v_rcp_f32 plus an FMA correction that makes
// the result correctly rounded, wrapped in scaling code for subnormal operands and results (x < 2^-96,
// quotients near the subnormal range).  Every operand on the ray path is a normal number (lengths of
// unit-scale vectors, discriminants >= 5e-4, |denominators| >= 1e-3 whenever the quotient is used), so
// the same correction is applied here without the scaling: 9 instead of 15 instructions per sqrt, 8
// instead of 11 per divide, bit-identical results on that range (tests compare t bit for bit).
RT_DEV float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);                       // within 1 ulp
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float r = (r_dn <= 0.0f) ? s_dn : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}
RT_DEV float div_rn(float n, float d)
{
    float r = __builtin_amdgcn_rcpf(d);                               // within 1 ulp
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return q;
}

// Vector3 / float : one reciprocal, three multiplies
RT_DEV f3 divs(f3 a, float s) { float inv = div_rn(1.0f, s); return F3(a.x * inv, a.y * inv, a.z * inv); }
RT_DEV float length(f3 a) { return sqrt_rn(dot(a, a)); }
RT_DEV f3 normalize(f3 a) { return divs(a, length(a)); }

// Numerics.fs:9  MathF.Round(x, digits) with p = 10^digits, midpoint to even
RT_DEV float round_digits(float x, float p)
{
    if (fabsf(x) < 1e8f) { x = x * p; x = rintf(x); x = x / p; }
    return x;
}
// Round(x, digits) == k / p for an integer k (as a float): since p = 10^digits is exact and k/p is
// correctly rounded, the quotient equals target/p only when rint(x*p) == target (neighbouring
// integers differ by 1/p >> ulp).  Saves the division of round_digits; |x| < 1e8 as in MathF.Round.
RT_DEV bool rounds_to(float x, float p, float target) { return fabsf(x) < 1e8f && rintf(x * p) == target; }

// The Ray constructor (Geometry/Core.fs:21) renormalises its direction only when
// Round(|d|,5) <> 1 (NormalizedOrFail, Numerics.fs:12-16).  Every direction built here is the
// output of Vector3.Normalize, whose f32 length is within 1e-6 of 1 (three roundings of
// ~6e-8 each), far inside the 5e-6 window of Round(.,5) = 1: the test can never fire, so the
// direction is returned as is.  (tests/test_gpu_parity.py covers this through bit-exact t.)
RT_DEV f3 make_ray_dir(f3 normalized) { return normalized; }

// ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter =
//      (pixel-sample index, ray-tree node index); replaces Henzai.Core RandomSampling.
RT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    // opaque copy of the key: keeps the ten-round key schedule (18 scalar adds) next to its use instead of
    // in 18 (or, once SLP-paired, 36) loop-invariant SGPRs that spill in the closest-hit loop
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
RT_DEV float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

// sin/cos of 2*pi*u, u in [0,1): exact octant reduction + degree-7/8 minimax polynomials on
// [-pi/4, pi/4]; each line is one f32 rounding so CPU and GPU agree bit for bit.
RT_DEV void sincos_2pi(float u, float& s_out, float& c_out)
{
    float t = u * 8.0f;
    int k = (int)t;
    int j = (k + 1) & ~1;
    float f = t - (float)j;
    float a = f * 0.78539816339744830962f;
    float z = a * a;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * a + a;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int q = (j >> 1) & 3;
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    s_out = (q & 2) ? -s : s;
    c_out = (q == 1 || q == 2) ? -c : c;
}

// RandomSampling.RandomInUnitHemisphere (call sites Surface.fs:72,99): uniform on the y-up
// hemisphere, .Y = cos(theta) (Surface.fs:73)
// `rf` receives the third word of the same Philox block as a float in [0,1): RandomSampling.RandomFloat
// (call site Surface.fs:157), used by the single-path estimator's Dielectric.Scatter.
RT_DEV f3 hemisphere_sample(uint32_t seed_lo, uint32_t seed_hi, uint64_t ps, uint64_t node, float& rf)
{
    uint32_t w0, w1, w2, w3;
    philox4x32_10((uint32_t)ps, (uint32_t)(ps >> 32), (uint32_t)node, (uint32_t)(node >> 32), seed_lo, seed_hi,
                  w0, w1, w2, w3);
    float r1 = u01(w0), r2 = u01(w1);
    rf = u01(w2);
    float sin_theta = sqrt_rn(1.0f - r1 * r1);
    float s, c;
    sincos_2pi(r2, s, c);
    return F3(sin_theta * c, r1, sin_theta * s);
}
// RandomSampling.RandomInUnitSphere (call site Camera.fs:9): rejection sampling keyed by
// (pixel, attempt); at most 32 attempts
RT_DEV f3 unit_sphere_sample(uint32_t seed_lo, uint32_t seed_hi, uint64_t pixel)
{
    for (uint32_t attempt = 0; attempt < 32; attempt++) {
        uint32_t w0, w1, w2, w3;
        philox4x32_10((uint32_t)pixel, (uint32_t)(pixel >> 32), attempt, 0xFFFFFFFFu, seed_lo, seed_hi, w0, w1, w2, w3);
        f3 p = F3(2.0f * u01(w0) - 1.0f, 2.0f * u01(w1) - 1.0f, 2.0f * u01(w2) - 1.0f);
        if (dot(p, p) < 1.0f) return p;
    }
    return F3(0.0f, 0.0f, 0.0f);
}
// Geometry.CreateCoordinateSystemAroundNormal (call sites Surface.fs:76,102)
RT_DEV void frame_around(f3 n, f3& nt, f3& nb)
{
    bool xbig = fabsf(n.x) > fabsf(n.y);
    f3 a = xbig ? F3(n.z, 0.0f, -n.x) : F3(0.0f, -n.z, n.y);
    float l2 = xbig ? (n.x * n.x + n.z * n.z) : (n.y * n.y + n.z * n.z);
    nt = divs(a, sqrt_rn(l2));
    nb = cross(n, nt);
}

#ifndef RT_SPHERE_EARLY_OUT
#define RT_SPHERE_EARLY_OUT 1
#endif
constexpr bool kSphereEarlyOut = RT_SPHERE_EARLY_OUT != 0;

// ---- closest hit --------------------------------------------------------------------------
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// One GeomRec into 20 SGPRs, asynchronously (the compiler's own scalar loads are placed right before
// their first use, which exposes the scalar-cache latency once per surface; issued by hand, the load of
// surface i+1 overlaps the arithmetic of surface i).  geom_wait() must precede the first use.
RT_DEV void geom_load(const GeomRec* rec, v16f& lo, v4f& hi)
{
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40" : "=&s"(lo), "=&s"(hi) : "s"(rec) : "memory");
}
RT_DEV void geom_wait(v16f& lo, v4f& hi) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi)); }

// findClosestIntersection (Surface.fs:49-58): linear scan in array order, strict '<' so the first
// surface wins ties; each candidate passes Hitable.Intersect + IntersectionAcceptable(.., 1.0, P).
// The loop index is wave-uniform: each surface's record sits in SGPRs and costs no vector register.
// Conditions are combined with '&' so that no short-circuit branch is generated.
// PREFETCH: issue the load of surface i+1 before surface i is evaluated (two records, 40 scalar registers, in flight).
// It paid in round 1 (19 waves per CU, the loop-carried state of the render loop in vector registers); with 22 waves per
// CU hiding the scalar-cache latency and the bookkeeping in scalar registers it costs more than it hides: one record at
// a time is 8.5 % faster on the 4K sphere scene and 5 % on the mixed scene (measured, round 2).  Kept for reference.
template <bool PREFETCH = false>
RT_DEV void closest_hit(const GeomRec* __restrict__ geom, int nsurf, f3 O, f3 D, int& idx_out, float& t_out)
{
    int idx = -1;
    float tmin = kFltMax;
    v16f r; v4f rh;
    if (PREFETCH && nsurf > 0) { geom_load(geom, r, rh); geom_wait(r, rh); }
    for (int i = 0; i < nsurf; i++) {
        v16f nr; v4f nrh;
        if (PREFETCH) geom_load(geom + ((i + 1 < nsurf) ? i + 1 : i), nr, nrh);
        else { geom_load(geom + i, r, rh); geom_wait(r, rh); }
        const int type = __float_as_int(rh[1]);
        float t; bool ok;
        if (type == kGeomSphere) {
            // Sphere.fs:18-36,42-43, with B = 2b and disc = 4d factored out: b = dot(ctr, D),
            // d = b*b - C.  Scaling by 2 and 4 is exact in binary floating point and commutes with
            // rounding, so (-B +- sqrt(disc))/2 == -b +- sqrt(d) bit for bit, disc < 0 <=> d < 0, and
            // Round(disc,3) = 0 <=> rint(4*fl(1000 d)) = 0 <=> |fl(1000 d)| <= 0.125.
            const f3 ctr = O - F3(r[0], r[1], r[2]);
            const float b = dot(ctr, D);
            const float C = dot(ctr, ctr) - r[3];
            const float d = b * b - C;
            const bool has = d * 1e3f > 0.125f;          // == !(d < 0) & !(|1000 d| <= 0.125) for every d that can still give a hit
            // A sphere is a small target: most of the time NO ray of the wave passes it (a sphere of radius 1.5 ten units
            // away is hit by 0.6 % of random rays: 70 % of the waves have no such lane), and then the correctly rounded
            // square root and the root selection — a third of the test — are skipped for the whole wave (a scalar branch
            // on the ballot; a lane without an intersection is rejected whatever its t).
            t = 0.0f; ok = false;
            if (kSphereEarlyOut ? (__ballot(has) != 0ull) : true) {
                const float sq = sqrt_rn(d);
                const float i1 = -b + sq, i2 = -b - sq;
                // sq >= 0, so i1 >= i2: 'both >= TMin' is 'i2 >= TMin' and the smaller root is i2; otherwise the reference takes i1
                // unless i1 < 0, and then either choice is below TMin and rejected (Sphere.fs:28-36 in 2 instead of 6 selects)
                t = (i2 >= kTMin) ? i2 : i1;
                ok = has & (t > kTMin);
            }
        } else {
            // Plane.fs:36-40,45-49
            const f3 N = F3(r[0], r[1], r[2]);
            const float num = -r[3] - dot(N, O);
            const float den = dot(N, D);
            const bool has = !(fabsf(den) < kTMin);
            t = div_rn(num, den);
            ok = has & (t > kTMin) & (t <= (kTMax / 1.0f));
            if (type == kGeomPlaneRect) {
                // Plane.fs:19-34 with the per-plane constants hoisted
                const f3 P = O + D * t;                                    // Core.fs:59
                const f3 kern = F3(r[12], r[13], r[14]);
                const f3 v = P - kern;
                const float nx = kern.x + (((v.x * r[4] + v.y * r[5]) + v.z * r[6]) + 0.0f);
                const float ny = kern.y + (((v.x * r[8] + v.y * r[9]) + v.z * r[10]) + 0.0f);
                // xmin <= nx <= xmax  <=>  med3(nx, xmin, xmax) == nx  (bounds ordered: w, h >= 0)
                ok = ok & (__builtin_amdgcn_fmed3f(nx, r[11], r[7]) == nx) & (__builtin_amdgcn_fmed3f(ny, rh[0], r[15]) == ny);
            }
        }
        const bool better = ok & (t < tmin);
        tmin = better ? t : tmin;
        idx = better ? __float_as_int(rh[2]) : idx;
        if (PREFETCH) { geom_wait(nr, nrh); r = nr; rh = nrh; }
    }
    idx_out = idx; t_out = tmin;
}

// MathF.Pow(x, 5) as the oracle states it: x^5 in double, rounded once
RT_DEV float pow5(float x) { double d = (double)x; double d2 = d * d; return (float)(d2 * d2 * d); }

RT_DEV f3 reflect(f3 d, f3 n) { return d - n * (2.0f * dot(d, n)); }

}  // namespace rt
