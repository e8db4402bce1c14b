// rt_kernels.h — the render kernel: one wave64 owns a tile of pixels and traces the whole ray
// tree of its pixel-samples out of a wave-private LDS pool ("wavefront loop inside a wave").
//
// Reference path: Scene.renderScene -> renderPass -> rayTraceBase -> rayTrace (Scene.fs:51-121).
// The reference's estimator is a branching recursion  L = E + (1/n) * sum_i shading_i * L(child_i)
// (Scene.fs:62-70).  It is evaluated here in its throughput form (cf. notes.txt:21-54): every
// pending ray carries beta = prod shading/n along its path and adds beta*E to its pixel-sample
// when it hits an emitter.  Per-ray arithmetic (intersection, hit point, scatter direction) is
// bit-identical to the recursion; only the order of the final additions differs.
//
// Pool: a LIFO of HIT RECORDS (hit point, incoming direction, beta, surface, tree node).  A record
// with n children is expanded lazily: each iteration takes the top records whose children fit in
// 64 lanes (ballot + prefix sum), every lane generates ONE child ray, intersects it against the
// scene and, if the child hits a scattering surface, pushes one new record (ballot + prefix-sum
// compaction).  Records stay sorted by tree level (deepest on top), which bounds the pool to
// slots + 64 * levels records — see DESIGN.md "Pool bound".
#pragma once
#include "rt_device.h"

namespace rt {

struct RenderArgs {
    float* rgba;                    // [ncols*H*4]
    float* depth;                   // [ncols*H]
    unsigned long long* ids;        // [ncols*H] or null
    unsigned long long* stats;      // [0] rays, [1] max depth bits; or null
    int* status;                    // overflow flag
    int nsurf, W, H, spp, max_depth, jitter, bg_nonzero, cap;
    int x_begin, x_step, ncols;
    int tw_log2, th_log2, ntx;      // pixel tile of one wave: 2^tw_log2 columns x 2^th_log2 rows
    int pix_per_group;              // = tile size
    int passes;                     // ceil(spp / S) when spp > S, else 1
    uint32_t seed_lo, seed_hi;
    float bg[3];
    float rot[9];                   // Rotation(cameraWS) rows (Scene.fs:100)
    float origin[3];                // cameraWS.Translation (Scene.fs:102)
    float look_n[3];                // Normalize(lookAt) (Scene.fs:76)
    float fw, fh, aspect, tan_half; // Camera.fs:20-32
};

// meta word of a record
//   bits 0..7 slot | 8..12 level | 13..15 children | 16..17 kind | 18..19 geometry type
RT_DEV uint32_t make_meta(int slot, int level, int n, int kind, int geom)
{
    return (uint32_t)slot | ((uint32_t)level << 8) | ((uint32_t)n << 13) | ((uint32_t)kind << 16) | ((uint32_t)geom << 18);
}

RT_DEV int mbcnt64(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// S = pixel-sample slots one wave traces together (multiple of 64).
// NODE64 = keep the ray-tree node index in 64 bits (needed when max_trace_depth > 14).
template <int S, bool NODE64>
__global__ __launch_bounds__(64) void render_kernel(const RenderArgs a, const GeomRec* __restrict__ geom,
                                                    const ShadeRec* __restrict__ shade)
{
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x;
    const int cap = a.cap;
    constexpr int NF = NODE64 ? 13 : 12;
    float* acc = reinterpret_cast<float*>(smem);   // [S*3] radiance per slot
    uint64_t* slot_ps = reinterpret_cast<uint64_t*>(smem + S * 3);   // [S] pixel-sample index (RNG counter)
    uint32_t* map = smem + S * 5;                   // [64]  child lane -> (record offset, child)
    uint32_t* pool = map + 64;                      // [NF][cap] records, one array per field
    float* fPx = reinterpret_cast<float*>(pool + 0 * cap);
    float* fPy = reinterpret_cast<float*>(pool + 1 * cap);
    float* fPz = reinterpret_cast<float*>(pool + 2 * cap);
    float* fDx = reinterpret_cast<float*>(pool + 3 * cap);
    float* fDy = reinterpret_cast<float*>(pool + 4 * cap);
    float* fDz = reinterpret_cast<float*>(pool + 5 * cap);
    float* fBx = reinterpret_cast<float*>(pool + 6 * cap);
    float* fBy = reinterpret_cast<float*>(pool + 7 * cap);
    float* fBz = reinterpret_cast<float*>(pool + 8 * cap);
    uint32_t* fSurf = pool + 9 * cap;
    uint32_t* fMeta = pool + 10 * cap;
    uint32_t* fNode = pool + 11 * cap;
    uint32_t* fNodeHi = pool + (NF - 1) * cap;      // only dereferenced when NODE64

    const int spp = a.spp;
    const f3 bg = F3(a.bg[0], a.bg[1], a.bg[2]);

    // ---- this wave's pixel tile ------------------------------------------------------------
    const int g = blockIdx.x;
    const int tx = g % a.ntx, ty = g / a.ntx;
    const int G = a.pix_per_group;

    int top = 0;                        // records in the pool (wave-uniform)
    uint32_t nrays = 0;                 // closest-hit queries by this lane
    float maxd = 0.0f;
    bool overflow = false;

    // pixel handled by lane p in the final reduction
    f3 pixsum = F3(0.0f, 0.0f, 0.0f);

    for (int pass = 0; pass < a.passes; pass++) {
        // ================= primaries: rayTraceBase (Scene.fs:75-95) ==========================
        for (int r = 0; r < S / 64; r++) {
            const int slot = r * 64 + lane;
            int p, s;
            bool valid;
            if (spp <= S) { p = slot / spp; s = slot - p * spp; valid = p < G; }
            else          { p = 0; s = pass * S + slot; valid = s < spp; }
            const int k = (tx << a.tw_log2) + (p & ((1 << a.tw_log2) - 1));
            const int py = (ty << a.th_log2) + (p >> a.tw_log2);
            valid = valid && k < a.ncols && py < a.H;
            const int px = a.x_begin + k * a.x_step;
            const uint64_t pixel = (uint64_t)px * (uint64_t)a.H + (uint64_t)py;
            const uint64_t ps = pixel * (uint64_t)spp + (uint64_t)s;

            f3 col = F3(0.0f, 0.0f, 0.0f);
            bool push = false;
            f3 P = F3(0, 0, 0), D = F3(0, 0, 0);
            int idx = -1; uint32_t info = 0;
            if (valid) {
                // Camera.fs:8-39 + Scene.fs:98-102
                f3 rv = F3(0.0f, 0.0f, 0.0f);
                if (a.jitter) rv = unit_sphere_sample(a.seed_lo, a.seed_hi, pixel);
                float xoff = rv.x / a.fw, yoff = rv.y / a.fh;
                float pxp = ((float)px + 0.5f) + xoff, pyp = ((float)py + 0.5f) + yoff;
                float xn = pxp / a.fw, yn = pyp / a.fh;
                float xs = 2.0f * xn - 1.0f, ys = 1.0f - 2.0f * yn;
                float xc = xs * a.aspect * a.tan_half, yc = ys * a.tan_half;
                f3 dirCS = normalize(F3(xc, yc, -1.0f));
                f3 dirT = F3((dirCS.x * a.rot[0] + dirCS.y * a.rot[3]) + dirCS.z * a.rot[6],
                             (dirCS.x * a.rot[1] + dirCS.y * a.rot[4]) + dirCS.z * a.rot[7],
                             (dirCS.x * a.rot[2] + dirCS.y * a.rot[5]) + dirCS.z * a.rot[8]);
                D = normalized_or_fail(normalize(dirT));
                f3 O = F3(a.origin[0], a.origin[1], a.origin[2]);
                float dot_view = dot(F3(a.look_n[0], a.look_n[1], a.look_n[2]), D);   // Scene.fs:76

                float t;
                closest_hit(geom, a.nsurf, O, D, idx, t);
                nrays++;
                // Scene.fs:79: the closest candidate is re-tested with dotView; planes then cull
                // at TMax/dotView (Plane.fs:46).  A failed re-test is a miss, not the next hit.
                bool hit = idx >= 0;
                if (hit) {
                    const float4 em = *reinterpret_cast<const float4*>(&shade[idx].emission[0]);
                    info = __float_as_uint(em.w);
                    if (((info >> 5) & 3) != kGeomSphere) hit = t <= (kTMax / dot_view);
                    if (hit) {
                        if (s == 0) {                                      // Scene.fs:80
                            const size_t o = (size_t)k * (size_t)a.H + (size_t)py;
                            a.depth[o] = t;
                            if (a.ids) a.ids[o] = shade[idx].id;
                            maxd = fmaxf(maxd, t);
                        }
                        col = F3(em.x, em.y, em.z);
                        const int n = (info >> 2) & 7;
                        push = n > 0;
                        P = O + D * t;
                    }
                }
                if (!hit) {
                    col = bg;
                    if (s == 0) {
                        const size_t o = (size_t)k * (size_t)a.H + (size_t)py;
                        a.depth[o] = kFltMax;                               // Scene.fs:29
                        if (a.ids) a.ids[o] = 0ull;
                    }
                }
            }
            acc[slot * 3 + 0] = col.x; acc[slot * 3 + 1] = col.y; acc[slot * 3 + 2] = col.z;
            slot_ps[slot] = ps;
            // level-0 records: at most S of them, pushed in reverse lane order like every push
            const unsigned long long pm = __ballot(push);
            const int npush = __popcll(pm);
            if (top + npush > cap) { overflow = true; push = false; }
            else if (push) {
                const int pos = top + (npush - 1 - mbcnt64(pm));
                fPx[pos] = P.x; fPy[pos] = P.y; fPz[pos] = P.z;
                fDx[pos] = D.x; fDy[pos] = D.y; fDz[pos] = D.z;
                fBx[pos] = 1.0f; fBy[pos] = 1.0f; fBz[pos] = 1.0f;
                fSurf[pos] = (uint32_t)idx;
                fMeta[pos] = make_meta(slot, 0, (info >> 2) & 7, info & 3, (info >> 5) & 3);
                fNode[pos] = 0u;
                if (NODE64) fNodeHi[pos] = 0u;
            }
            if (top + npush <= cap) top += npush;
        }
        __syncthreads();

        // ================= bounces: rayTrace (Scene.fs:51-72), one child ray per lane ==========
        while (top > 0) {
            // ---- take the top records whose children fit in 64 lanes (prefix sum over ballots)
            const int ridx = top - 1 - lane;
            const uint32_t rmeta = (ridx >= 0) ? fMeta[ridx] : 0u;
            const int rn = (rmeta >> 13) & 7;
            const unsigned long long b0 = __ballot(rn & 1), b1 = __ballot(rn & 2), b2 = __ballot(rn & 4);
            const int excl = mbcnt64(b0) + 2 * mbcnt64(b1) + 4 * mbcnt64(b2);
            const bool take = rn > 0 && excl + rn <= 64;
            const unsigned long long tk = __ballot(take);
            const int ntake = __popcll(tk);
            const int nchild = __popcll(b0 & tk) + 2 * __popcll(b1 & tk) + 4 * __popcll(b2 & tk);
            if (take)
                for (int c = 0; c < rn; c++) map[excl + c] = (uint32_t)(lane * 4 + c);
            __syncthreads();
            const bool active = lane < nchild;
            const uint32_t mp = active ? map[lane] : 0u;
            const int rec = top - 1 - (int)(mp >> 2);
            const int child = (int)(mp & 3);
            top -= ntake;

            bool push = false;
            f3 P = F3(0, 0, 0), Dn = F3(0, 0, 0), beta = F3(0, 0, 0);
            int hidx = -1; uint32_t hinfo = 0, meta = 0; uint64_t cnode = 0; int clevel = 0;
            if (active) {
                const f3 O = F3(fPx[rec], fPy[rec], fPz[rec]);
                const f3 Din = F3(fDx[rec], fDy[rec], fDz[rec]);
                const f3 pbeta = F3(fBx[rec], fBy[rec], fBz[rec]);
                const uint32_t surf = fSurf[rec];
                meta = fMeta[rec];
                uint64_t node = fNode[rec];
                if (NODE64) node |= (uint64_t)fNodeHi[rec] << 32;
                const int slot = meta & 0xff, level = (meta >> 8) & 31, n = (meta >> 13) & 7;
                const int kind = (meta >> 16) & 3, gty = (meta >> 18) & 3;
                clevel = level + 1;
                cnode = 4ull * node + (uint64_t)child + 1ull;

                const uint64_t ps = slot_ps[slot];   // the RNG counter

                const ShadeRec* sr = shade + surf;
                const float4 al = *reinterpret_cast<const float4*>(&sr->albedo[0]);
                const float4 nc = *reinterpret_cast<const float4*>(&sr->nc[0]);
                const f3 albedo = F3(al.x, al.y, al.z);

                // Hitable.NormalForSurfacePoint: Sphere.fs:37-38, Plane.fs:50-51
                f3 normal = F3(nc.x, nc.y, nc.z);
                if (gty == kGeomSphere) normal = normalize(O - normal);

                bool alive = true;
                f3 shd;                 // shading_i; folded into beta below
                float inv_n;
                if (kind == kKindDielectric) {
                    // Surface.fs:124-153 CalcFresnel + :168-181 GenerateSamples
                    normal = normalize(normal);
                    const f3 refl = normalize(reflect(Din, normal));
                    float eta_i, eta_t; f3 fn;
                    if (dot(Din, normal) > 0.0f) { eta_i = al.w; eta_t = 1.0f; fn = neg(normal); }
                    else                         { eta_i = 1.0f; eta_t = al.w; fn = normal; }
                    const float cos_i = dot(Din, neg(fn));
                    const float eta = eta_i / eta_t;
                    const float disc = 1.0f - (eta * eta) * (1.0f - (cos_i * cos_i));
                    const bool refracted = disc > 0.0f;
                    float F = 1.0f;
                    if (refracted) {
                        const float q = (eta_t - eta_i) / (eta_t + eta_i);
                        const float R0 = q * q;
                        F = R0 + (1.0f - R0) * pow5(1.0f - cos_i);
                    }
                    const bool single = round_digits(F, 1e3f) == 1.0f;
                    if (child == 0) {
                        Dn = normalized_or_fail(refl);
                        shd = albedo * F;                    // (2*albedo*F)/2 == albedo*F exactly
                    } else {
                        alive = !single;
                        const f3 q = (Din + fn * cos_i) * eta - fn * sqrtf(disc);
                        Dn = normalized_or_fail(normalize(q));
                        shd = albedo * (1.0f - F);
                    }
                    inv_n = 1.0f;
                } else {
                    // Lambertian.Scatter (Surface.fs:66-83) / Metal.Scatter (Surface.fs:93-109)
                    if (kind == kKindMetal) normal = normalize(normal);
                    const f3 rn = hemisphere_sample(a.seed_lo, a.seed_hi, ps, cnode);
                    f3 nt, nb;
                    frame_around(normal, nt, nb);
                    // Vector4.Transform(rn, ChangeOfBase nt n nb): rows nt, n, nb; M4x = 0
                    const f3 ns = F3(((rn.x * nt.x + rn.y * normal.x) + rn.z * nb.x) + 0.0f,
                                     ((rn.x * nt.y + rn.y * normal.y) + rn.z * nb.y) + 0.0f,
                                     ((rn.x * nt.z + rn.y * normal.z) + rn.z * nb.z) + 0.0f);
                    if (kind == kKindLambert) {
                        Dn = normalized_or_fail(normalize(ns));
                        const float pdf = 1.0f / (2.0f * 3.14159265358979323846f);
                        const f3 brdf = divs(albedo, 3.14159265358979323846f);
                        shd = brdf * (rn.y / pdf);
                    } else {
                        const float fuzz = al.w;
                        const f3 mod = normalize(normal * (1.0f - fuzz) + ns * fuzz);
                        Dn = normalized_or_fail(normalize(reflect(Din, mod)));
                        shd = albedo * (1.0f / 1.0f);
                    }
                    inv_n = 1.0f / (float)n;
                }
                beta = (pbeta * shd) * inv_n;

                if (alive) {
                    if (clevel > a.max_depth + 1) {
                        // depth guard (Scene.fs:52-54): the child returns backgroundColor unintersected
                        atomicAdd(&acc[slot * 3 + 0], beta.x * bg.x);
                        atomicAdd(&acc[slot * 3 + 1], beta.y * bg.y);
                        atomicAdd(&acc[slot * 3 + 2], beta.z * bg.z);
                    } else {
                        float t;
                        closest_hit(geom, a.nsurf, O, Dn, hidx, t);
                        nrays++;
                        if (hidx >= 0) {
                            const float4 em = *reinterpret_cast<const float4*>(&shade[hidx].emission[0]);
                            hinfo = __float_as_uint(em.w);
                            if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) {
                                atomicAdd(&acc[slot * 3 + 0], beta.x * em.x);
                                atomicAdd(&acc[slot * 3 + 1], beta.y * em.y);
                                atomicAdd(&acc[slot * 3 + 2], beta.z * em.z);
                            }
                            const int hn = (hinfo >> 2) & 7;
                            // children of a hit at level max_depth+1 are all cut by the depth guard;
                            // with a black background they contribute exactly 0 and are skipped
                            push = hn > 0 && (clevel <= a.max_depth || a.bg_nonzero);
                            P = O + Dn * t;
                        } else if (a.bg_nonzero) {
                            atomicAdd(&acc[slot * 3 + 0], beta.x * bg.x);
                            atomicAdd(&acc[slot * 3 + 1], beta.y * bg.y);
                            atomicAdd(&acc[slot * 3 + 2], beta.z * bg.z);
                        }
                    }
                }
            }
            // ---- compaction: hits on scattering surfaces become records, deepest on top
            const unsigned long long pm = __ballot(push);
            const int npush = __popcll(pm);
            if (top + npush > cap) { overflow = true; }
            else {
                if (push) {
                    const int pos = top + (npush - 1 - mbcnt64(pm));
                    fPx[pos] = P.x; fPy[pos] = P.y; fPz[pos] = P.z;
                    fDx[pos] = Dn.x; fDy[pos] = Dn.y; fDz[pos] = Dn.z;
                    fBx[pos] = beta.x; fBy[pos] = beta.y; fBz[pos] = beta.z;
                    fSurf[pos] = (uint32_t)hidx;
                    int hn = (hinfo >> 2) & 7;
                    if ((hinfo & 3) == kKindDielectric) hn = 2;
                    fMeta[pos] = make_meta(meta & 0xff, clevel, hn, hinfo & 3, (hinfo >> 5) & 3);
                    fNode[pos] = (uint32_t)cnode;
                    if (NODE64) fNodeHi[pos] = (uint32_t)(cnode >> 32);
                }
                top += npush;
            }
            __syncthreads();
        }

        // ====== per-pixel sum over samples in sample order, mean, gamma 2 (Scene.fs:109,114) ======
        {
            const int cnt = (spp <= S) ? spp : min(S, spp - pass * S);
            for (int pb = 0; pb < G; pb += 64) {
                const int p = pb + lane;
                if (p < G) {
                    const int base = (spp <= S) ? p * spp : 0;
                    f3 sum = pixsum;                       // carried across passes only when G == 1
                    for (int i = 0; i < cnt; i++) {
                        const f3 c = F3(acc[(base + i) * 3 + 0], acc[(base + i) * 3 + 1], acc[(base + i) * 3 + 2]);
                        sum = (pass == 0 && i == 0) ? c : sum + c;
                    }
                    pixsum = sum;
                    const int k = (tx << a.tw_log2) + (p & ((1 << a.tw_log2) - 1));
                    const int py = (ty << a.th_log2) + (p >> a.tw_log2);
                    if (pass == a.passes - 1 && k < a.ncols && py < a.H) {
                        const f3 avg = divs(sum, (float)spp);
                        const size_t o = (size_t)k * (size_t)a.H + (size_t)py;
                        *reinterpret_cast<float4*>(a.rgba + 4 * o) =
                            make_float4(sqrtf(avg.x), sqrtf(avg.y), sqrtf(avg.z), 1.0f);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ================= statistics ================================================================
    if (a.stats) {
        unsigned long long r = nrays;
        float m = maxd;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_xor(r, off);
            m = fmaxf(m, __shfl_xor(m, off));
        }
        if (lane == 0) {
            atomicAdd(a.stats + 0, r);
            atomicMax(reinterpret_cast<unsigned int*>(a.stats + 1), __float_as_uint(m));
        }
    }
    if (overflow && lane == 0) atomicExch(a.status, 1);
}

// findClosestIntersection as a batch query (Surface.fs:49-58): one ray per thread.
__global__ __launch_bounds__(256) void find_closest_kernel(const GeomRec* __restrict__ geom, int nsurf, long long nrays,
                                                           const float* __restrict__ o, const float* __restrict__ d,
                                                           float* __restrict__ t_out, int* __restrict__ idx_out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrays) return;
    int idx; float t;
    closest_hit(geom, nsurf, F3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), F3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), idx, t);
    t_out[i] = t; idx_out[i] = idx;
}

}  // namespace rt
