// Device-side scene layout and the reference-precision ("exact") primitive
// routines.  Everything here keeps the reference's evaluation order and its
// double -> float rounding points (raytracer.cpp:312-357, 393-414, 32-34);
// the translation unit is built with -ffp-contract=off so nothing is fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_MISS_T 3.402823466e+38f          // MAXFLOAT: the reference's miss sentinel
#define RT_MISS_KEY 0x7f7fffffffffffffull   // (bits(MAXFLOAT) << 32) | 0xffffffff

#define RT_GRP_IDENTITY 0x40000000
struct d3 { double x, y, z; };

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const v16f cv16f;

// Regular triangle, FP32 filter form: 16 floats (layout at filter_z2 in rt_kernels.h).
// Reference-precision form of the same triangle (128 bytes):
struct __attribute__((aligned(16))) TriExact {
    double v1[3];      // vertex1
    double e1[3];      // edge1 = vertex3 - vertex1 (raytracer.cpp:288)
    double e2[3];      // edge2 = vertex2 - vertex1 (raytracer.cpp:289)
    double n[3];       // (edge2 ^ edge1).normalize() (raytracer.cpp:313)
    float uu, uv, vv, D;   // raytracer.cpp:334-340, each rounded as the reference does
    int32_t obj;       // index in Scene::objects
    int32_t grp;       // index of the transform group; RT_GRP_IDENTITY set: the group's transform is the identity
                       // (whoever only needs to know that is spared the load of the group's record)
    float box_pad;     // how far outside the triangle the reference's own rounding can accept a point
    int32_t hole;      // 1: no triangle here (a slot between the end of an axis range and the next multiple of 64);
                       // 2 (irregular triangles only): the unit normal is not a number -- the reference's answer for this
                       // triangle is then the same for every ray (see brute_scan), and no ray is tested against it
};

struct __attribute__((aligned(16))) SphereRec {
    double c[3];
    double r2;         // (double)radius * (double)radius, pow(radius, 2) at raytracer.cpp:396
    int32_t obj, xf, identity, pad;
};

// FP32 form of a sphere for the conservative pre-test of k_brute (see sphere_may_hit): the inverse
// transform's upper three rows, centre and squared radius as floats, and the magnitudes the error
// bound of the test needs.
struct __attribute__((aligned(16))) SphereFilt {
    float m[12];       // rows 0..2 of M^-1 (row-major 3 x 4): object-space point = m * (p, 1), direction = m * (v, 0)
    float c[3], r2;    // centre, radius^2
    float norm;        // largest row sum of |m[i][0..2]| (how much the transform can magnify a coordinate's rounding)
    float tabs;        // |m[0][3]| + |m[1][3]| + |m[2][3]| + |c|_1 (magnitudes that enter the object-space offset p - c)
    int32_t identity;  // the transform is the identity (rays stay as they are)
    int32_t pad;
    float wc[3], wr;   // a ball in world space around the transformed sphere (nearest-hit queries; see spheres_in_reach)
};

// Thresholds of the FP32 candidate filter for one kind of query against one transform
// group.  A pair (ray, triangle) may be dropped only when the reference-precision test is
// certain to miss; the thresholds follow from the magnitudes the group's rays and vertices
// can have (derived on the host when the scene is uploaded).
struct GroupFilter {
    float r_lo;            // ray parameter below this: the reference's r < eps reject is certain
    float a_zero;          // |plane distance of the origin| at or below this: a == 0.0f is possible
    float b_hi;            // |n.dir| above this: the reference's grazing branch (raytracer.cpp:319) is not taken
    float origin_bound;    // the thresholds hold for rays with |origin component| <= this
    float dir_len2_bound;  // ... and |dir|^2 <= this; other rays take the reference-precision scan
    float e_a, e_b;        // error bounds of the FP32 plane distance a and of b = n.dir
    float r_hi;            // ray parameter at or above this: the reference's r < eps reject is certainly not taken
};

// Triangles that share one transform.  Filter slots [fax[0], fax[3]) are regular
// triangles; exact slots [ibeg, iend) hold the group's irregular triangles
// (degenerate, non-finite, or too ill-conditioned for the FP32 filter), which
// are always tested in reference precision.
struct Group {
    int32_t xf, identity;
    int32_t fax[4];    // filter slots from fax[k] on: triangles whose normal's dominant axis is k; every fax[k] is a
                       // multiple of 64, [fax[0], fax[3]) is the group's whole slot range (holes included)
    int32_t fcnt[3];   // ... and how many: the range of axis k is [fax[k], fax[k] + fcnt[k]); holes up to fax[k + 1]
    int32_t pad;
    int32_t ibeg, iend;
    GroupFilter q[2];  // [0] nearest hit: rays in the group's object space (Scene::eps);
                       // [1] shadow: world-space rays against the untransformed vertices (0.1f)
};

struct DevMaterial { double ka[3], kd[3], ks[3], kr[3]; double sp; };
struct DevLight { double color[3], pos[3]; int32_t falloff, is_point; };

struct DevScene {
    const v16f* tri_filt;        // [n_regular + 1]: one padding record at the end
    const TriExact* tri_exact;
    const SphereRec* spheres;
    const SphereFilt* sphere_filt;   // [n_spheres], same order
    const Group* groups;
    const double* xforms;        // [n_transforms][48]: M, inverse, inverse-transpose (row-major)
    const int32_t* xf_identity;  // [n_transforms] 1 when M is bit-for-bit the identity
    const int32_t* obj_kind;
    const int32_t* obj_slot;     // triangle: slot in tri_exact; sphere: slot in spheres
    const int32_t* obj_material;
    const int32_t* obj_transform;
    const int32_t* obj_interp;
    const double* obj_geom;
    const double* obj_normals;
    const DevMaterial* materials;
    const DevLight* lights;
    int32_t n_regular, n_irregular, n_spheres, n_groups;
    int32_t n_objects, n_lights, n_transforms, last_xf;
    int32_t width, height, max_depth;
    int32_t n_prims_nominal;     // primitives of the scene (the copies that fill up the axis ranges not counted)
    float eps, shadow_eps;
    int32_t robust_filter;       // 1: filter records carry per-test margins (see filter_z2)
    int32_t n_void_normal;       // irregular triangles whose unit normal is not a number (TriExact::hole == 2, see brute_scan)
    double cam_eye[3], cam_ll[3], cam_lr[3], cam_ul[3], cam_ur[3];
    double ambient[3];
};

// ---- double-precision vector helpers (same operation order as algebra3.h) ----
__device__ __forceinline__ d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 ld3(const double* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ d3 add3(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 sub3(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 neg3(d3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ d3 scl3(d3 a, double d) { return mk3(d * a.x, d * a.y, d * a.z); }
__device__ __forceinline__ d3 mul3(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross3(d3 a, d3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec3::normalize multiplies by 1/length (algebra3.h:638-662)
__device__ __forceinline__ d3 nrm3(d3 a) {
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    double inv = 1. / len;
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
// float distance(vec3&, vec3&) (raytracer.cpp:32-34)
__device__ __forceinline__ float dist_f(d3 a, d3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return (float)sqrt(dx * dx + dy * dy + dz * dz);
}
// vec3(M * vec4(p, 1)): the vec3 constructor divides by w (algebra3.h:853, 610)
__device__ __forceinline__ d3 xf_point(const double* m, d3 p) {
    double o0 = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0;
    double o1 = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0;
    double o2 = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0;
    double o3 = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * 1.0;
    return mk3(o0 / o3, o1 / o3, o2 / o3);
}
// vec3(M * vec4(v, 0), VW): w dropped (algebra3.h:614-621)
__device__ __forceinline__ d3 xf_dir(const double* m, d3 v) {
    double o0 = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 0.0;
    double o1 = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 0.0;
    double o2 = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 0.0;
    return mk3(o0, o1, o2);
}

// Triangle::intersection (raytracer.cpp:312-357) on a precomputed record.
__device__ __forceinline__ float tri_intersect_exact(const TriExact& T, d3 pos, d3 vec, float eps,
                                                     bool* in_plane_origin = nullptr) {
    d3 v1 = ld3(T.v1), e1 = ld3(T.e1), e2 = ld3(T.e2), nrm = ld3(T.n);
    d3 w0 = sub3(pos, v1);
    float a = (float)(-dot3(nrm, w0));
    float b = (float)dot3(nrm, vec);
    float r = a / b;
    if (fabsf(b) < eps) {
        if (a == 0.0f) {                 // origin in the plane, ray grazing it: "hit" at r = 0/b (raytracer.cpp:320)
            if (in_plane_origin) *in_plane_origin = true;
            return r;
        }
        return RT_MISS_T;
    }
    if (r < 0.0f) return RT_MISS_T;
    d3 isect = add3(pos, scl3(vec, (double)r));
    d3 w = sub3(isect, v1);
    float wu = (float)dot3(w, e1);
    float wv = (float)dot3(w, e2);
    float s = (T.uv * wv - T.vv * wu) / T.D;
    if (s < 0.0f || s > 1.0f) return RT_MISS_T;
    float t = (T.uv * wu - T.uu * wv) / T.D;
    if (t < 0.0f || (s + t) > 1.0f) return RT_MISS_T;
    if (r < eps) return RT_MISS_T;
    return r;
}

// Sphere::intersection (raytracer.cpp:393-414).
__device__ __forceinline__ float sph_intersect_exact(const SphereRec& S, d3 pos, d3 vec, float eps) {
    d3 pc = sub3(pos, ld3(S.c));
    double dpc = dot3(vec, pc);
    double dd = dot3(vec, vec);
    float disc = (float)(dpc * dpc - (dd * (dot3(pc, pc) - S.r2)));
    if (disc < 0) return RT_MISS_T;
    float sq = sqrtf(disc);
    double mdpc = dot3(neg3(vec), pc);
    float t1 = (float)((mdpc + (double)sq) / dd);
    float t2 = (float)((mdpc - (double)sq) / dd);
    if (t1 < eps && t2 < eps) return RT_MISS_T;
    d3 p1 = add3(pos, scl3(vec, (double)t1));
    d3 p2 = add3(pos, scl3(vec, (double)t2));
    float d1 = dist_f(pos, p1), d2 = dist_f(pos, p2);
    return d1 < d2 ? t1 : t2;
}

// Conservative FP32 pre-test of a ray against a sphere: false only when Sphere::intersection
// (raytracer.cpp:393-414) is certain to return its miss value -- the discriminant is negative beyond the
// rounding of this evaluation, or both roots lie below eps (the ray points away from the sphere, or
// leaves it within eps: `t1 < eps && t2 < eps`, raytracer.cpp:402).  Everything else, NaNs included,
// goes on to the reference-precision test.  (wp, wv): the world-space ray as floats; p1, v1: the sums of
// the absolute values of their components; transformed: take the ray into the sphere's object space
// first (nearest-hit queries; shadow rays meet the untransformed sphere, raytracer.cpp:553).
__device__ __forceinline__ bool sphere_may_hit(const SphereFilt& F, const bool transformed, float wpx, float wpy, float wpz,
                                               float wvx, float wvy, float wvz, float p1, float v1, float eps) {
    const float EPS32 = 1.1920929e-7f;
    float px = wpx, py = wpy, pz = wpz, vx = wvx, vy = wvy, vz = wvz;
    float norm = 1.0f;
    if (transformed) {
        px = __builtin_fmaf(F.m[0], wpx, __builtin_fmaf(F.m[1], wpy, __builtin_fmaf(F.m[2], wpz, F.m[3])));
        py = __builtin_fmaf(F.m[4], wpx, __builtin_fmaf(F.m[5], wpy, __builtin_fmaf(F.m[6], wpz, F.m[7])));
        pz = __builtin_fmaf(F.m[8], wpx, __builtin_fmaf(F.m[9], wpy, __builtin_fmaf(F.m[10], wpz, F.m[11])));
        vx = __builtin_fmaf(F.m[0], wvx, __builtin_fmaf(F.m[1], wvy, F.m[2] * wvz));
        vy = __builtin_fmaf(F.m[4], wvx, __builtin_fmaf(F.m[5], wvy, F.m[6] * wvz));
        vz = __builtin_fmaf(F.m[8], wvx, __builtin_fmaf(F.m[9], wvy, F.m[10] * wvz));
        norm = F.norm;
    }
    // how far the FP32 offset p - c and direction can be from the reference's doubles, per component
    // (inputs rounded to float, matrix entries rounded to float, four roundings per coordinate)
    const float dp = 16.0f * EPS32 * (norm * p1 + F.tabs), dv = 16.0f * EPS32 * norm * v1;
    const float cx = px - F.c[0], cy = py - F.c[1], cz = pz - F.c[2];
    const float dpc = __builtin_fmaf(vx, cx, __builtin_fmaf(vy, cy, vz * cz));
    const float dd = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float pp = __builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz));
    const float disc = __builtin_fmaf(dpc, dpc, -dd * (pp - F.r2));
    // bounds on |p - c| and |v| (1.75 > sqrt(3): the per-component bounds as a length)
    const float L = sqrtf(pp) + 1.75f * dp, V = sqrtf(dd) + 1.75f * dv;
    const float e_dpc = 1.75f * (V * dp + L * dv) + 8.0f * EPS32 * V * L;
    const float e_dd = 2.0f * V * 1.75f * dv + 8.0f * EPS32 * V * V;
    const float e_pp = 2.0f * L * 1.75f * dp + 8.0f * EPS32 * (L * L + F.r2);
    const float E = 2.0f * ((2.0f * fabsf(dpc) + e_dpc) * e_dpc + e_dd * (pp + F.r2) + (dd + e_dd) * e_pp +
                            8.0f * EPS32 * (dpc * dpc + dd * (pp + F.r2))) + 1e-30f;
    const bool no_root = disc < -E;
    // the larger root, from above: (-dpc + sqrt(disc)) / dd
    const float top = -dpc + e_dpc + sqrtf(fmaxf(disc + E, 0.0f)) * (1.0f + 4.0f * EPS32);
    const float dd_lo = dd - e_dd;
    const bool behind = (eps > 0.0f) & ((top < 0.0f) | ((dd_lo > 0.0f) & (top < eps * (1.0f - 1e-3f) * dd_lo)));
    return !(no_root | behind);
}

// What the 64 rays of a wave have in common, for dropping spheres for all of them at once: every origin lies within
// `radius` of `m`, every direction within the angle alpha of the unit vector `d` (cos_a, sin_a; alpha < 90 degrees), so
// every ray stays inside the cone of that angle around `d` whose apex lies `back` = radius / sin_a behind `m` (the cone
// that touches the ball of the origins), and in front of the plane through m - radius * d across `d`.
struct RayCone {
    float mx, my, mz, radius;
    float dx, dy, dz, back;
    float cos_a, sin_a;
    float p1_max;          // largest |x| + |y| + |z| of an origin (what the rays lost when they became floats scales with it)
    bool usable;           // false: the rays have no such cone (too far apart in direction, or not finite); nothing is dropped
};

// (all lanes of the wave call this; `valid`: the lane holds a ray; p: origin, v: direction as floats)
__device__ __forceinline__ RayCone ray_cone_of_wave(bool valid, float px, float py, float pz, float vx, float vy, float vz, float p1) {
    RayCone C;
    const float len2 = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const bool fine = !valid || (p1 < INFINITY && len2 > 1e-30f && len2 < INFINITY);      // (false for NaNs)
    const float inv = valid && fine ? 1.0f / sqrtf(len2) : 0.0f;
    const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
    float lo[3] = {valid ? px : INFINITY, valid ? py : INFINITY, valid ? pz : INFINITY};
    float hi[3] = {valid ? px : -INFINITY, valid ? py : -INFINITY, valid ? pz : -INFINITY};
    float sx = valid ? ux : 0.f, sy = valid ? uy : 0.f, sz = valid ? uz : 0.f, pm = valid ? p1 : 0.f;
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off);
        pm = fmaxf(pm, __shfl_xor(pm, off));
    }
    C.mx = 0.5f * (lo[0] + hi[0]); C.my = 0.5f * (lo[1] + hi[1]); C.mz = 0.5f * (lo[2] + hi[2]);
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    C.radius = 0.5f * sqrtf(__builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez))) * 1.00001f + 1e-6f * pm;
    const float sl = sqrtf(__builtin_fmaf(sx, sx, __builtin_fmaf(sy, sy, sz * sz)));
    const float si = sl > 0.f ? 1.0f / sl : 0.f;
    C.dx = sx * si; C.dy = sy * si; C.dz = sz * si;
    float low = valid ? __builtin_fmaf(ux, C.dx, __builtin_fmaf(uy, C.dy, uz * C.dz)) : 1.0f;
    for (int off = 32; off > 0; off >>= 1) low = fminf(low, __shfl_xor(low, off));
    C.cos_a = fminf(low, 1.0f) - 1e-5f;               // (the angle from above: a cosine too small)
    C.sin_a = sqrtf(fmaxf(1.0f - C.cos_a * C.cos_a, 0.f));
    C.back = C.radius / C.sin_a;                      // (sin_a >= 4e-3 by the line above)
    C.p1_max = pm;
    C.usable = !__any(!fine) && sl > 0.f && C.cos_a > 0.2f && C.radius < INFINITY;
    return C;
}

// May a ray of the wave meet the ball of radius r around c?  False only when the ball (grown by what the floats of this
// test and of the rays can be off) lies outside the cone, or behind the plane the rays start from (then both roots of
// raytracer.cpp:399-401 are negative, below an eps that is not: `behind_counts`).  NaNs compare false: kept.
__device__ __forceinline__ bool ball_in_reach(const RayCone& C, float cx, float cy, float cz, float r, bool behind_counts) {
    const float wx0 = cx - C.mx, wy0 = cy - C.my, wz0 = cz - C.mz;
    const float wx = __builtin_fmaf(C.dx, C.back, wx0), wy = __builtin_fmaf(C.dy, C.back, wy0), wz = __builtin_fmaf(C.dz, C.back, wz0);   // from the apex
    const float along = __builtin_fmaf(wx, C.dx, __builtin_fmaf(wy, C.dy, wz * C.dz));
    const float kx = wy * C.dz - wz * C.dy, ky = wz * C.dx - wx * C.dz, kz = wx * C.dy - wy * C.dx;
    const float across = sqrtf(__builtin_fmaf(kx, kx, __builtin_fmaf(ky, ky, kz * kz)));
    const float dist = sqrtf(__builtin_fmaf(wx, wx, __builtin_fmaf(wy, wy, wz * wz)));
    const float slack = 1e-4f * (dist + C.radius + r) + 1e-5f * (C.p1_max + fabsf(cx) + fabsf(cy) + fabsf(cz));
    const bool outside = __builtin_fmaf(across, C.cos_a, -along * C.sin_a) > r + slack;
    const float ahead = __builtin_fmaf(wx0, C.dx, __builtin_fmaf(wy0, C.dy, wz0 * C.dz));
    const bool behind = behind_counts && (ahead + r + slack < -C.radius);
    return !(outside || behind);
}

// The nearest-hit metric of Scene::traceRay (raytracer.cpp:522-524): distance
// from the OBJECT-space origin to the WORLD-space hit point.
__device__ __forceinline__ float hit_metric(const double* m, int identity, d3 rpos, d3 rvec, float t, d3* world) {
    d3 p = add3(rpos, scl3(rvec, (double)t));
    if (!identity) p = xf_point(m, p);
    *world = p;
    return dist_f(rpos, p);
}

__device__ __forceinline__ unsigned long long make_key(float d, int obj) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)obj;
}
