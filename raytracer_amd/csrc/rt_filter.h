// The FP32 candidate filter and the bounded scans built around it: the triangle test two rays at a time
// (filter_z2), the bounding questions of clusters, groups and batches (filter_range), the task loop (k_filter).
#pragma once

#include "rt_trace_args.h"

// The FP32 filter for one (ray, triangle) pair: it may reject only when the reference-precision
// test of the pair is certain to miss.  Record f (wave-uniform, in scalar registers):
//   f[0..3]   unit normal n and d0 = n.v1
//   f[4..6]   s as an affine function of the hit point's two coordinates OTHER than the
//             normal's dominant axis (the third follows from the plane equation),
//   f[7..9]   t likewise, f[10..12] u = 1 - s - t likewise; every offset includes the margin,
//   f[13,14]  per-test margin coefficients (ROBUST form), f[15] the largest static margin.
// (pu, pv, vu, vv): those two coordinates of the ray origin and direction.
// Q: the query's thresholds for the triangle's group (GroupFilter).
//
// With b = n.dir, a = d0 - n.origin, r = a / b and (s, t, u) evaluated at origin + r dir, the
// pair is kept when z = min(s, t, u, r - r_lo) >= 0, or when the origin may lie exactly in the
// plane (|a| <= a_zero; raytracer.cpp:319-325 -- a ray too parallel to the plane needs no test
// of its own: the reference then misses unless that is the case).  Per ray this function
// produces the ONE number z, so that the scan loop can ask "does any of my 8 rays keep this
// triangle?" with a maximum over the z's and a single comparison, instead of three comparisons
// per ray whose results have to be merged on the scalar unit.  min/max drop a NaN operand, so
// a NaN among s, t, u can only make the test keep more; z itself is NaN only for 0 * inf
// (origin in the plane AND direction in it), a pair the in-plane term keeps -- *a_min collects
// the smallest |a| of the lane's rays for it.
//
// Two rays at a time: MI355X reaches its FP32 vector peak only through the packed
// instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of FMAs per issue slot), so the
// lane's rays are kept as float2 pairs and every multiply-add of the test is a packed one.
// Reciprocal and min/max have no packed form and stay per ray.
//
// ROBUST = false: the margins in the offsets are the worst case over every ray the group can
// meet (the most grazing one included) -- the cheapest test, right for scenes whose triangles
// are large against the FP32 resolution of the scene.
// ROBUST = true: the offsets carry only the ray-independent part; the part that grows with
// 1/|n.dir| is added per test from f[13 + MODE] (nearest hit / shadow), and the in-plane-origin
// term also requires a grazing ray, so a ray leaving a flat wall does not list every triangle
// of that wall.  One more instruction per ray; chosen per scene when it is uploaded.
//
// SHARED: what all rays of the task have in common, with the matching per-triangle constant
// `shared` computed on the host in double precision (at least as accurate as the FP32
// evaluation it replaces, so the same margins hold):
//   1  the origin (camera rays): shared = a = d0 - n.origin, the three multiply-adds go;
//   2  the direction (shadow rays towards a directional light): shared = 1 / (n.dir), the
//      three multiply-adds and the reciprocal go.
struct FilterQuery { float r_lo, a_zero, b_hi; };

template <int MODE, bool ROBUST, int SHARED>
__device__ __forceinline__ void filter_z2(const v16f f, v2f px, v2f py, v2f pz, v2f vx, v2f vy, v2f vz,
                                          v2f pu, v2f pv, v2f vu, v2f vv, const FilterQuery Q, const float shared,
                                          float* z0, float* z1, float* a_min) {
    const v2f n0 = {f[0], f[0]}, n1 = {f[1], f[1]}, n2 = {f[2], f[2]}, d0 = {f[3], f[3]};
    v2f a, rinv;
    if (SHARED == 2) {
        rinv.x = rinv.y = shared;
    } else {
        v2f b = n0 * vx;
        b = __builtin_elementwise_fma(n1, vy, b);
        b = __builtin_elementwise_fma(n2, vz, b);
        rinv.x = __builtin_amdgcn_rcpf(b.x); rinv.y = __builtin_amdgcn_rcpf(b.y);
    }
    if (SHARED == 1) {
        a.x = a.y = shared;
    } else {
        a = __builtin_elementwise_fma(-n0, px, d0);
        a = __builtin_elementwise_fma(-n1, py, a);
        a = __builtin_elementwise_fma(-n2, pz, a);
        *a_min = __builtin_fminf(__builtin_fminf(*a_min, fabsf(a.x)), fabsf(a.y));
    }
    const v2f r = a * rinv;
    const v2f hu = __builtin_elementwise_fma(r, vu, pu);
    const v2f hv = __builtin_elementwise_fma(r, vv, pv);
    const v2f c4 = {f[4], f[4]}, c5 = {f[5], f[5]}, c6 = {f[6], f[6]};
    const v2f c7 = {f[7], f[7]}, c8 = {f[8], f[8]}, c9 = {f[9], f[9]};
    const v2f c10 = {f[10], f[10]}, c11 = {f[11], f[11]}, c12 = {f[12], f[12]};
    const v2f s = __builtin_elementwise_fma(c4, hu, __builtin_elementwise_fma(c5, hv, c6));
    const v2f t = __builtin_elementwise_fma(c7, hu, __builtin_elementwise_fma(c8, hv, c9));
    const v2f u = __builtin_elementwise_fma(c10, hu, __builtin_elementwise_fma(c11, hv, c12));
    float m0 = __builtin_fminf(__builtin_fminf(s.x, t.x), u.x);
    float m1 = __builtin_fminf(__builtin_fminf(s.y, t.y), u.y);
    if (ROBUST) {
        const float gm = f[13 + MODE];
        m0 = __builtin_fmaf(gm, fabsf(rinv.x), m0);
        m1 = __builtin_fmaf(gm, fabsf(rinv.y), m1);
    }
    *z0 = m0;
    *z1 = m1;
}

// The ray-parameter term of the test for two rays: false only when r - r_lo < 0 for certain
// (rare path; recomputes r exactly as filter_z2 does).
template <int SHARED>
__device__ __forceinline__ void filter_param2(const v16f f, v2f px, v2f py, v2f pz, v2f vx, v2f vy, v2f vz,
                                              const FilterQuery Q, const float shared, bool* k0, bool* k1) {
    const v2f n0 = {f[0], f[0]}, n1 = {f[1], f[1]}, n2 = {f[2], f[2]}, d0 = {f[3], f[3]};
    v2f a, rinv;
    if (SHARED == 2) {
        rinv.x = rinv.y = shared;
    } else {
        v2f b = n0 * vx;
        b = __builtin_elementwise_fma(n1, vy, b);
        b = __builtin_elementwise_fma(n2, vz, b);
        rinv.x = __builtin_amdgcn_rcpf(b.x); rinv.y = __builtin_amdgcn_rcpf(b.y);
    }
    if (SHARED == 1) {
        a.x = a.y = shared;
    } else {
        a = __builtin_elementwise_fma(-n0, px, d0);
        a = __builtin_elementwise_fma(-n1, py, a);
        a = __builtin_elementwise_fma(-n2, pz, a);
    }
    const v2f r = a * rinv;
    *k0 = !(r.x - Q.r_lo < 0.0f);
    *k1 = !(r.y - Q.r_lo < 0.0f);
}

// The in-plane-origin term of the test for two rays (rare path; recomputes a and b).
template <bool ROBUST, int SHARED>
__device__ __forceinline__ void filter_in_plane2(const v16f f, v2f px, v2f py, v2f pz, v2f vx, v2f vy, v2f vz,
                                                 const FilterQuery Q, const float shared, bool* k0, bool* k1) {
    const v2f n0 = {f[0], f[0]}, n1 = {f[1], f[1]}, n2 = {f[2], f[2]}, d0 = {f[3], f[3]};
    v2f a;
    if (SHARED == 1) {
        a.x = a.y = shared;
    } else {
        a = __builtin_elementwise_fma(-n0, px, d0);
        a = __builtin_elementwise_fma(-n1, py, a);
        a = __builtin_elementwise_fma(-n2, pz, a);
    }
    // The branch is the reference's only for a grazing ray, |b| < eps (raytracer.cpp:319), and b_hi lies above eps by
    // more than the FP32 b can be off: a ray that is not grazing goes by the other terms alone.  (Without this every
    // ray that leaves a surface lists the triangle it starts on and that triangle's coplanar neighbours: two thirds
    // of all listed pairs of the benchmark frame.)
    bool g0, g1;
    if (SHARED == 2) {
        g0 = g1 = !(fabsf(shared) * Q.b_hi < 1.0f);                 // |b| <= b_hi, or NaN
    } else {
        v2f b = n0 * vx;
        b = __builtin_elementwise_fma(n1, vy, b);
        b = __builtin_elementwise_fma(n2, vz, b);
        g0 = fabsf(b.x) <= Q.b_hi; g1 = fabsf(b.y) <= Q.b_hi;
    }
    *k0 = (fabsf(a.x) <= Q.a_zero) & g0;
    *k1 = (fabsf(a.y) <= Q.a_zero) & g1;
}

template <int RPL> struct FilterLane {     // ray slots 2q and 2q+1 share a float2
    v2f px[RPL / 2], py[RPL / 2], pz[RPL / 2], vx[RPL / 2], vy[RPL / 2], vz[RPL / 2];
    v2f hx[RPL / 2], hy[RPL / 2], hz[RPL / 2];   // unit directions (only filled for tasks that use a cone test)
    unsigned int on;     // bit r: candidates of slot r count (the slot holds a ray the filter may judge)
};

template <int AXIS> __device__ __forceinline__ v2f pick_u(v2f x, v2f y, v2f z) { return AXIS == 0 ? y : (AXIS == 1 ? z : x); }
template <int AXIS> __device__ __forceinline__ v2f pick_v(v2f x, v2f y, v2f z) { return AXIS == 0 ? z : (AXIS == 1 ? x : y); }

// Reserves `total` consecutive entries of the pair list for the wave: one atomic per wave.
// The counter is 64 bits wide and keeps counting past pair_cap, so a full list stays full.
__device__ __forceinline__ unsigned long long wave_append(unsigned long long* counter, unsigned int total, int lane) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)total);
    const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)base, 0);
    const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(base >> 32), 0);
    return ((unsigned long long)hi << 32) | lo;
}

// Appends the lane's staged candidates (LDS) to the pair list.
template <int RPL>
__device__ __forceinline__ int flush_candidates(const TraceArgs& A, const unsigned short* my_cand, int n_cand, int cbeg,
                                                unsigned int ray0, unsigned int n_rays, int lane) {
    constexpr int SLOT_SHIFT = RT_SLOT_SHIFT(RPL);
    // The list is written round by round: round k holds the k-th candidate of every lane that has one.
    // Neighbouring lanes hold neighbouring rays, which tend to keep the same triangles in the same order,
    // so neighbouring entries of the list name the same triangle and consecutive rays: the re-test kernel,
    // one thread per entry, then fetches one triangle record and a run of consecutive rays per few lanes
    // instead of one record and one ray from anywhere per lane.
    int total = n_cand, most = n_cand;
    for (int off = 32; off > 0; off >>= 1) { total += __shfl_xor(total, off); most = max(most, __shfl_xor(most, off)); }
    // (which sub-list: any spread will do; this one sends a task's flushes to different lists and neighbouring tasks' too)
    // (ray0 is the lane's first ray: lane 0's value for the whole wave)
    const unsigned int list = (((unsigned int)__builtin_amdgcn_readfirstlane((int)ray0) >> 6) * 5u + ((unsigned int)cbeg >> 6) * 3u + blockIdx.x) & (A.pair_lists - 1u);
    unsigned long long* const pairs = A.pairs + (size_t)list * A.pair_seg;
    const unsigned long long base = wave_append(A.pair_count + 16 * list, (unsigned int)total, lane);
    if (base + (unsigned int)total > A.pair_seg) {
        // Sub-list full: every ray of this lane that has a candidate gets the full scan instead.
        // Entries that still fit are written all the same (they are valid pairs), so the list
        // has no unwritten slot below its fill (capped at pair_seg).
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            const unsigned int ray = ray0 + r * RT_WAVE;
            if (ray < n_rays && n_cand > 0) { A.ray_flags[ray] = 1; *A.any_flagged = 1u; }
        }
    }
    unsigned long long at = base;
    for (int k = 0; k < most; k++) {
        const bool have = k < n_cand;
        const unsigned long long holders = __ballot(have);
        const unsigned long long mine = at + (unsigned int)__popcll(holders & ((1ull << lane) - 1ull));
        if (have && mine < A.pair_seg) {
            const unsigned int e = my_cand[k * RT_BLOCK];
            const unsigned int r = e >> SLOT_SHIFT;
            const unsigned int slot = (unsigned int)cbeg + (e & ((1u << SLOT_SHIFT) - 1u));
            pairs[mine] = ((unsigned long long)(ray0 + r * RT_WAVE) << 32) | slot;
        }
        at += (unsigned int)__popcll(holders);
    }
    return total;
}

// Filter loop over the regular triangles [abeg, aend) whose normals share the dominant axis AXIS.
typedef __attribute__((address_space(4))) const float cfloat;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const v4f cv4f;
typedef float v16u __attribute__((ext_vector_type(16), aligned(16)));      // four consecutive 16-byte records
typedef __attribute__((address_space(4))) const v16u cv16u;
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(4))) const v4u cv4u;

// PRE = 1: before the triangle test proper, a test of the triangle's bounding sphere that costs
// three multiply-adds per ray.  All rays of the task start at one point E; a hit the reference
// can accept lies inside the sphere (its radius carries the same slack as the LBVH boxes), i.e.
// inside the cone from E around the sphere, so a triangle none of the lane's unit directions
// points into (h . axis < cos) cannot be hit by any of them and the whole lane skips it.  The
// in-plane-origin case (raytracer.cpp:319-325) ignores where the triangle is: E in the plane is a
// property of the triangle alone here (the shared constant a), checked on the scalar unit.
//
// PRE = 2: shadow rays towards a point light.  Every ray lies on a line through the light, so
// the same test works with the light as the apex and both nappes of the cone (the half-line
// runs from the hit point through the light and on, raytracer.cpp:551-558): |v . axis| >= cos.
//
// PRE = 3: shadow rays towards a directional light are parallel: in the plane perpendicular
// to the light's direction a ray is a point and the sphere a disc.  |p - c|^2 <= rho^2 is
// evaluated as p.p - 2 p.c <= rho^2 - c.c (two multiply-adds per ray; the rounding of the
// expanded form is part of the radius).
//
// PRE = 2, 3 and the in-plane-origin case: it needs the ray's origin in the triangle's plane
// and the ray nearly parallel to that plane.  Both have a necessary condition that involves
// only the triangle's plane and what the task's rays share: the plane passes through the box
// around the task's origins, and (point light: b * dist = (d0 - n.light) + a) the light lies
// within b_hi * (largest distance to the light) of the plane, or (directional light) |n . dir|
// <= b_hi.  Once per 64 triangles every lane evaluates that for one of them and a ballot turns
// the answers into a scalar mask; triangles it names take the triangle test whatever the cone says.
//
// PRE = 1, 2, 3, one level up: the task's 512 rays are neighbours on the film (a patch of
// 32 x 16 pixels in the queue's tile order, see queue_pixel), so one cone (one disc) bounds them
// all, and most triangles are out of reach of the whole bundle.  Once per 64 triangles every lane
// asks that for one triangle -- cones meet when the angle between their axes is at most the sum
// of their half-angles, discs when their centres are closer than the sum of their radii -- and a
// ballot turns the answers into a scalar mask; the scan then visits only batches of four with a
// triangle the bundle can reach or the plane guard names.  Where a task's rays are a strip (the
// last rows of a pass) they also fan out in (nearly) one plane through the shared point; a sphere
// whose axis leaves that plane by more than the two angles together is out of reach as well.
//
// PRE = 4: rays with nothing in common (reflection rays).  Squared distance of the sphere's
// centre from the ray's line, |C - P|^2 - ((C - P).V)^2 / |V|^2 <= rho^2, expanded so that the
// per-ray parts (P.V, |P|^2, 1 / |V|^2) are computed once per task: nine multiply-adds per ray,
// no reciprocal; the rounding of the expanded form is part of the radius.  The in-plane-origin
// case is asked per pair here: a = d0 - n.P from a compact copy of the planes (three more).
// One level up: a bundle per pair of ray slots (128 consecutive rays of the queue) -- a ball of
// origins (m, ro) and a cone of directions.  A line through P stays within |P - m| of the parallel
// line through m, so the bundle reaches the sphere (C, rho) only if a line through m inside the
// cone of directions reaches (C, rho + ro); the in-plane-origin case needs the plane to pass
// through the ball.  A pair whose bundle reaches none of a batch's triangles sits the batch out.
struct PlaneGuard {
    const float* planes;                 // PRE 4: (n, d0) per triangle, 4 floats each
    const float* records;                // the filter records as plain memory (16 floats per triangle)
    float lx, ly, lz;                    // point light: position; directional light: unit direction
    float limit;                         // point light: b_hi * reach + 2 a_zero; directional light: b_hi (1 + 1e-3)
    float cx, cy, cz, hx, hy, hz;        // box around the task's ray origins (centre, half extents)
    float slack;                         // a_zero
    int point;
    // What bounds all rays of the task together (PRE 1, 2: a cone of directions -- unit axis, cosine and
    // sine of its half-angle; PRE 3: a disc of origins in the light's plane -- centre in (bx, by), radius
    // in bcos).  cull = 0: no such bound (directions too far apart, or not finite).
    const float* bounds;                 // the table `cone` points to, as plain memory
    const float* cl_bounds;              // the clusters' table of the same kind (PRE 4: their spheres)
    const float* cl_normals;             // PRE 4: the clusters' normal cones
    const int32_t* graze_slots;          // PRE 2, 3: see TraceArgs
    const int32_t* graze_start;          // ... the light's row
    float bx, by, bz, bcos, bsin;
    int cull;
    // PRE 1, 2: rays of one scanline fan out in (nearly) one plane through the shared point: unit
    // normal of that plane (zero: no such plane), sine and cosine of the largest angle a ray makes with it
    float fx, fy, fz, fsin, fcos;
    // PRE 4: two bundles per pair of ray slots (128 consecutive rays of the queue, split in two by
    // direction when they do not run together), bundle 2q + c of pair q: the origins -- centre, radius about
    // the bundle's axis line (see filter_range), radius of the ball around the centre -- and the cone of
    // directions (unit axis, cosine and sine of the half-angle); qgraze: largest |n . axis-direction| at which a
    // plane with unit normal n still holds a direction the reference calls grazing (raytracer.cpp:319).
    // on = 1: bounds hold; 0: no bound (everything is in reach); -1: the bundle has no ray
    // The eleven numbers of a bundle wait in LDS (qsh: [bundle][12] floats -- centre, the two radii, axis, cosine,
    // sine, grazing bound) and are fetched into vector registers for the judging of a chunk: in scalar
    // registers they would push the batch records of the scan loop out (spilled to lanes of a vector register
    // right behind their load, which serialises the prefetch).
    const __attribute__((address_space(3))) float* qsh;
    int qon[8];
};

// Square root for the bounding questions of the judging stage: the hardware's v_sqrt_f32 (1 ulp), without the
// refinement steps of sqrtf -- every use sits next to a relative slack of 1e-5 or more.
__device__ __forceinline__ float judge_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// PLANE = false: an instance for launches that do not ask the in-plane-origin question (TraceArgs::plane_guard == 0)
// without the machinery that names triangles for it next to the bounding tests -- the plane records beside the bounding
// records of a batch, the plane distances of the ray pairs, the clusters and graze-list entries named for it: a third
// of the scalar registers of the reflection instance, and its scratch.
template <int MODE, int RPL, int AXIS, bool ROBUST, int SHARED, int PRE, bool WORK, bool PLANE>
__device__ __forceinline__ void filter_range(const cv16f* filt, const cfloat* shared, const cv4f* cone,
                                             const PlaneGuard& guard, const TraceArgs& A,
                                             const FilterLane<RPL>& L, const FilterQuery fp, int abeg, int aend,
                                             unsigned short* my_cand, unsigned int ray0, unsigned int n_rays, int lane,
                                             WorkCount& wk) {
    constexpr int SLOT_SHIFT = RT_SLOT_SHIFT(RPL);
    // (with a bounding test: at most 64 groups of 64 triangles, one lane per group's masks)
    constexpr int CHUNK = (PRE && SLOT_SHIFT > 12) ? 4096 : (1 << SLOT_SHIFT);
    v2f pu[RPL / 2], pv[RPL / 2], vu[RPL / 2], vv[RPL / 2];
#pragma unroll
    for (int q = 0; q < RPL / 2; q++) {
        pu[q] = pick_u<AXIS>(L.px[q], L.py[q], L.pz[q]); pv[q] = pick_v<AXIS>(L.px[q], L.py[q], L.pz[q]);
        vu[q] = pick_u<AXIS>(L.vx[q], L.vy[q], L.vz[q]); vv[q] = pick_v<AXIS>(L.vx[q], L.vy[q], L.vz[q]);
    }
    int n_cand = 0;
    int cbeg = abeg;
    // The FP32 test of triangle j (record `cur`, shared constant `sh_cur`) for the lanes that ask for it.
    auto test_triangle = [&](const int j, const v16f cur, const float sh_cur) {
        float z[RPL], a_min = INFINITY;
        int z_top = INT_MIN;
#pragma unroll
        for (int q = 0; q < RPL / 2; q++) {
            filter_z2<MODE, ROBUST, SHARED>(cur, L.px[q], L.py[q], L.pz[q], L.vx[q], L.vy[q], L.vz[q], pu[q], pv[q],
                                            vu[q], vv[q], fp, sh_cur, &z[2 * q], &z[2 * q + 1], &a_min);
            z_top = max(max(z_top, __float_as_int(z[2 * q])), __float_as_int(z[2 * q + 1]));
        }
        if (SHARED == 1) a_min = fabsf(sh_cur);
        const bool maybe_in_plane = a_min <= fp.a_zero;
        // "some z is >= 0 or NaN", asked of the bit patterns as signed integers: a float with
        // a clear sign bit is a non-negative integer, a NaN with the sign bit set lies above
        // the pattern of -inf (0xFF800000); every negative number (and -0) lies at or below it
        if ((z_top > (int)0xFF800000) | maybe_in_plane) {      // rarely true for any lane of the wave
            // (The empty asm statements pin this block's arithmetic inside the branch: the
            // compiler would otherwise hoist it into the loop body that every triangle pays for.)
            bool keep[RPL];
#pragma unroll
            for (int q = 0; q < RPL / 2; q++) {
                float za = z[2 * q], zb = z[2 * q + 1];
                asm volatile("" : "+v"(za), "+v"(zb));
                bool k0, k1;
                v2f pxq = L.px[q], vxq = L.vx[q];
                asm volatile("" : "+v"(pxq), "+v"(vxq));
                filter_param2<SHARED>(cur, pxq, L.py[q], L.pz[q], vxq, L.vy[q], L.vz[q], fp, sh_cur, &k0, &k1);
                keep[2 * q] = !(za < 0.0f) & k0;
                keep[2 * q + 1] = !(zb < 0.0f) & k1;
            }
            if (maybe_in_plane) {
#pragma unroll
                for (int q = 0; q < RPL / 2; q++) {
                    bool k0, k1;
                    v2f pxq = L.px[q], vxq = L.vx[q];
                    asm volatile("" : "+v"(pxq), "+v"(vxq));
                    filter_in_plane2<ROBUST, SHARED>(cur, pxq, L.py[q], L.pz[q], vxq, L.vy[q], L.vz[q], fp, sh_cur, &k0, &k1);
                    keep[2 * q] |= k0; keep[2 * q + 1] |= k1;
                }
            }
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                if (keep[r] && ((L.on >> r) & 1u)) {
                    my_cand[n_cand * RT_BLOCK] = (unsigned short)((j - cbeg) | (r << SLOT_SHIFT));
                    n_cand++;
                }
            }
        }
    };
    v16f cones_cur = {}, cones_nxt = {}, planes_cur = {}, planes_nxt = {};
    int nxt_index = -1;                       // the batch whose records cones_nxt / planes_nxt hold (-1: none)
    unsigned long long guard_mask = 0, reach_mask = ~0ull;
    unsigned long long pair_mask[RPL / 2];    // PRE 4: per pair of ray slots, the triangles of the group its bundle can reach
#pragma unroll
    for (int q = 0; q < RPL / 2; q++) pair_mask[q] = ~0ull;
    int guard_base = 0;
    // Per group of 64 triangles every lane looks at one triangle (bounding record, plane) and the
    // answers become lane masks.  A chunk's groups are judged before its triangles are scanned,
    // RT_GROUPS_AHEAD groups per round with all their records requested together (one memory
    // latency per round, not per group); group g's masks wait in lane g of a register pair, and
    // one bit per group says whether any triangle of it is wanted at all.
    constexpr int NM = (PRE == 4) ? RPL / 2 : (PLANE ? 2 : 1);        // (masks kept per group: PRE 4 one per pair of ray slots; else reach, and named)
    int kept_lo[NM], kept_hi[NM];
#pragma unroll
    for (int k = 0; k < NM; k++) { kept_lo[k] = 0; kept_hi[k] = 0; }
    unsigned long long live_groups = 0ull;
    // Can bundle sb (PRE 4) reach the sphere (centre, rho)?  The ray's LINE against the sphere widened by the
    // origins' distance from the bundle's axis line, and the ray's HALF-line (the reference accepts no hit
    // behind the origin, raytracer.cpp:325) against the sphere widened by the whole ball of origins.
    struct BundleRegs { float m[5], a[5], graze; };
    BundleRegs QB[PRE == 4 ? RPL : 1];
    auto bundle_reach = [&](const int sb, const float cx, const float cy, const float cz, const float rho) -> bool {
        const float dx = cx - QB[sb].m[0], dy = cy - QB[sb].m[1], dz = cz - QB[sb].m[2];
        const float dist2 = dx * dx + dy * dy + dz * dz;
        const float dist = judge_sqrt(dist2);
        const float fwd = dx * QB[sb].a[0] + dy * QB[sb].a[1] + dz * QB[sb].a[2];
        const float cb = QB[sb].a[3], sbn = QB[sb].a[4];
        const float R1 = (rho + QB[sb].m[3]) * (1.0f + 1e-5f);
        const float cs1 = judge_sqrt(fmaxf(dist2 - R1 * R1, 0.0f));
        const bool line_reach = !(R1 < 0.999f * dist) | !(fabsf(fwd) < cs1 * cb - R1 * sbn - 1e-5f * dist);
        const float R2 = (rho + QB[sb].m[4]) * (1.0f + 1e-5f);
        const float cs2 = judge_sqrt(fmaxf(dist2 - R2 * R2, 0.0f));
        const bool half_reach = !(R2 < 0.999f * dist) | !(fwd < cs2 * cb - R2 * sbn - 1e-5f * dist);
        return line_reach & half_reach;
    };
    // Can the task's cone of directions from the shared point (PRE 1, 2) reach the cone `rec` (unit axis, cosine)
    // around a sphere?  Cones meet when the angle between their axes is at most the sum of their half-angles;
    // and the sphere's axis may leave the fan's plane by at most the two angles together.
    const bool guard_only = PLANE && A.in_plane_only != 0;          // (a kernel argument: wave-uniform)
    auto cone_reach = [&](const float4 rec) -> bool {
        // (in-plane-origin question only, camera rays: the host marked the records of triangles -- and of clusters with
        // such a member -- in whose plane the eye may lie with "every direction", cos = -2)
        if (PRE == 1 && guard_only) return rec.w < -1.0f;
        float along = rec.x * guard.bx + rec.y * guard.by + rec.z * guard.bz;
        if (PRE == 2) along = fabsf(along);
        const float ct = rec.w, st = judge_sqrt(fmaxf(1.0f - ct * ct, 0.0f));
        const bool in_reach = (ct < -1.0f) | !(along < ct * guard.bcos - st * guard.bsin - 1e-5f);
        const float aside = fabsf(rec.x * guard.fx + rec.y * guard.fy + rec.z * guard.fz);
        const bool in_fan = (ct < -1.0f) | !(ct * guard.fcos - st * guard.fsin > 0.0f) |
                            !(aside > st * guard.fcos + ct * guard.fsin + 1e-5f);
        return in_reach & in_fan;
    };
    // PRE 3: the task's disc of origins in the light's plane against the disc `rec` = (-2 cu, -2 cv, rho^2 - c.c)
    auto disc_reach = [&](const float4 rec) -> bool {
        const float cu = -0.5f * rec.x, cv = -0.5f * rec.y;
        const float rho = judge_sqrt(fmaxf(rec.z + cu * cu + cv * cv, 0.0f));      // (radius^2 = k + c.c; inf stays inf)
        const float du = cu - guard.bx, dv = cv - guard.by, sum = rho + guard.bcos;
        return !(du * du + dv * dv > sum * sum * (1.0f + 1e-5f));
    };
    // `reach_possible`: the group's cluster passed the test one level up (wave-uniform); otherwise only the
    // in-plane-origin question is asked (PRE 4; the other kinds skip such groups altogether).
    auto group_masks = [&](const int j, const int cend, const float4 rec, const float4 pl, const bool reach_possible,
                           unsigned long long* out) {
        if (PRE == 4) {
            // rec = centre, rho^2 - |C|^2; pl = (n, d0)
            float rho = 0.0f;
            if (reach_possible) rho = judge_sqrt(fmaxf(rec.w + rec.x * rec.x + rec.y * rec.y + rec.z * rec.z, 0.0f));
#pragma unroll
            for (int q = 0; q < RPL / 2; q++) {
                bool live = false;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int sb = 2 * q + c;
                    if (guard.qon[sb] < 0) continue;                     // (wave-uniform: the bundle holds no ray)
                    if (guard.qon[sb] == 0) { live = true; continue; }
                    if constexpr (PLANE) {
                        // in-plane origin: the plane passes through the ball of origins and holds a grazing direction of the cone
                        const float off = fabsf(pl.x * QB[sb].m[0] + pl.y * QB[sb].m[1] + pl.z * QB[sb].m[2] - pl.w);
                        const float g = fabsf(pl.x * QB[sb].a[0] + pl.y * QB[sb].a[1] + pl.z * QB[sb].a[2]);
                        // the smallest |n . v| over the cone is g cos - sqrt(1 - g^2) sin (<= 0: some v lies in the plane);
                        // "at most the grazing bound" without the root: lhs = g cos - bound <= sqrt(1 - g^2) sin
                        const float lhs = g * QB[sb].a[3] - QB[sb].graze;
                        const bool grazing = !(lhs > 0.0f) | !(lhs * lhs > (1.0f - g * g) * QB[sb].a[4] * QB[sb].a[4] * (1.0f + 1e-5f));
                        const bool is_named = !(off > QB[sb].m[4] + 2.0f * guard.slack) & grazing;
                        live |= is_named;
                    }
                    if (reach_possible) live |= bundle_reach(sb, rec.x, rec.y, rec.z, rho);
                }
                live &= j < cend;
                out[q] = __ballot(live);
            }
        } else {
            bool in_reach = j < cend;
            if ((guard.cull || (PRE == 1 && guard_only)) && j < cend) in_reach = (PRE == 3) ? disc_reach(rec) : cone_reach(rec);
            if constexpr (PLANE) out[1] = 0ull;          // (the plane guard's answers are added afterwards, see below)
            out[0] = __ballot(in_reach);
        }
    };
    // One level up: can a ray of the task reach anything in cluster `cl` (64 consecutive slots)?  Its record is
    // the bound of the same kind around a sphere that holds the bounding spheres of its triangles.  The
    // in-plane-origin case ignores where a triangle is and is asked separately:
    //   camera rays: the cluster's record says "every direction" when the eye lies in a member's plane;
    //   shadow rays: only triangles on the light's graze list can take that branch; the entries of the list
    //     that fall into the chunk are asked the plane guard's question one by one (below);
    //   reflection rays: per triangle, for every group (group_masks).
    auto cluster_reach = [&](const int cl) -> bool {
        const float4 rec = *(const float4*)(guard.cl_bounds + 4 * (size_t)cl);
        if (PRE == 4) {
            bool live = false;
#pragma unroll
            for (int sb = 0; sb < RPL; sb++) {
                if (guard.qon[sb] < 0) continue;
                if (guard.qon[sb] == 0) { live = true; continue; }
                live |= bundle_reach(sb, rec.x, rec.y, rec.z, rec.w);
            }
            return live;
        } else {
            if (PRE == 1 && guard_only) return rec.w < -1.0f;
            if (!guard.cull) return true;
            return (PRE == 3) ? disc_reach(rec) : cone_reach(rec);
        }
    };
    // PRE 4, one level up for the in-plane-origin question: can ANY member of cluster `cl` be named by some bundle?
    // The members' unit normals lie within an angle alpha of the cluster's mean normal N (cl_normals: N, sin alpha),
    // their planes pass through the cluster's sphere (C, rho).  For a bundle (centre m, ball radius R, cone axis a,
    // half-angle beta, grazing bound) a member can only be named when
    //   its plane can pass through the ball:  |n . (m - C)| - rho <= R, and |n . (m - C)| >= |m - C| cos(phi + alpha)
    //     with phi the angle between m - C and +-N (while phi + alpha < 90 degrees), and
    //   its plane can hold a grazing direction of the cone:  cos(theta + beta) <= bound with theta the angle between n
    //     and +-a, and theta <= theta_N + alpha.
    auto cluster_named = [&](const int cl) -> bool {
        if (PRE != 4) return false;
        const float4 sp = *(const float4*)(guard.cl_bounds + 4 * (size_t)cl);
        const float4 nc = *(const float4*)(guard.cl_normals + 4 * (size_t)cl);
        const float sa = nc.w, ca = judge_sqrt(fmaxf(1.0f - sa * sa, 0.0f));
        bool named = false;
#pragma unroll
        for (int sb = 0; sb < (PRE == 4 ? RPL : 1); sb++) {
            if (guard.qon[sb] < 0) continue;
            if (guard.qon[sb] == 0) { named = true; continue; }
            const float c1 = fabsf(nc.x * QB[sb].a[0] + nc.y * QB[sb].a[1] + nc.z * QB[sb].a[2]);
            const float s1 = judge_sqrt(fmaxf(1.0f - c1 * c1, 0.0f));
            const float cab = ca * QB[sb].a[3] - sa * QB[sb].a[4], sab = sa * QB[sb].a[3] + ca * QB[sb].a[4];   // cos, sin of alpha + beta
            const bool can_graze = !(cab > 0.0f) | !(c1 * cab - s1 * sab > QB[sb].graze + 1e-4f);
            const float dx = QB[sb].m[0] - sp.x, dy = QB[sb].m[1] - sp.y, dz = QB[sb].m[2] - sp.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const float along = fabsf(nc.x * dx + nc.y * dy + nc.z * dz);
            const float least = along * ca - judge_sqrt(fmaxf(d2 - along * along, 0.0f)) * sa;                 // <= |n . (m - C)| for every member
            const bool can_pass = !(least - sp.w > (QB[sb].m[4] + 2.0f * guard.slack) * (1.0f + 1e-4f) + 1e-5f * judge_sqrt(d2));
            named |= can_graze & can_pass;
        }
        return named;
    };
    for (cbeg = abeg; cbeg < aend; cbeg += CHUNK) {
        const int cend = min(cbeg + CHUNK, aend);
        n_cand = 0;
#ifdef RT_PHASE_CLOCK
        const unsigned int clk_judge = RT_CLOCK();
        unsigned int clk_flush = 0;
#endif
        if (PRE) {
            live_groups = 0ull;
            const int n_grp = (cend - cbeg + 63) >> 6;
            if (PRE == 4) {
                const volatile __attribute__((address_space(3))) float* q = guard.qsh;   // (volatile: fetched here, for this chunk's judging only)
#pragma unroll
                for (int sb = 0; sb < RPL; sb++) {
#pragma unroll
                    for (int k = 0; k < 5; k++) { QB[sb].m[k] = q[12 * sb + k]; QB[sb].a[k] = q[12 * sb + 5 + k]; }
                    QB[sb].graze = q[12 * sb + 10];
                }
            }
            // clusters first, one per lane (a chunk holds at most 64 and starts at a multiple of 64) ...
            // (in-plane-origin question only: nothing is asked about reach; shadow rays go by the graze lists below,
            // reflection rays by cluster_named, camera rays by the records the host marked)
            const unsigned long long cl_reach = (guard_only && PRE != 1) ? 0ull : __ballot(lane < n_grp && cluster_reach((cbeg >> 6) + lane));
            if (WORK) wk.v[RT_WK_CLUSTERS] += (unsigned int)n_grp;
            // ... then the triangles, one per lane, RT_GROUPS_AHEAD groups per round: of the clusters that
            // passed, and (reflection rays) of those a member of which could be named, for that question alone
            unsigned long long todo = cl_reach;
            if (PRE == 4 && PLANE && A.plane_guard) todo |= __ballot(lane < n_grp && cluster_named((cbeg >> 6) + lane));
            while (todo != 0ull) {
                int gi[RT_GROUPS_AHEAD];
                float4 rec[RT_GROUPS_AHEAD], pl[RT_GROUPS_AHEAD];
#pragma unroll
                for (int u = 0; u < RT_GROUPS_AHEAD; u++) {
                    gi[u] = -1;
                    if (todo != 0ull) { gi[u] = (int)__builtin_ctzll(todo); todo &= todo - 1ull; }
                    const int j = cbeg + gi[u] * 64 + lane;
                    rec[u] = make_float4(0.f, 0.f, 0.f, 0.f); pl[u] = rec[u];
                    if (gi[u] >= 0 && j < cend) {
                        if (PRE != 4 || ((cl_reach >> gi[u]) & 1ull)) rec[u] = *(const float4*)(guard.bounds + 4 * (size_t)j);
                        if (PRE == 4 && PLANE && A.plane_guard) pl[u] = *(const float4*)(guard.planes + 4 * (size_t)j);
                    }
                }
#pragma unroll
                for (int u = 0; u < RT_GROUPS_AHEAD; u++) {
                    if (gi[u] >= 0) {
                        unsigned long long out[NM];
                        const bool reach_possible = (cl_reach >> gi[u]) & 1ull;
                        group_masks(cbeg + gi[u] * 64 + lane, cend, rec[u], pl[u], reach_possible, out);
                        if (WORK) {
                            wk.v[RT_WK_GROUPS]++;
                            if (reach_possible) wk.v[RT_WK_GROUP_TESTS] += (unsigned int)min(64, cend - (cbeg + gi[u] * 64)) * (PRE == 4 ? RPL : 1);
                        }
                        unsigned long long any = out[0];
                        if (PRE == 4) {
#pragma unroll
                            for (int k = 1; k < NM; k++) any |= out[k];
                        }
                        if (any != 0ull) {
                            if (WORK) wk.v[RT_WK_GROUPS_LIVE]++;
                            live_groups |= 1ull << gi[u];
#pragma unroll
                            for (int k = 0; k < NM; k++) {
                                kept_lo[k] = (lane == gi[u]) ? (int)(unsigned int)out[k] : kept_lo[k];
                                kept_hi[k] = (lane == gi[u]) ? (int)(unsigned int)(out[k] >> 32) : kept_hi[k];
                            }
                        }
                    }
                }
            }
            if constexpr ((PRE == 2 || PRE == 3) && PLANE) if (A.plane_guard) {
                // The plane guard: the light's graze-list entries inside this chunk, one per lane.  An entry is
                // named when the light is close enough to its plane for THIS task's rays (the list was cut for
                // the whole scene's) and the plane passes through the box around the task's origins.
                const int e0 = guard.graze_start[cbeg >> 6], e1 = guard.graze_start[(cend + 63) >> 6];
                for (int eb = e0; eb < e1; eb += RT_WAVE) {
                    const int e = eb + lane;
                    int j = -1;
                    bool is_named = false;
                    if (e < e1) {
                        j = guard.graze_slots[e];
                        const float4 p4 = *(const float4*)(guard.planes + 4 * (size_t)j);
                        const float across = p4.x * guard.lx + p4.y * guard.ly + p4.z * guard.lz;
                        const float off = guard.point ? fabsf(p4.w - across) : fabsf(across);
                        const float centre = fabsf(p4.x * guard.cx + p4.y * guard.cy + p4.z * guard.cz - p4.w);
                        const float reach = fabsf(p4.x) * guard.hx + fabsf(p4.y) * guard.hy + fabsf(p4.z) * guard.hz + guard.slack;
                        is_named = !(off > guard.limit) && !(centre > reach) && j >= cbeg && j < cend;
                    }
                    unsigned long long hits = __ballot(is_named);
                    while (hits != 0ull) {                        // (rare)
                        const int src = (int)__builtin_ctzll(hits);
                        hits &= hits - 1ull;
                        const int rel = __builtin_amdgcn_readlane(j, src) - cbeg;
                        const int g = rel >> 6, bit = rel & 63;
                        const int lo_bit = bit < 32 ? (int)(1u << bit) : 0, hi_bit = bit < 32 ? 0 : (int)(1u << (bit - 32));
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int had_lo = ((live_groups >> g) & 1ull) ? kept_lo[k] : 0, had_hi = ((live_groups >> g) & 1ull) ? kept_hi[k] : 0;
                            kept_lo[k] = (lane == g) ? (had_lo | lo_bit) : kept_lo[k];
                            kept_hi[k] = (lane == g) ? (had_hi | hi_bit) : kept_hi[k];
                        }
                        live_groups |= 1ull << g;
                    }
                }
            }
        }
#ifdef RT_PHASE_CLOCK
        const unsigned int clk_scan = RT_CLOCK();
        if (WORK) wk.t[2] += clk_scan - clk_judge;
#endif
        for (int j0 = cbeg; j0 < cend; j0 += RT_BATCH) {
            int jn = min(j0 + RT_BATCH, cend);
            int j_next = j0 + RT_BATCH;               // where the scan goes on after this batch
            if (PRE) {
                // Only batches with a wanted triangle are visited (the masks say which).  The cone
                // records of a batch (4 x 16 bytes) come in one scalar load, requested while the wanted
                // batch before it is evaluated; the four tests run back to back (independent chains),
                // one branch per batch asks whether any lane wants any of the four triangles, and only
                // then is a triangle record fetched (the scalar cache cannot feed 64-byte records at
                // the pace of the cone test).  The arrays are readable a batch past their end.
                // (Scalar loads return out of order, so waiting for one waits for all: the only load
                // in flight while a batch is evaluated must be the next wanted batch's.  The first
                // wanted batch of a chunk requests its own records and waits for them.)
                // (one comparison per triangle, kept as a lane mask; the four are merged on the scalar
                // unit and turned into per-lane bits only inside the branch)
                bool wants[RT_BATCH];
                unsigned int named = 0;
                if (((j0 - cbeg) & 63) == 0) {
                    int g = (j0 - cbeg) >> 6;
                    const unsigned long long rest = live_groups >> g;
                    if (rest == 0ull) {
                        // nothing wanted up to the end of the chunk
                        reach_mask = 0ull; guard_mask = 0ull;
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) pair_mask[q] = 0ull;
                        j0 = cend - RT_BATCH;
                        jn = cend;
                        j_next = cend;
                        guard_base = j0;
                    } else {
                        const int skip = __builtin_ctzll(rest);
                        g += skip;
                        j0 += 64 * skip;
                        unsigned long long got[NM];
#pragma unroll
                        for (int k = 0; k < NM; k++) {
                            got[k] = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(kept_lo[k], g) |
                                     ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(kept_hi[k], g) << 32);
                        }
                        if (PRE == 4) {
                            reach_mask = 0ull;
#pragma unroll
                            for (int q = 0; q < RPL / 2; q++) { pair_mask[q] = got[q]; reach_mask |= got[q]; }
                        } else {
                            reach_mask = got[0];
                            if constexpr (PLANE) guard_mask = got[1];
                        }
                        guard_base = j0;
                        // on to the group's first wanted batch (a live group has one)
                        j0 += __builtin_ctzll(reach_mask) & ~(RT_BATCH - 1);
                        jn = min(j0 + RT_BATCH, cend);
                        j_next = j0 + RT_BATCH;
                    }
                }
                const bool out_of_reach = ((reach_mask >> (j0 - guard_base)) & ((1ull << RT_BATCH) - 1ull)) == 0ull;
                if (!out_of_reach) {
                if (nxt_index != j0) {
                    cones_nxt = *(const cv16u*)((const cfloat*)cone + 4 * (size_t)j0);
                    if (PRE == 4 && PLANE) planes_nxt = *(const cv16u*)((const cfloat*)guard.planes + 4 * (size_t)j0);
                }
                cones_cur = cones_nxt;
                if (PRE == 4 && PLANE) planes_cur = planes_nxt;
                // the next wanted batch of this group (the first one of another group fetches its own records)
                {
                    const int off = j0 - guard_base + RT_BATCH;
                    const unsigned long long ahead = (off < 64) ? (reach_mask >> off) : 0ull;
                    int jl = -1;
                    if (ahead != 0ull) {
                        jl = j0 + RT_BATCH + (__builtin_ctzll(ahead) & ~(RT_BATCH - 1));
                        j_next = jl;
                    } else {
                        j_next = min(guard_base + 64, cend);
                    }
                    if (jl >= 0) {
                        cones_nxt = *(const cv16u*)((const cfloat*)cone + 4 * (size_t)jl);
                        if (PRE == 4 && PLANE) planes_nxt = *(const cv16u*)((const cfloat*)guard.planes + 4 * (size_t)jl);
                    }
                    nxt_index = jl;
                }
                if ((PRE == 2 || PRE == 3) && PLANE) named = (unsigned int)(guard_mask >> (j0 - guard_base)) & ((1u << RT_BATCH) - 1u);
                if (WORK) {
                    wk.v[RT_WK_BATCHES]++;
                    if (PRE != 4) wk.v[RT_WK_RAY_BOUND_TESTS] += RT_WAVE * RPL * RT_BATCH;
                }
#pragma unroll
                for (int jb = 0; jb < RT_BATCH; jb++) {
                    const float ax = cones_cur[4 * jb], ay = cones_cur[4 * jb + 1], az = cones_cur[4 * jb + 2];
                    const v2f cx = {ax, ax}, cy = {ay, ay}, cz = {az, az};
                    v2f d[RPL / 2];
                    if (PRE == 4) {
                        // (evaluated below, pair by pair)
                        wants[jb] = false;
                    } else if (PRE == 3) {
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) d[q] = __builtin_elementwise_fma(cy, L.hy[q], L.hz[q]);
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) d[q] = __builtin_elementwise_fma(cx, L.hx[q], d[q]);
                        float low = __builtin_fminf(d[0].x, d[0].y);
#pragma unroll
                        for (int q = 1; q < RPL / 2; q++) low = __builtin_fminf(__builtin_fminf(low, d[q].x), d[q].y);
                        wants[jb] = low <= az;
                    } else {
                        const v2f* hx = (PRE == 2) ? L.vx : L.hx;         // shadow-ray directions have unit length
                        const v2f* hy = (PRE == 2) ? L.vy : L.hy;
                        const v2f* hz = (PRE == 2) ? L.vz : L.hz;
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) d[q] = cx * hx[q];
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) d[q] = __builtin_elementwise_fma(cy, hy[q], d[q]);
#pragma unroll
                        for (int q = 0; q < RPL / 2; q++) d[q] = __builtin_elementwise_fma(cz, hz[q], d[q]);
                        float top;
                        if (PRE == 2) {
                            top = __builtin_fmaxf(fabsf(d[0].x), fabsf(d[0].y));
#pragma unroll
                            for (int q = 1; q < RPL / 2; q++) top = __builtin_fmaxf(__builtin_fmaxf(top, fabsf(d[q].x)), fabsf(d[q].y));
                        } else {
                            top = __builtin_fmaxf(d[0].x, d[0].y);
#pragma unroll
                            for (int q = 1; q < RPL / 2; q++) top = __builtin_fmaxf(__builtin_fmaxf(top, d[q].x), d[q].y);
                        }
                        // (cos = -2 where the shared point may lie in the triangle's plane or inside the sphere)
                        wants[jb] = top >= cones_cur[4 * jb + 3];
                    }
                }
                if (PRE == 4) {
                    // (hx, hy, hz) = (P.V, |P|^2, 1 / |V|^2); the sphere record's fourth entry is rho^2 - |C|^2.
                    // A pair of ray slots whose bundle reaches none of the four triangles sits the batch out.
                    float low[RT_BATCH], a_low[RT_BATCH];
#pragma unroll
                    for (int jb = 0; jb < RT_BATCH; jb++) { low[jb] = INFINITY; a_low[jb] = INFINITY; }
                    const v2f m2 = {-2.0f, -2.0f};
#pragma unroll
                    for (int q = 0; q < RPL / 2; q++) {
                        if (((pair_mask[q] >> (j0 - guard_base)) & ((1ull << RT_BATCH) - 1ull)) == 0ull) continue;
                        if (WORK) wk.v[RT_WK_RAY_BOUND_TESTS] += RT_WAVE * 2 * RT_BATCH;
                        v2f t[RT_BATCH], d[RT_BATCH], a[RT_BATCH];
#pragma unroll
                        for (int jb = 0; jb < RT_BATCH; jb++) {
                            const float cx1 = cones_cur[4 * jb], nx1 = planes_cur[4 * jb];
                            const v2f cx = {cx1, cx1}, n0 = {nx1, nx1}, dd = {planes_cur[4 * jb + 3], planes_cur[4 * jb + 3]};
                            t[jb] = __builtin_elementwise_fma(cx, L.vx[q], -L.hx[q]);
                            d[jb] = cx * L.px[q];
                            if (PLANE) a[jb] = __builtin_elementwise_fma(-n0, L.px[q], dd);
                        }
#pragma unroll
                        for (int jb = 0; jb < RT_BATCH; jb++) {
                            const float cy1 = cones_cur[4 * jb + 1], ny1 = planes_cur[4 * jb + 1];
                            const v2f cy = {cy1, cy1}, n1 = {ny1, ny1};
                            t[jb] = __builtin_elementwise_fma(cy, L.vy[q], t[jb]);
                            d[jb] = __builtin_elementwise_fma(cy, L.py[q], d[jb]);
                            if (PLANE) a[jb] = __builtin_elementwise_fma(-n1, L.py[q], a[jb]);
                        }
#pragma unroll
                        for (int jb = 0; jb < RT_BATCH; jb++) {
                            const float cz1 = cones_cur[4 * jb + 2], nz1 = planes_cur[4 * jb + 2];
                            const v2f cz = {cz1, cz1}, n2 = {nz1, nz1};
                            t[jb] = __builtin_elementwise_fma(cz, L.vz[q], t[jb]);
                            d[jb] = __builtin_elementwise_fma(cz, L.pz[q], d[jb]);
                            if (PLANE) a[jb] = __builtin_elementwise_fma(-n2, L.pz[q], a[jb]);
                        }
#pragma unroll
                        for (int jb = 0; jb < RT_BATCH; jb++) {
                            d[jb] = __builtin_elementwise_fma(m2, d[jb], L.hy[q]);
                            t[jb] = t[jb] * t[jb];
                            d[jb] = __builtin_elementwise_fma(-t[jb], L.hz[q], d[jb]);
                            low[jb] = __builtin_fminf(__builtin_fminf(low[jb], d[jb].x), d[jb].y);
                            if (PLANE) a_low[jb] = __builtin_fminf(__builtin_fminf(a_low[jb], fabsf(a[jb].x)), fabsf(a[jb].y));
                        }
                    }
#pragma unroll
                    for (int jb = 0; jb < RT_BATCH; jb++) wants[jb] = PLANE ? ((!guard_only & (low[jb] <= cones_cur[4 * jb + 3])) | (a_low[jb] <= fp.a_zero)) : (low[jb] <= cones_cur[4 * jb + 3]);
                }
                bool any_wanted = named != 0;
#pragma unroll
                for (int jb = 0; jb < RT_BATCH; jb++) any_wanted |= wants[jb];
                if (any_wanted) {                                 // rarely true for any lane of the wave
                    unsigned int want = named;
#pragma unroll
                    for (int jb = 0; jb < RT_BATCH; jb++) want |= wants[jb] ? (1u << jb) : 0u;
                    if (WORK) wk.v[RT_WK_BATCHES_TESTED]++;
#pragma unroll 1
                    for (int jb = 0; jb < RT_BATCH; jb++) {
                        const int j = j0 + jb;
                        if (j >= jn) break;
                        if (WORK) {
                            const unsigned long long asking = __ballot((want >> jb) & 1u);
                            if (asking) {
                                wk.v[RT_WK_TRI_TESTS] += RT_WAVE * RPL;
                                wk.v[RT_WK_TRI_WANTED] += (unsigned int)__popcll(asking) * RPL;
                            }
                        }
                        if ((want >> jb) & 1u) test_triangle(j, filt[j], SHARED ? shared[j] : 0.0f);
                    }
                }
                }
            } else {
                // The record of triangle j+1 is requested before triangle j is evaluated (the
                // array carries padding records, so j+1 is always readable).
                v16f cur = filt[j0];
                float sh_cur = SHARED ? shared[j0] : 0.0f;
                if (WORK) {
                    wk.v[RT_WK_TRI_TESTS] += (unsigned int)(jn - j0) * RT_WAVE * RPL;
                    wk.v[RT_WK_TRI_WANTED] += (unsigned int)(jn - j0) * RT_WAVE * RPL;
                }
                for (int j = j0; j < jn; j++) {
                    const v16f nxt = filt[j + 1];
                    const float sh_nxt = SHARED ? shared[j + 1] : 0.0f;
                    test_triangle(j, cur, sh_cur);
                    cur = nxt;
                    sh_cur = sh_nxt;
                }
            }
            // Flush when some lane could overflow its staging slots during the next batch,
            // and at the end of the chunk (entries are relative to the chunk base).
            const bool last = j_next >= cend;
            if (__any(n_cand > RT_CAND_CAP - RT_BATCH * RPL) || (last && __any(n_cand > 0))) {
#ifdef RT_PHASE_CLOCK
                const unsigned int clk_f0 = RT_CLOCK();
#endif
                const int listed = flush_candidates<RPL>(A, my_cand, n_cand, cbeg, ray0, n_rays, lane);
                if (WORK) wk.v[RT_WK_LISTED] += (unsigned int)listed;
                n_cand = 0;
#ifdef RT_PHASE_CLOCK
                clk_flush += RT_CLOCK() - clk_f0;
#endif
            }
            j0 = j_next - RT_BATCH;
        }
#ifdef RT_PHASE_CLOCK
        if (WORK) { wk.t[3] += RT_CLOCK() - clk_scan - clk_flush; wk.t[4] += clk_flush; }
#endif
    }
}

__device__ __forceinline__ float uniform_f(float x) {      // a value every lane of the wave holds: keep it scalar
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

template <int MODE, int RPL, bool ROBUST, int CONE, bool WORK = false, bool PLANE = true>
// The bounded instances are compiled for four waves per SIMD (128 registers; a few values spill to scratch).  Timed
// alone a scan gains 3-6 % from the fourth wave and the frame nothing while one frame is in flight (the second
// stream already fills the idle slots); with three frames in flight, where the sum of the work counts, the frame
// gains 2 % (5.58 -> 5.45 ms).  Five and six waves (with fewer rays per lane to make room) measured slower.
#ifndef RT_MIN_WAVES_BOUNDED
#define RT_MIN_WAVES_BOUNDED 4
#endif
#ifndef RT_MIN_WAVES_CAMERA
#define RT_MIN_WAVES_CAMERA RT_MIN_WAVES_BOUNDED
#endif
#ifndef RT_MIN_WAVES_SHADOW
#define RT_MIN_WAVES_SHADOW RT_MIN_WAVES_BOUNDED
#endif
#ifndef RT_MIN_WAVES_ANY
#define RT_MIN_WAVES_ANY RT_MIN_WAVES_BOUNDED
#endif
__global__ void __launch_bounds__(RT_BLOCK, (CONE == 1 ? RT_MIN_WAVES_CAMERA : CONE == 2 ? RT_MIN_WAVES_SHADOW : CONE == 4 ? RT_MIN_WAVES_ANY : RT_FILTER_MIN_WAVES))
k_filter(const DevScene* __restrict__ sc, const v16f* __restrict__ filt_global, TraceArgs A) {
    // The filter records are never written while a kernel runs: reading them through the
    // constant address space lets the compiler use scalar loads (s_load_dwordx16).
    const cv16f* filt = (const cv16f*)filt_global;
    __shared__ unsigned short cand[RT_CAND_CAP * RT_BLOCK];
    __shared__ float bundle_all[(CONE == 4) ? (RT_BLOCK / RT_WAVE) * 8 * 12 : 1];
    const int lane = threadIdx.x & (RT_WAVE - 1);
    // threadIdx.x / 64 is the same in every lane of a wave; say so, so that everything
    // derived from it (task, triangle index) stays in scalar registers.
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / RT_WAVE);
    unsigned short* my_cand = cand + wave_in_block * RT_WAVE + lane;   // entry k at my_cand[k * RT_BLOCK]
    float* bundle_sh = bundle_all + ((CONE == 4) ? wave_in_block * 8 * 12 : 0);   // this wave's bundles (see PlaneGuard::qsh)
    const unsigned int n_waves = gridDim.x * (RT_BLOCK / RT_WAVE);
    constexpr unsigned int RAYS_PER_TASK = RT_WAVE * RPL;

    const unsigned int n_lights = sc->n_lights;
    // the queue length is the same for every lane: keep it (and all loop bounds derived from it) scalar
    const unsigned int n_hits = __builtin_amdgcn_readfirstlane(*A.n_items);
    const unsigned int n_rays = (MODE == 0) ? n_hits : n_hits * n_lights;
    // Shadow rays are cut into blocks per light (a block never holds rays towards two lights).
    const unsigned int blocks_per_light = (n_hits + RAYS_PER_TASK - 1) / RAYS_PER_TASK;
    const unsigned int n_ray_blocks = (MODE == 0) ? blocks_per_light : blocks_per_light * n_lights;
    const int NR = sc->n_regular;
    if (NR == 0 || n_ray_blocks == 0) return;
    // Tasks = (block of 64*RPL rays) x (part of the triangle range), handed out through a
    // global counter.  Enough parts that every wave gets several tasks: short tasks keep the
    // chip full until the end of the launch.
    // (with the bounding tests a triangle costs a task a fraction of what loading its rays costs:
    // larger parts)
#ifndef RT_MIN_PART_BOUNDED
#define RT_MIN_PART_BOUNDED 256
#endif
    int max_split = CONE ? (NR + RT_MIN_PART_BOUNDED - 1) / RT_MIN_PART_BOUNDED : (NR + 127) / 128;
    int split = 1;
#ifndef RT_TASKS_PER_WAVE_BOUNDED
#define RT_TASKS_PER_WAVE_BOUNDED 3
#endif
#ifndef RT_TASKS_PER_WAVE_SHADOW
#define RT_TASKS_PER_WAVE_SHADOW 6
#endif
#ifndef RT_TASKS_PER_WAVE_ANY
#define RT_TASKS_PER_WAVE_ANY 6
#endif
    // tasks per wave to aim for (camera rays cost alike and want few, long tasks; the tasks of shadow and
    // reflection rays differ a lot in cost, and the launch ends with its longest one)
    unsigned int per_wave = CONE == 1 ? RT_TASKS_PER_WAVE_BOUNDED : CONE == 2 ? RT_TASKS_PER_WAVE_SHADOW
                          : CONE == 4 ? RT_TASKS_PER_WAVE_ANY : 8u;
    // A launch that shares the chip with the launches of other frames (TraceArgs::beside_others) need not keep the
    // chip full to its end -- the others fill what it leaves idle -- and every part of the triangle range pays for
    // loading the block's rays again: three tasks per wave then, of half as many waves (the host halves the grid of
    // such a launch).  Four frames in flight: 5.01 -> 4.70 ms per frame; a frame on its own would take 6.6 ms instead
    // of 6.0 with so few tasks.
#ifndef RT_TASKS_PER_WAVE_BESIDE
#define RT_TASKS_PER_WAVE_BESIDE 3
#endif
    if (CONE && A.beside_others && per_wave > RT_TASKS_PER_WAVE_BESIDE) per_wave = RT_TASKS_PER_WAVE_BESIDE;
    // A launch with few blocks of rays (a deep level, one rank's share of an N-GPU frame) gets its tasks by cutting
    // the triangle range, and every part loads the block's rays and builds their bundles again -- 40-60 % of
    // such a task.  Two tasks per wave then: 2.33 -> 2.13 ms for one rank's eighth of the 4K frame, and the
    // whole frame on one GPU gains as well (its deep levels), 7.62 -> 7.35 ms.
#ifndef RT_TASKS_PER_WAVE_SMALL
#define RT_TASKS_PER_WAVE_SMALL 2
#endif
    if (n_ray_blocks < per_wave * n_waves) {
        split = (int)((per_wave * n_waves + n_ray_blocks - 1) / n_ray_blocks);
        if (CONE && n_ray_blocks < 2u * n_waves) {
            // (only where parts are short anyway: with parts of more than 2048 slots the scan outweighs the
            // reloading, and a scene of a million triangles wants its many tasks back)
            const int fewer = (int)((RT_TASKS_PER_WAVE_SMALL * n_waves + n_ray_blocks - 1) / n_ray_blocks);
            const int keep_short = (NR + 2047) / 2048;
            split = max(fewer, min(split, keep_short));
        }
        if (split > max_split) split = max_split;
        // (a launch that cannot give every wave a task even so -- a few thousand rays, the deep
        // reflection levels -- lasts as long as its longest task: parts down to 128 triangles)
        if (CONE && (unsigned int)split * n_ray_blocks < n_waves) {
            const int finest = (NR + 127) / 128;
            split = (int)((n_waves + n_ray_blocks - 1) / n_ray_blocks);
            if (split > finest) split = finest;
        }
    }
    // (a launch that asks the in-plane-origin question alone does little per triangle: one task per block of rays,
    // cut further only while there are fewer blocks than waves)
    if (CONE && A.in_plane_only) {
        split = (int)((n_waves + n_ray_blocks - 1) / n_ray_blocks);
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    const unsigned int n_tasks = n_ray_blocks * (unsigned int)split;
    const int n_groups = sc->n_groups;

    // A wave's first task is its own number; further ones come from the counter.  A launch with
    // fewer tasks than waves (the deep reflection levels) then costs the idle waves nothing: a
    // few thousand waves queueing at one counter only to learn that there is no work set a
    // floor of ~0.1 ms under every launch.
    const unsigned int wave_id = blockIdx.x * (RT_BLOCK / RT_WAVE) + (unsigned int)wave_in_block;
    bool first_task = true;
#ifdef RT_PHASE_CLOCK
    const unsigned int clk_wave = RT_CLOCK();
    unsigned long long clk_sum[7] = {0, 0, 0, 0, 0, 0, 0};      // tasks, then the six phases: flushed once per wave
    unsigned long long clk_loop = 0;                            // the ray-loading loop alone (part of phase 2)
#endif
    for (;;) {
#ifdef RT_PHASE_CLOCK
        const unsigned int clk_t0 = RT_CLOCK();
#endif
        unsigned int task = wave_id;
        if (!first_task) {
            // (asked for here, not ahead of time: a wave that holds the number of its next task while it is still busy
            // with a long one keeps that task from the waves that are idle -- the launch ends later.  Requested at
            // the task's start, or behind the loads of the task's rays: both measured 5-9 % slower.)
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(A.task_counter, 1u);
            task = n_waves + (unsigned int)__builtin_amdgcn_readfirstlane(t);
        }
        first_task = false;
        if (task >= n_tasks) break;
        WorkCount wk = {};
        if (WORK) wk.v[RT_WK_TASKS] = 1;
#ifdef RT_PHASE_CLOCK
        const unsigned int clk_t1 = RT_CLOCK();
        wk.t[0] = clk_t1 - clk_t0;
#endif
        // integer division goes through the vector ALU; its result is still wave-uniform
        const int part = __builtin_amdgcn_readfirstlane((int)(task / n_ray_blocks));
        const unsigned int rb = task - (unsigned int)part * n_ray_blocks;
        // What the task's rays share (see filter_z2 and filter_range): camera rays their origin;
        // shadow rays towards a directional light their direction, towards a point light the
        // light's position.
        unsigned int ray_base = rb * RAYS_PER_TASK, ray_end = n_rays;
        int shared_kind = 0, cone_kind = 0, light = 0;
        const cfloat* shared = nullptr;
        const cv4f* cone = nullptr;
        if (MODE == 0) {
            if (A.shared_origin_a) { shared_kind = 1; shared = (const cfloat*)A.shared_origin_a; }
            if (CONE == 1) { cone_kind = 1; cone = (const cv4f*)A.cone_camera; }
            if (CONE == 4) { cone_kind = 4; cone = (const cv4f*)A.sphere_any; }
        } else {
            light = __builtin_amdgcn_readfirstlane((int)(rb / blocks_per_light));
            const unsigned int b = rb - (unsigned int)light * blocks_per_light;
            ray_base = (unsigned int)light * n_hits + b * RAYS_PER_TASK;
            ray_end = ((unsigned int)light + 1u) * n_hits;
            const int row = __builtin_amdgcn_readfirstlane(A.light_dir_row[light]);
            if (A.shared_dir_rinv && row >= 0) {
                shared_kind = 2;
                shared = (const cfloat*)(A.shared_dir_rinv + (size_t)row * (size_t)(NR + 1));
            }
            if (CONE == 2) {
                cone_kind = (row >= 0) ? 3 : 2;
                cone = (const cv4f*)(A.bound_lights + (size_t)light * (size_t)(NR + 8) * 4);
            }
        }
        const unsigned int ray0 = ray_base + lane;                    // slot r: ray0 + r * 64
        float reach_max = 0.0f;                                       // CONE 2, point light: largest distance to the light
        float dir_x = 0.f, dir_y = 0.f, dir_z = 1.f;                  // CONE 2, directional light: the rays' direction
        float box_lo[3] = {INFINITY, INFINITY, INFINITY}, box_hi[3] = {-INFINITY, -INFINITY, -INFINITY};   // CONE 2: the lane's origins
        bool have_rays = false;
        FilterLane<RPL> L;
#pragma unroll
        for (int k = 0; k < RPL / 2; k++) {
            const v2f zero = {0.f, 0.f}, one = {1.f, 1.f};
            L.px[k] = L.py[k] = L.pz[k] = L.vx[k] = L.vy[k] = zero; L.vz[k] = one;
            L.hx[k] = L.hy[k] = zero; L.hz[k] = one;
        }
        for (int g = 0; g < n_groups; g++) {
            // group records are the same for every lane: keep the bounds in scalar registers
            const Group& G = sc->groups[g];
            int fax[4], fcnt[3];
#pragma unroll
            for (int k = 0; k < 4; k++) fax[k] = __builtin_amdgcn_readfirstlane(G.fax[k]);
#pragma unroll
            for (int k = 0; k < 3; k++) fcnt[k] = __builtin_amdgcn_readfirstlane(G.fcnt[k]);
            // Shadow rays are tested against every group's untransformed geometry
            // (raytracer.cpp:553): for them the groups only delimit the axis ranges.
            const int ident = (MODE == 0) ? __builtin_amdgcn_readfirstlane(G.identity) : 1;
            const double* X = sc->xforms + 48 * (size_t)__builtin_amdgcn_readfirstlane(G.xf);
            if (fax[0] >= fax[3]) continue;
            const GroupFilter& GQ = G.q[MODE];
            // (the in-plane-origin question not asked, TraceArgs::plane_guard: no plane distance is "at most a_zero", and
            // every place that names a triangle for that question compares against it)
            const FilterQuery fp = {uniform_f(GQ.r_lo), A.plane_guard ? uniform_f(GQ.a_zero) : -INFINITY, uniform_f(GQ.b_hi)};
            // (shadow rays: every group carries the same two bounds, those of world-space rays)
            const float ob = uniform_f(GQ.origin_bound), vb = uniform_f(GQ.dir_len2_bound);
            // FP32 copies of the lane's rays in this group's object space
            if (MODE == 0 || !have_rays) {
                have_rays = true;
                L.on = 0;
                const bool copies = CONE != 0 && A.rays.scan != nullptr &&
                                    (MODE == 0 ? (A.scan_nearest != 0 && ident) : (A.shadow_scan != nullptr));
                // The FP32 scan copies (RayQueue::scan) when the launch has them: three loads per ray, no
                // double-precision arithmetic.  Otherwise the ray's doubles are converted here.
                const float4* SA = (const float4*)A.rays.scan;
                const float4* SB = SA + A.rays.scan_stride;
                const float* SC = (const float*)(SB + A.rays.scan_stride);
                float udx = 0.f, udy = 0.f, udz = 1.f;        // directional light: the direction every ray of the task has
                if (copies && CONE == 2 && cone_kind == 3) {
                    const d3 dv = nrm3(neg3(ld3(sc->lights[light].pos)));      // Light::lightRay (raytracer.cpp:113-117)
                    udx = (float)dv.x; udy = (float)dv.y; udz = (float)dv.z;
                }
                // A real loop (not fully unrolled): a ray's temporaries are dead before the next round's are
                // born, so this prologue does not set the kernel's register count.
                // The results are routed into the (statically indexed) register arrays by selects.
                // (two rays per round: two rays' loads in flight; four or all of them made the scans slower)
#ifndef RT_RAY_LOAD_UNROLL
#define RT_RAY_LOAD_UNROLL 2
#endif
#ifdef RT_PHASE_CLOCK
                const unsigned int clk_r0 = RT_CLOCK();
#endif
#pragma unroll RT_RAY_LOAD_UNROLL
                for (int r = 0; r < RPL; r++) {
                    // A slot past the end of the queue mirrors the queue's last ray: its candidates are
                    // dropped (bit r of L.on stays clear), so no per-slot thresholds are needed.
                    const unsigned int ray = ray0 + r * RT_WAVE;
                    const bool valid = ray < ray_end;
                    const unsigned int at = valid ? ray : ray_end - 1;
                    float fx, fy, fz, gx, gy, gz, ux = 0.f, uy = 0.f, uz = 1.f, len = 0.f;
                    float rdx = 0.f, rdy = 0.f, rdz = 1.f;      // CONE 2, directional light: the ray's direction
                    if (copies) {
                        if (MODE == 0) {
                            const float4 a = SA[at], b = SB[at];
                            fx = a.x; fy = a.y; fz = a.z; gx = b.x; gy = b.y; gz = b.z;
                            if (CONE == 1 || CONE == 4) { ux = a.w; uy = b.w; uz = SC[at]; }
                        } else {
                            const unsigned int hit = at - (unsigned int)light * n_hits;     // (a block never holds rays towards two lights)
                            const float4 a = SA[hit];
                            const float4 rec = A.shadow_scan[(size_t)light * A.shadow_stride + hit];
                            fx = a.x; fy = a.y; fz = a.z;
                            gx = rec.x; gy = rec.y; gz = rec.z; len = rec.w;
                            if (cone_kind == 3) { ux = rec.x; uy = rec.y; uz = rec.z; gx = udx; gy = udy; gz = udz; }
                        }
                        rdx = gx; rdy = gy; rdz = gz;
                    } else {
                        d3 pos, vec;
                        int source;
                        load_world_ray<MODE>(A, sc, at, n_hits, &pos, &vec, &source);
                        to_object_space(X, ident, &pos, &vec);
                        fx = (float)pos.x; fy = (float)pos.y; fz = (float)pos.z;
                        gx = (float)vec.x; gy = (float)vec.y; gz = (float)vec.z;
                        rdx = gx; rdy = gy; rdz = gz;
                        if (CONE == 1) scan_h_camera(vec.x, vec.y, vec.z, &ux, &uy, &uz);
                        // (directional light: the origin in the light's plane -- coordinates about the scene's centre -- and its square)
                        if (CONE == 2 && cone_kind == 3) scan_plane_point(A.light_frame + 8 * (size_t)light, pos.x, pos.y, pos.z, &ux, &uy, &uz);
                        if (CONE == 2 && cone_kind == 2) len = scan_len_point(sc->lights[light].pos, pos.x, pos.y, pos.z);
                    }
                    // Outside the bounds the margins were derived for (or NaN): the ray takes the
                    // full reference-precision scan of k_brute instead of the filter.
                    const bool trusted = fabsf(fx) <= ob && fabsf(fy) <= ob && fabsf(fz) <= ob &&
                                         (gx * gx + gy * gy + gz * gz) <= vb;
                    if (CONE == 2 && cone_kind == 2 && valid) reach_max = fmaxf(reach_max, len);   // (a NaN drops out; such a ray is flagged below)
                    if (CONE == 2 && valid && trusted) {
                        box_lo[0] = fminf(box_lo[0], fx); box_lo[1] = fminf(box_lo[1], fy); box_lo[2] = fminf(box_lo[2], fz);
                        box_hi[0] = fmaxf(box_hi[0], fx); box_hi[1] = fmaxf(box_hi[1], fy); box_hi[2] = fmaxf(box_hi[2], fz);
                    }
                    if (!trusted) {
                        if (valid && part == 0) { A.ray_flags[ray] = 1; *A.any_flagged = 1u; }
                        // (camera rays keep the origin they all share)
                        if (CONE != 1) { fx = 0.f; fy = 0.f; fz = 0.f; }
                        gx = 0.f; gy = 0.f; gz = 1.f;
                    }
                    // (per-ray parts of the expanded distance test, from the rounded values the test multiplies;
                    // the copies hold them for the ray's own values)
                    if (CONE == 4 && (!copies || !trusted)) scan_h_any(fx, fy, fz, gx, gy, gz, &ux, &uy, &uz);
                    if (CONE == 2 && cone_kind == 3) {
                        // Every ray of the task has the light's direction (kept in scalar registers);
                        // the direction slots carry the origin in the light's plane instead.
                        if (valid || r == 0) { dir_x = rdx; dir_y = rdy; dir_z = rdz; }
                        gx = ux; gy = uy; gz = uz;
                    }
                    if (valid && trusted) L.on |= 1u << r;
#pragma unroll
                    for (int k = 0; k < RPL; k++) {
                        if (k == r) {
                            if (k & 1) { L.px[k / 2].y = fx; L.py[k / 2].y = fy; L.pz[k / 2].y = fz;
                                         L.vx[k / 2].y = gx; L.vy[k / 2].y = gy; L.vz[k / 2].y = gz; }
                            else { L.px[k / 2].x = fx; L.py[k / 2].x = fy; L.pz[k / 2].x = fz;
                                   L.vx[k / 2].x = gx; L.vy[k / 2].x = gy; L.vz[k / 2].x = gz; }
                            if (CONE == 1 || CONE == 4) {
                                if (k & 1) { L.hx[k / 2].y = ux; L.hy[k / 2].y = uy; L.hz[k / 2].y = uz; }
                                else { L.hx[k / 2].x = ux; L.hy[k / 2].x = uy; L.hz[k / 2].x = uz; }
                            }
                        }
                    }
                }
#ifdef RT_PHASE_CLOCK
                if (g == 0) { asm volatile("" :: "v"(L.px[0]), "v"(L.vz[RPL / 2 - 1])); wk.t[6] = RT_CLOCK() - clk_r0; }
#endif
            }
            PlaneGuard guard = {};
            if (CONE == 4) { guard.planes = A.plane_any; guard.cl_bounds = A.cl_sphere; guard.cl_normals = A.cl_normal; guard.qsh = (const __attribute__((address_space(3))) float*)bundle_sh; }
            if (CONE == 1) guard.cl_bounds = A.cl_camera;
            if (CONE == 2) {
                guard.cl_bounds = A.cl_lights + (size_t)light * (size_t)(A.n_clusters + 1) * 4;
                guard.graze_slots = A.graze_slots;
                guard.graze_start = A.graze_start + (size_t)light * (size_t)(A.n_clusters + 1);
                guard.planes = A.plane_any;
            }
            if (CONE == 2) {
                float m = reach_max;
                for (int off = 32; off > 0; off >>= 1) {
                    m = fmaxf(m, __shfl_xor(m, off));
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        box_lo[a] = fminf(box_lo[a], __shfl_xor(box_lo[a], off));
                        box_hi[a] = fmaxf(box_hi[a], __shfl_xor(box_hi[a], off));
                    }
                }
                const DevLight& lt = sc->lights[light];
                guard.records = (const float*)filt_global;
                guard.point = cone_kind == 2;
                if (guard.point) {
                    guard.lx = (float)lt.pos[0]; guard.ly = (float)lt.pos[1]; guard.lz = (float)lt.pos[2];
                    // (the light may lie far outside the bounds a_zero was derived for: its own rounding on top)
                    guard.limit = uniform_f(m) * fp.b_hi + 2.0f * fp.a_zero +
                                  1e-6f * (fabsf(guard.lx) + fabsf(guard.ly) + fabsf(guard.lz));
                } else {
                    const double* F = A.light_frame + 8 * (size_t)light;       // e1 x e2 = the light's direction
                    guard.lx = (float)(F[1] * F[5] - F[2] * F[4]); guard.ly = (float)(F[2] * F[3] - F[0] * F[5]);
                    guard.lz = (float)(F[0] * F[4] - F[1] * F[3]);
                    guard.limit = fp.b_hi * (1.0f + 1e-3f);
                }
                // (no trusted ray at all: lo = +inf, hi = -inf; the half extents become -inf and nothing is named)
                guard.cx = uniform_f(0.5f * (box_lo[0] + box_hi[0])); guard.hx = uniform_f(0.5f * (box_hi[0] - box_lo[0]) * (1.0f + 1e-6f));
                guard.cy = uniform_f(0.5f * (box_lo[1] + box_hi[1])); guard.hy = uniform_f(0.5f * (box_hi[1] - box_lo[1]) * (1.0f + 1e-6f));
                guard.cz = uniform_f(0.5f * (box_lo[2] + box_hi[2])); guard.hz = uniform_f(0.5f * (box_hi[2] - box_lo[2]) * (1.0f + 1e-6f));
                guard.slack = fp.a_zero;
            }
            if (CONE == 4) {
                // What bounds the rays of each pair of slots (128 consecutive rays of the queue) together:
                // a ball of origins and a cone of directions (see filter_range).
                guard.bounds = (const float*)(const void*)cone;
                guard.slack = fp.a_zero;
#pragma unroll
                for (int k = 0; k < RPL / 2; k++) {
                    const v2f sc2 = {sqrtf(L.hz[k].x), sqrtf(L.hz[k].y)};                       // hz = 1 / |V|^2
                    const v2f ux = L.vx[k] * sc2, uy = L.vy[k] * sc2, uz = L.vz[k] * sc2;
                    // members: the rays whose candidates count (a slot past the queue's end or a ray outside the
                    // filter's bounds lists nothing, so nothing has to be promised about it)
                    const bool in0 = (L.on >> (2 * k)) & 1u, in1 = (L.on >> (2 * k + 1)) & 1u;
                    // Rays reflected by one flat facet run together like a camera's; a pair of slots that
                    // straddles two facets holds two such beams some 10-40 degrees apart, and one cone around
                    // both reaches half the scene.  So: the ray farthest from the mean direction seeds one
                    // bundle, the ray farthest from that one the other, every ray joins the nearer seed.
                    float s0 = (in0 ? ux.x : 0.f) + (in1 ? ux.y : 0.f), s1 = (in0 ? uy.x : 0.f) + (in1 ? uy.y : 0.f),
                          s2 = (in0 ? uz.x : 0.f) + (in1 ? uz.y : 0.f);
                    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
                    const float inv0 = 1.0f / sqrtf(s0 * s0 + s1 * s1 + s2 * s2);
                    const float a0x = s0 * inv0, a0y = s1 * inv0, a0z = s2 * inv0;
                    const float d0x = in0 ? a0x * ux.x + a0y * uy.x + a0z * uz.x : 2.f, d0y = in1 ? a0x * ux.y + a0y * uy.y + a0z * uz.y : 2.f;
                    float low0 = fminf(d0x, d0y);
                    const float mine0 = low0;
                    for (int off = 32; off > 0; off >>= 1) low0 = fminf(low0, __shfl_xor(low0, off));
                    bool b0 = false, b1 = false;                       // which of the lane's two rays belong to the second bundle
                    // (NaN anywhere: low0 is NaN or 2, no split; the bundle below then reports "no bound")
                    if (low0 < 0.99985f) {                              // wave-uniform: more than about 1 degree apart
                        int src = (int)__builtin_ctzll(__ballot(mine0 == low0) | (1ull << 63));
                        const bool first = d0x <= d0y;
                        const float e1x = __shfl(first ? ux.x : ux.y, src), e1y = __shfl(first ? uy.x : uy.y, src),
                                    e1z = __shfl(first ? uz.x : uz.y, src);
                        const float d1x = in0 ? e1x * ux.x + e1y * uy.x + e1z * uz.x : 2.f, d1y = in1 ? e1x * ux.y + e1y * uy.y + e1z * uz.y : 2.f;
                        float low1 = fminf(d1x, d1y);
                        const float mine1 = low1;
                        for (int off = 32; off > 0; off >>= 1) low1 = fminf(low1, __shfl_xor(low1, off));
                        src = (int)__builtin_ctzll(__ballot(mine1 == low1) | (1ull << 63));
                        const bool first1 = d1x <= d1y;
                        const float e2x = __shfl(first1 ? ux.x : ux.y, src), e2y = __shfl(first1 ? uy.x : uy.y, src),
                                    e2z = __shfl(first1 ? uz.x : uz.y, src);
                        b0 = e2x * ux.x + e2y * uy.x + e2z * uz.x > d1x;
                        b1 = e2x * ux.y + e2y * uy.y + e2z * uz.y > d1y;
                    }
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const int sb = 2 * k + c;
                        const bool m0 = in0 && (b0 == (c == 1)), m1 = in1 && (b1 == (c == 1));
                        float sx = (m0 ? ux.x : 0.f) + (m1 ? ux.y : 0.f), sy = (m0 ? uy.x : 0.f) + (m1 ? uy.y : 0.f),
                              sz = (m0 ? uz.x : 0.f) + (m1 ? uz.y : 0.f);
                        float ox = (m0 ? L.px[k].x : 0.f) + (m1 ? L.px[k].y : 0.f), oy = (m0 ? L.py[k].x : 0.f) + (m1 ? L.py[k].y : 0.f),
                              oz = (m0 ? L.pz[k].x : 0.f) + (m1 ? L.pz[k].y : 0.f);
                        float cnt = (m0 ? 1.f : 0.f) + (m1 ? 1.f : 0.f);
                        float slow = fmaxf(m0 ? L.hz[k].x : 0.f, m1 ? L.hz[k].y : 0.f);        // largest 1 / |V|^2
                        for (int off = 32; off > 0; off >>= 1) {
                            sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off);
                            ox += __shfl_xor(ox, off); oy += __shfl_xor(oy, off); oz += __shfl_xor(oz, off);
                            cnt += __shfl_xor(cnt, off); slow = fmaxf(slow, __shfl_xor(slow, off));
                        }
                        const float inv = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
                        const float ax = sx * inv, ay = sy * inv, az = sz * inv;
                        const float icnt = 1.0f / cnt;
                        const float mx = ox * icnt, my = oy * icnt, mz = oz * icnt;
                        // per ray: cosine to the axis; its origin's offset from the centre, along and across the axis
                        const float ex = L.px[k].x - mx, ey = L.py[k].x - my, ez = L.pz[k].x - mz;
                        const float fx2 = L.px[k].y - mx, fy2 = L.py[k].y - my, fz2 = L.pz[k].y - mz;
                        const float el = ex * ax + ey * ay + ez * az, fl = fx2 * ax + fy2 * ay + fz2 * az;
                        const float e2 = ex * ex + ey * ey + ez * ez, f2 = fx2 * fx2 + fy2 * fy2 + fz2 * fz2;
                        float low = fminf(m0 ? ax * ux.x + ay * uy.x + az * uz.x : 1.f, m1 ? ax * ux.y + ay * uy.y + az * uz.y : 1.f);
                        float far2 = fmaxf(m0 ? e2 : 0.f, m1 ? f2 : 0.f);
                        float across2 = fmaxf(m0 ? e2 - el * el : 0.f, m1 ? f2 - fl * fl : 0.f);
                        float along1 = fmaxf(m0 ? fabsf(el) : 0.f, m1 ? fabsf(fl) : 0.f);
                        for (int off = 32; off > 0; off >>= 1) {
                            low = fminf(low, __shfl_xor(low, off)); far2 = fmaxf(far2, __shfl_xor(far2, off));
                            across2 = fmaxf(across2, __shfl_xor(across2, off)); along1 = fmaxf(along1, __shfl_xor(along1, off));
                        }
                        const float cb = low - 4e-6f;
                        const float sn = sqrtf(fmaxf(1.0f - cb * cb, 0.0f)) + 1e-6f;
                        const float pad = 1e-6f * (fabsf(mx) + fabsf(my) + fabsf(mz));
                        const float rfull = sqrtf(far2) * (1.0f + 1e-5f) + pad;
                        // A line through P with direction v stays within |P - m|'s part across v of the parallel line
                        // through m: at most the part across the axis plus the part along it times the sine of the
                        // angle between v and the axis.  (The square of the part across the axis is a difference of
                        // squares: its rounding, a few ulps of |P - m|^2, goes on top.)
                        const float racross = sqrtf(fmaxf(across2, 0.0f) + 8e-7f * far2) * (1.0f + 1e-5f) + along1 * sn * (1.0f + 1e-5f) + pad;
                        // (a NaN anywhere makes the axis NaN: fminf would drop it, the flags do not)
                        const bool finite = (ax == ax) && (ay == ay) && (az == az) && (rfull == rfull) && (racross == racross) && (low == low);
                        if (lane == 0) {
                            float* q = bundle_sh + 12 * sb;
                            q[0] = mx; q[1] = my; q[2] = mz; q[3] = fminf(racross, rfull); q[4] = rfull;
                            q[5] = ax; q[6] = ay; q[7] = az; q[8] = cb; q[9] = sn;
                        }
                        // |n . V| <= b_hi with V = |V| v: |n . v| <= b_hi / |V| <= b_hi * sqrt(largest 1 / |V|^2)
                        if (lane == 0) bundle_sh[12 * sb + 10] = fp.b_hi * sqrtf(slow) * (1.0f + 1e-4f) + 4e-6f;
                        guard.qon[sb] = __builtin_amdgcn_readfirstlane(!(cnt > 0.5f) ? -1 : ((finite && cb > 0.0f) ? 1 : 0));
                    }
                }
            }
            if (CONE == 1 || CONE == 2) {
                // what bounds the task's rays together (see filter_range)
                guard.bounds = (const float*)(const void*)cone;
                guard.cull = 0;
                float sx = 0.f, sy = 0.f, sz = 0.f;
                const v2f* bxs = (CONE == 1) ? L.hx : L.vx;     // camera rays: unit directions; shadow rays: directions, or points in the light's plane
                const v2f* bys = (CONE == 1) ? L.hy : L.vy;
                const v2f* bzs = (CONE == 1) ? L.hz : L.vz;
                const bool disc = CONE == 2 && cone_kind == 3;
#pragma unroll
                for (int k = 0; k < RPL / 2; k++) {
                    sx += bxs[k].x + bxs[k].y; sy += bys[k].x + bys[k].y;
                    if (!disc) sz += bzs[k].x + bzs[k].y;
                }
                for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off); }
                if (disc) {
                    const float mu = sx * (1.0f / (RT_WAVE * RPL)), mv = sy * (1.0f / (RT_WAVE * RPL));
                    float far2 = 0.f;
#pragma unroll
                    for (int k = 0; k < RPL / 2; k++) {
                        const float ax = bxs[k].x - mu, ay = bys[k].x - mv, bx2 = bxs[k].y - mu, by2 = bys[k].y - mv;
                        far2 = fmaxf(far2, fmaxf(ax * ax + ay * ay, bx2 * bx2 + by2 * by2));
                    }
                    for (int off = 32; off > 0; off >>= 1) far2 = fmaxf(far2, __shfl_xor(far2, off));
                    const float rb = sqrtf(far2) * (1.0f + 1e-5f) + 1e-6f * (fabsf(mu) + fabsf(mv));
                    guard.bx = uniform_f(mu); guard.by = uniform_f(mv); guard.bz = 0.f; guard.bcos = uniform_f(rb); guard.bsin = 0.f;
                    guard.cull = (rb == rb) ? 1 : 0;            // (a NaN among the points: no bound)
                } else {
                    const float inv = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
                    const float ax = sx * inv, ay = sy * inv, az = sz * inv;
                    float low = 1.0f;
#pragma unroll
                    for (int k = 0; k < RPL / 2; k++) {
                        low = fminf(low, fminf(ax * bxs[k].x + ay * bys[k].x + az * bzs[k].x, ax * bxs[k].y + ay * bys[k].y + az * bzs[k].y));
                    }
                    for (int off = 32; off > 0; off >>= 1) low = fminf(low, __shfl_xor(low, off));
                    // NaN anywhere: the sum, the axis and therefore every product is NaN, and fminf(1, NaN) = 1
                    const bool finite = (ax == ax) && (ay == ay) && (az == az);
                    const float cb = low - 4e-6f;
                    guard.bx = uniform_f(ax); guard.by = uniform_f(ay); guard.bz = uniform_f(az);
                    guard.bcos = uniform_f(cb);
                    guard.bsin = uniform_f(sqrtf(fmaxf(1.0f - cb * cb, 0.0f)) + 1e-6f);
                    guard.cull = (finite && cb > 0.0f) ? 1 : 0;
                    guard.cull = __builtin_amdgcn_readfirstlane(guard.cull);
                    // the fan: plane through the shared point spanned by the task's first and last direction
                    const float f0x = uniform_f(bxs[0].x), f0y = uniform_f(bys[0].x), f0z = uniform_f(bzs[0].x);
                    const float l0x = __shfl(bxs[RPL / 2 - 1].y, RT_WAVE - 1), l0y = __shfl(bys[RPL / 2 - 1].y, RT_WAVE - 1),
                                l0z = __shfl(bzs[RPL / 2 - 1].y, RT_WAVE - 1);
                    float nx = f0y * l0z - f0z * l0y, ny = f0z * l0x - f0x * l0z, nz = f0x * l0y - f0y * l0x;
                    const float nl = sqrtf(nx * nx + ny * ny + nz * nz);
                    nx /= nl; ny /= nl; nz /= nl;
                    float wide = 0.f;
#pragma unroll
                    for (int k = 0; k < RPL / 2; k++) {
                        wide = fmaxf(wide, fmaxf(fabsf(nx * bxs[k].x + ny * bys[k].x + nz * bzs[k].x),
                                                 fabsf(nx * bxs[k].y + ny * bys[k].y + nz * bzs[k].y)));
                    }
                    for (int off = 32; off > 0; off >>= 1) wide = fmaxf(wide, __shfl_xor(wide, off));
                    const float sg = wide * (1.0f + 1e-5f) + 4e-6f;
                    const bool fan = (nx == nx) && (ny == ny) && (nz == nz) && nl > 1e-6f && sg < 0.5f;   // (NaN: no fan)
                    guard.fx = uniform_f(fan ? nx : 0.f); guard.fy = uniform_f(fan ? ny : 0.f); guard.fz = uniform_f(fan ? nz : 0.f);
                    guard.fsin = uniform_f(fan ? sg : 0.f);
                    guard.fcos = uniform_f(fan ? sqrtf(1.0f - sg * sg) * (1.0f - 1e-6f) : 1.0f);
                }
            }
            if (CONE == 1) {
                // every slot of every lane holds the same origin (the eye in this group's space):
                // say so, and the 24 registers become 3 scalar ones
                const float ex = uniform_f(L.px[0].x), ey = uniform_f(L.py[0].x), ez = uniform_f(L.pz[0].x);
#pragma unroll
                for (int k = 0; k < RPL / 2; k++) { L.px[k] = {ex, ex}; L.py[k] = {ey, ey}; L.pz[k] = {ez, ez}; }
            }
#ifdef RT_PHASE_CLOCK
            if (g == 0) wk.t[1] = RT_CLOCK() - clk_t1;
#endif
            // each part takes its share of every axis range
#pragma unroll
            for (int axis = 0; axis < 3; axis++) {
                const int alen = fcnt[axis];
                // (parts begin at multiples of 64, like the axis ranges: clusters are never cut)
                const int per_part = __builtin_amdgcn_readfirstlane((((alen + split - 1) / split) + 63) & ~63);
                const int abeg = fax[axis] + part * per_part;
                const int aend = min(fax[axis] + alen, abeg + per_part);
                if (abeg >= aend) continue;
                constexpr int KIND = (MODE == 0) ? 1 : 2;
#define RT_RANGE(AX, SH, PR) filter_range<MODE, RPL, AX, ROBUST, SH, PR, WORK, PLANE>(filt, shared, cone, guard, A, L, fp, abeg, aend, my_cand, ray0, ray_end, lane, wk)
                if (CONE == 1) {
                    if (axis == 0) RT_RANGE(0, 1, 1);
                    if (axis == 1) RT_RANGE(1, 1, 1);
                    if (axis == 2) RT_RANGE(2, 1, 1);
                } else if (CONE == 4) {
                    if (axis == 0) RT_RANGE(0, 0, 4);
                    if (axis == 1) RT_RANGE(1, 0, 4);
                    if (axis == 2) RT_RANGE(2, 0, 4);
                } else if (CONE == 2 && cone_kind == 2) {
                    if (axis == 0) RT_RANGE(0, 0, 2);
                    if (axis == 1) RT_RANGE(1, 0, 2);
                    if (axis == 2) RT_RANGE(2, 0, 2);
                } else if (CONE == 2 && cone_kind == 3) {
                    const float sx = uniform_f(dir_x), sy = uniform_f(dir_y), sz = uniform_f(dir_z);
                    FilterLane<RPL> D;
                    D.on = L.on;
#pragma unroll
                    for (int k = 0; k < RPL / 2; k++) {
                        D.px[k] = L.px[k]; D.py[k] = L.py[k]; D.pz[k] = L.pz[k];
                        D.hx[k] = L.vx[k]; D.hy[k] = L.vy[k]; D.hz[k] = L.vz[k];
                        D.vx[k] = {sx, sx}; D.vy[k] = {sy, sy}; D.vz[k] = {sz, sz};
                    }
#define RT_RANGE_D(AX) filter_range<MODE, RPL, AX, ROBUST, 2, 3, WORK, PLANE>(filt, shared, cone, guard, A, D, fp, abeg, aend, my_cand, ray0, ray_end, lane, wk)
                    if (axis == 0) RT_RANGE_D(0);
                    if (axis == 1) RT_RANGE_D(1);
                    if (axis == 2) RT_RANGE_D(2);
#undef RT_RANGE_D
                } else if (shared_kind == 0) {
                    if (axis == 0) RT_RANGE(0, 0, 0);
                    if (axis == 1) RT_RANGE(1, 0, 0);
                    if (axis == 2) RT_RANGE(2, 0, 0);
                } else {
                    if (axis == 0) RT_RANGE(0, KIND, 0);
                    if (axis == 1) RT_RANGE(1, KIND, 0);
                    if (axis == 2) RT_RANGE(2, KIND, 0);
                }
#undef RT_RANGE
            }
        }
#ifdef RT_PHASE_CLOCK
        wk.t[5] = RT_CLOCK() - clk_t0;
        clk_sum[0] += 1;
        for (int f = 0; f < 6; f++) clk_sum[1 + f] += wk.t[f];
        clk_loop += wk.t[6];
        if (WORK && lane == 0 && A.work && (CONE == 4 || CONE == 2)) {
            // histogram of task durations: bucket b holds tasks of [2^(14+b), 2^(15+b)) cycles, in the row of a kind
            // of scan the bounded configuration does not launch (reflection rays: row 0, shadow rays: row 5)
            int b = 31 - __builtin_clz(wk.t[5] | 1u) - 14;
            b = b < 0 ? 0 : (b > 10 ? 10 : b);
            atomicAdd(&A.work[(CONE == 4 ? 0 : 5) * RT_WORK_FIELDS + b], 1ull);
        }
        for (int f = 0; f < RT_WORK_FIELDS; f++) wk.v[f] = 0;
#endif
        if (WORK && lane == 0 && A.work) {
            const int kind = (CONE == 1) ? 1 : (CONE == 4) ? 4 : (CONE == 2) ? cone_kind : (MODE == 0 ? 0 : 5);
#pragma unroll
            for (int f = 0; f < RT_WORK_FIELDS; f++)
                if (wk.v[f]) atomicAdd(&A.work[kind * RT_WORK_FIELDS + f], (unsigned long long)wk.v[f]);
        }
    }
#ifdef RT_PHASE_CLOCK
    if (WORK && lane == 0 && A.work && clk_sum[0] != 0) {
        // (a launch holds one kind of scan, except that shadow launches mix the two kinds of light: booked under the plain kinds' rows' neighbour)
        const int kind = (CONE == 1) ? 1 : (CONE == 4) ? 4 : (CONE == 2) ? 2 : (MODE == 0 ? 0 : 5);
        for (int f = 0; f < 7; f++) atomicAdd(&A.work[kind * RT_WORK_FIELDS + f], clk_sum[f]);
        atomicAdd(&A.work[kind * RT_WORK_FIELDS + 7], (unsigned long long)(RT_CLOCK() - clk_wave));
        atomicAdd(&A.work[kind * RT_WORK_FIELDS + 8], 1ull);
        atomicAdd(&A.work[kind * RT_WORK_FIELDS + 9], clk_loop);
    }
#endif
}
