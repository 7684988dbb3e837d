// Between the trace launches of a level: compaction of the hits and the next level's rays (k_hit_counts, k_hit),
// Phong shading (k_phong), the deepest-first fold into the frame (k_fold), the ray counters (k_count).
#pragma once

#include "rt_trace_args.h"

// ---------------------------------------------------------------------------
// After the nearest-hit kernels: the part of Scene::traceRay behind the object loop
// (raytracer.cpp:531-538) for every ray that hit, compacted.
// ---------------------------------------------------------------------------
struct HitArgs {
    const DevScene* sc;
    RayQueue cur, next;                  // next[h] = (hit point, reflected direction)
    const unsigned long long* keys;
    unsigned long long* next_keys;
    const uint32_t* n_rays;
    uint32_t* n_hits;                    // out: the number of hits = the next queue's length
    const uint32_t* hit_counts;          // per block: hits in its range of the queue (k_hit_counts, same grid)
    LevelStore lvl;
    int32_t* parent_next;                // the level above's LevelStore::next (null: none kept, or the camera rays' level)
    int32_t chain;                       // 1: keep the chains (LevelStore::next) for k_fold_all
    int32_t camera_rays;                 // 1: `cur` holds the pass's camera rays (origin = the eye, not stored)
    HitAux aux;
    uint32_t* occluded;
    int32_t occl_words;
    float4* shadow_scan;                 // see TraceArgs (null: none kept); written for every light
    uint32_t shadow_stride;
    const double* light_frame;           // [n_lights][8], directional lights (see TraceArgs)
};

// The hits keep the order of their rays (ray i before ray j => its hit before j's): rays that
// are neighbours on the film stay neighbours in the next level's queue and in the shadow
// queue, which is what the bounding tests of k_filter live on (a task's 512 rays ask for few
// triangles when they run close together).  Every block owns a contiguous range of the queue.
// k_hit_counts counts the hits per range, nothing else; every block of k_hit (same grid) adds up the counts of the
// ranges before its own, and the block of the last range publishes the total as the next queue's length.
// (Earlier forms and what they cost: the counting kernel's last block -- found by a ticket behind a fence of device
// scope per block -- turning the counts into offsets: 28 us per launch, most of it the 2048 tickets on one address and
// the L2 write-backs of the fences.  One kernel with a look-back along per-block descriptors: a frame on its own
// 1 % faster, four calls in flight 7 % slower -- blocks that wait for a predecessor another kernel's waves keep from
// starting hold their wave slots while they wait.)
__device__ __forceinline__ unsigned int hit_range_length(unsigned int n, unsigned int n_blocks) {
    const unsigned int per = (n + n_blocks - 1) / n_blocks;
    return (per + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
}

__global__ void __launch_bounds__(RT_BLOCK)
k_hit_counts(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ n_rays, uint32_t* __restrict__ counts) {
    __shared__ unsigned int wave_hits[RT_BLOCK / RT_WAVE];
    const unsigned int n = *n_rays;
    const unsigned int per = hit_range_length(n, gridDim.x);
    const unsigned int begin = blockIdx.x * per;
    unsigned int mine = 0;
    for (unsigned int it = 0; it < per && begin + it < n; it += RT_BLOCK) {
        const unsigned int i = begin + it + threadIdx.x;
        if (i < n && keys[i] != RT_MISS_KEY) mine++;
    }
    for (int off = RT_WAVE / 2; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & (RT_WAVE - 1)) == 0) wave_hits[threadIdx.x / RT_WAVE] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int sum = 0;
#pragma unroll
        for (int w = 0; w < RT_BLOCK / RT_WAVE; w++) sum += wave_hits[w];
        counts[blockIdx.x] = sum;                                     // (blocks without rays write 0)
    }
}

__global__ void __launch_bounds__(RT_BLOCK)
k_hit(HitArgs A) {
    const DevScene* __restrict__ sc = A.sc;
    __shared__ unsigned int wave_hits[RT_BLOCK / RT_WAVE];
    __shared__ unsigned int range_base;
    const unsigned int n = *A.n_rays;
    const int lane = threadIdx.x & (RT_WAVE - 1);
    const int wave = threadIdx.x / RT_WAVE;
    const unsigned int per = hit_range_length(n, gridDim.x);
    const unsigned int begin = blockIdx.x * per;
    // Only blocks that own rays take part; an empty queue is block 0's to report.
    if (n == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) *A.n_hits = 0u; return; }
    const unsigned int active = (n + per - 1) / per;
    if (blockIdx.x >= active) return;
    {
        // hits of the ranges before this one
        unsigned int before = 0;
        for (unsigned int k = threadIdx.x; k < blockIdx.x; k += RT_BLOCK) before += A.hit_counts[k];
        for (int off = RT_WAVE / 2; off > 0; off >>= 1) before += __shfl_xor(before, off);
        if (lane == 0) wave_hits[wave] = before;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int sum = 0;
#pragma unroll
            for (int w = 0; w < RT_BLOCK / RT_WAVE; w++) sum += wave_hits[w];
            range_base = sum;
            if (blockIdx.x == active - 1u) *A.n_hits = sum + A.hit_counts[blockIdx.x];      // the next queue's length
        }
        __syncthreads();
    }
    unsigned int block_base = range_base;
    for (unsigned int it = 0; it < per && begin + it < n; it += RT_BLOCK) {
        const unsigned int i = begin + it + threadIdx.x;
        unsigned long long key = (i < n) ? A.keys[i] : RT_MISS_KEY;
        const bool hit = key != RT_MISS_KEY;
        // position of the hit: hits of earlier blocks, of earlier iterations, of earlier waves, of earlier lanes
        const unsigned long long ballot = __ballot(hit);
        if (lane == 0) wave_hits[wave] = (unsigned int)__popcll(ballot);
        __syncthreads();
        unsigned int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < RT_BLOCK / RT_WAVE; w++) {
            const unsigned int cnt = wave_hits[w];
            if (w < wave) before += cnt;
            total += cnt;
        }
        __syncthreads();
        const unsigned int h = block_base + before + (unsigned int)__popcll(ballot & ((1ull << lane) - 1ull));
        block_base += total;
        if (!hit) continue;
        const int obj = (int)(unsigned int)(key & 0xffffffffu);
        const int xf = sc->obj_transform[obj];
        const double* X = sc->xforms + 48 * (size_t)xf;
        const int ident = sc->xf_identity[xf];
        // (camera rays all start at the eye; their queue holds no origins)
        d3 wpos = A.camera_rays ? ld3(sc->cam_eye) : mk3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]);
        d3 wvec = mk3(A.cur.dx[i], A.cur.dy[i], A.cur.dz[i]);
        d3 rpos = wpos, rvec = wvec;
        if (!ident) { rpos = xf_point(X + 16, wpos); rvec = xf_dir(X + 16, wvec); }
        const int slot = sc->obj_slot[obj];
        float t;
        d3 nobj;
        d3 point;
        if (sc->obj_kind[obj] == 0) {
            const TriExact& T = sc->tri_exact[slot];
            t = tri_intersect_exact(T, rpos, rvec, sc->eps);
            hit_metric(X, ident, rpos, rvec, t, &point);
            d3 local = ident ? point : xf_point(X + 16, point);      // inverseTransform * minIntersection (raytracer.cpp:533)
            if (sc->obj_interp[obj]) {
                // Triangle::normal, interpolated (raytracer.cpp:360-369)
                const double* g = sc->obj_geom + 9 * (size_t)obj;
                const double* nn = sc->obj_normals + 9 * (size_t)obj;
                d3 v1 = ld3(g), v2 = ld3(g + 3), v3 = ld3(g + 6);
                d3 fn = ld3(T.n);
                float a1 = (float)dot3(fn, cross3(sub3(v2, v1), sub3(v3, v1)));
                float a2 = (float)dot3(fn, cross3(sub3(v2, local), sub3(v3, local)));
                float a3 = (float)dot3(fn, cross3(sub3(v3, local), sub3(v1, local)));
                float alpha = a2 / a1;
                float beta = a3 / a1;
                float gamma = 1.0f - alpha - beta;
                nobj = add3(add3(scl3(ld3(nn), (double)alpha), scl3(ld3(nn + 3), (double)beta)),
                            scl3(ld3(nn + 6), (double)gamma));
            } else {
                nobj = cross3(ld3(T.e2), ld3(T.e1));                 // raytracer.cpp:371
            }
        } else {
            const SphereRec& S = sc->spheres[slot];
            t = sph_intersect_exact(S, rpos, rvec, sc->eps);
            hit_metric(X, ident, rpos, rvec, t, &point);
            d3 local = ident ? point : xf_point(X + 16, point);
            nobj = sub3(local, ld3(S.c));                            // Sphere::normal (raytracer.cpp:416)
        }
        d3 nw = ident ? nobj : xf_dir(X + 32, nobj);                 // transformNormal (raytracer.cpp:260-266)
        nw = nrm3(nw);
        // `ray` after the object loop is the LAST object's transformed ray (raytracer.cpp:518-520, 536)
        d3 rv = wvec;
        if (!sc->xf_identity[sc->last_xf]) rv = xf_dir(sc->xforms + 48 * (size_t)sc->last_xf + 16, wvec);
        d3 refl = sub3(rv, scl3(nw, 2 * dot3(rv, nw)));
        refl = nrm3(refl);
        A.next.ox[h] = point.x; A.next.oy[h] = point.y; A.next.oz[h] = point.z;
        A.next.dx[h] = refl.x; A.next.dy[h] = refl.y; A.next.dz[h] = refl.z;
        if (A.next.scan) {
            const float fx = (float)point.x, fy = (float)point.y, fz = (float)point.z;
            const float gx = (float)refl.x, gy = (float)refl.y, gz = (float)refl.z;
            float ux, uy, uz;
            scan_h_any(fx, fy, fz, gx, gy, gz, &ux, &uy, &uz);
            scan_store(A.next, h, fx, fy, fz, gx, gy, gz, ux, uy, uz);
            if (A.shadow_scan) {
                for (int l = 0; l < sc->n_lights; l++) {
                    const DevLight& lt = sc->lights[l];
                    float4 rec;
                    if (lt.is_point) {
                        // Light::lightRay (raytracer.cpp:108-119): the same expressions as load_world_ray<1>
                        const d3 dir = nrm3(sub3(ld3(lt.pos), point));
                        rec = make_float4((float)dir.x, (float)dir.y, (float)dir.z, scan_len_point(lt.pos, point.x, point.y, point.z));
                    } else {
                        float px, py, pz;
                        scan_plane_point(A.light_frame + 8 * (size_t)l, point.x, point.y, point.z, &px, &py, &pz);
                        rec = make_float4(px, py, pz, 0.0f);
                    }
                    A.shadow_scan[(size_t)l * A.shadow_stride + h] = rec;
                }
            }
        }
        A.next_keys[h] = RT_MISS_KEY;
        A.aux.nx[h] = nw.x; A.aux.ny[h] = nw.y; A.aux.nz[h] = nw.z;
        A.aux.vx[h] = rv.x; A.aux.vy[h] = rv.y; A.aux.vz[h] = rv.z;
        A.lvl.parent[h] = (int32_t)i;
        A.lvl.obj[h] = obj;
        if (A.chain) {
            A.lvl.next[h] = -1;
            if (A.parent_next) A.parent_next[i] = (int32_t)h;
        }
        for (int w = 0; w < A.occl_words; w++) A.occluded[(size_t)h * A.occl_words + w] = 0u;
    }
}

// ---------------------------------------------------------------------------
// Scene::phongShade (raytracer.cpp:560-581) given the shadow results.
// ---------------------------------------------------------------------------
struct PhongArgs {
    const DevScene* sc;
    RayQueue next;                       // origins = hit points
    HitAux aux;
    LevelStore lvl;
    const uint32_t* occluded;
    int32_t occl_words;
    uint32_t cap;                        // stride of lvl.phong / lvl.child component planes
    const uint32_t* n_hits;
    int32_t init_child;                  // 1: the fold goes level by level and wants lvl.child at zero
};

__global__ void __launch_bounds__(RT_BLOCK)
k_phong(PhongArgs A) {
    const DevScene* __restrict__ sc = A.sc;
    const unsigned int n = *A.n_hits;
    for (unsigned int h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
        const DevMaterial& m = sc->materials[sc->obj_material[A.lvl.obj[h]]];
        d3 point = mk3(A.next.ox[h], A.next.oy[h], A.next.oz[h]);
        d3 normal = mk3(A.aux.nx[h], A.aux.ny[h], A.aux.nz[h]);
        d3 rayvec = mk3(A.aux.vx[h], A.aux.vy[h], A.aux.vz[h]);
        d3 total = mk3(0, 0, 0);
        total = add3(total, mul3(ld3(sc->ambient), ld3(m.ka)));
        for (int i = 0; i < sc->n_lights; i++) {
            if ((A.occluded[(size_t)h * A.occl_words + (i >> 5)] >> (i & 31)) & 1u) continue;
            const DevLight& lt = sc->lights[i];
            d3 lpos = ld3(lt.pos), color = ld3(lt.color);
            d3 intensity = color;
            if (lt.falloff != 0) {                       // Light::intensityAt (raytracer.cpp:121-128)
                float d = dist_f(lpos, point);
                double dd = (double)d;
                double p = lt.falloff == 1 ? dd : (lt.falloff == 2 ? dd * dd : pow(dd, (double)lt.falloff));
                double inv = 1. / p;
                intensity = mk3(color.x * inv, color.y * inv, color.z * inv);
            }
            d3 l = lt.is_point ? sub3(lpos, point) : neg3(lpos);
            d3 v = neg3(rayvec);
            d3 r = add3(neg3(l), scl3(normal, 2 * dot3(l, normal)));
            l = nrm3(l); v = nrm3(v); r = nrm3(r);
            double ln = dot3(l, normal), rv = dot3(r, v);
            double diff = ln > 0.0f ? ln : 0.0f;          // MAX(l * normal, 0.0f) (algebra3.h:34)
            double sbase = rv > 0.0f ? rv : 0.0f;
            double spec = pow(sbase, m.sp);
            d3 coeffs = add3(scl3(ld3(m.kd), diff), scl3(ld3(m.ks), spec));
            total = add3(total, mul3(intensity, coeffs));
        }
        A.lvl.phong[h] = total.x; A.lvl.phong[h + (size_t)A.cap] = total.y; A.lvl.phong[h + 2 * (size_t)A.cap] = total.z;
        if (A.init_child) { A.lvl.child[h] = 0.0; A.lvl.child[h + (size_t)A.cap] = 0.0; A.lvl.child[h + 2 * (size_t)A.cap] = 0.0; }
    }
}

// ---------------------------------------------------------------------------
// One step of the deepest-first fold.  For level k > 0 the value goes into the
// parent hit's `child`; for level 0 it is quantised into the frame
// (floatToByte / writeToFilm / the uchar store, raytracer.cpp:174-203, 225).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quantise(double c) {
    double scaled = c * 255.0f;
    int v = (scaled >= -2147483648.0 && scaled < 2147483648.0) ? (int)scaled : (-2147483647 - 1);
    if (v > 255) v = 255;
    return (uint32_t)(v & 0xff);
}

struct FoldArgs {
    const DevScene* sc;
    LevelStore lvl, parent_lvl;
    uint32_t cap;
    const uint32_t* n_hits;
    int32_t level;
    const int32_t* pixel_of_ray;
    uint32_t* frame;
};

__global__ void __launch_bounds__(RT_BLOCK)
k_fold(FoldArgs A) {
    const DevScene* __restrict__ sc = A.sc;
    const unsigned int n = *A.n_hits;
    const size_t cap = A.cap;
    for (unsigned int h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
        const DevMaterial& m = sc->materials[sc->obj_material[A.lvl.obj[h]]];
        d3 phong = mk3(A.lvl.phong[h], A.lvl.phong[h + cap], A.lvl.phong[h + 2 * cap]);
        d3 child = mk3(A.lvl.child[h], A.lvl.child[h + cap], A.lvl.child[h + 2 * cap]);
        d3 val = add3(phong, mul3(child, ld3(m.kr)));     // raytracer.cpp:542-543
        const int parent = A.lvl.parent[h];
        if (A.level > 0) {
            A.parent_lvl.child[parent] = val.x;
            A.parent_lvl.child[parent + cap] = val.y;
            A.parent_lvl.child[parent + 2 * cap] = val.z;
        } else {
            A.frame[A.pixel_of_ray[parent]] =
                quantise(val.x) | (quantise(val.y) << 8) | (quantise(val.z) << 16) | 0xff000000u;
        }
    }
}

// The whole fold in one launch, for scenes of at most RT_FOLD_MAX_LEVELS levels: a thread per hit of a camera ray walks
// down the chain of reflected hits (LevelStore::next) and folds it from its deepest hit up -- the same additions and
// multiplications in the same order as the level-by-level fold (a hit without a reflected hit takes 0 for it,
// raytracer.cpp:542-543).  Compaction keeps the order of the rays, so neighbouring threads read neighbouring entries
// at every level.  Block 0 also does what k_count does.
struct FoldAllArgs {
    const DevScene* sc;
    const int32_t* parent;               // [levels][cap]
    const int32_t* obj;                  // [levels][cap]
    const double* phong;                 // [levels][3][cap]
    const int32_t* next;                 // level k at next + k * 6 * cap (see LevelStore::next)
    uint32_t cap;
    int32_t levels;
    const uint32_t* counts;              // counts[k + 1]: hits of level k
    const int32_t* pixel_of_ray;
    uint32_t* frame;
    DevCounters* counters;
    uint32_t* hint;                      // pinned host memory for the pass's queue lengths ([levels + 2]), or null
};

// `hint`: the queue lengths also go to pinned host memory, for the launch geometry of the next pass of this size
// (written from here, no copy behind the pass: a copy to the host ends in a release of system scope, which
// writes the L2 back under everything else that is running).
__device__ __forceinline__ void count_pass(const DevScene* __restrict__ sc, const uint32_t* __restrict__ counts, int levels,
                                           DevCounters* __restrict__ c, uint32_t* hint) {
    if (hint)
        for (int k = 0; k < levels + 2; k++) __hip_atomic_store(&hint[k], counts[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long prims = (unsigned long long)sc->n_prims_nominal;
    unsigned long long rays = 0, hits = 0;
    for (int k = 0; k < levels; k++) { rays += counts[k]; hits += counts[k + 1]; }
    // (atomic: passes of several render calls in flight end side by side)
    atomicAdd(&c->rays_primary, (unsigned long long)counts[0]);
    atomicAdd(&c->rays_reflect, rays - counts[0]);
    atomicAdd(&c->rays_shadow, hits * (unsigned long long)sc->n_lights);
    atomicAdd(&c->hits, hits);
    atomicAdd(&c->closest_tests, rays * prims);
    atomicAdd(&c->anyhit_tests, hits * (unsigned long long)sc->n_lights * prims);
}

__global__ void __launch_bounds__(RT_BLOCK)
k_fold_all(FoldAllArgs A) {
    const DevScene* __restrict__ sc = A.sc;
    if (blockIdx.x == 0 && threadIdx.x == 0) count_pass(sc, A.counts, A.levels, A.counters, A.hint);
    const unsigned int n = A.counts[1];
    const size_t cap = A.cap;
    for (unsigned int h0 = blockIdx.x * blockDim.x + threadIdx.x; h0 < n; h0 += gridDim.x * blockDim.x) {
        int at[RT_FOLD_MAX_LEVELS];
        at[0] = (int)h0;
        int deepest = 0;
        bool on = true;
#pragma unroll
        for (int k = 0; k + 1 < RT_FOLD_MAX_LEVELS; k++) {
            at[k + 1] = 0;
            if (on && k + 1 < A.levels) {
                const int down = A.next[(size_t)k * 6 * cap + (size_t)at[k]];
                on = down >= 0;
                if (on) { at[k + 1] = down; deepest = k + 1; }
            } else {
                on = false;
            }
        }
        d3 val = mk3(0.0, 0.0, 0.0);
#pragma unroll
        for (int k = RT_FOLD_MAX_LEVELS - 1; k >= 0; k--) {
            if (k <= deepest) {
                const size_t h = (size_t)at[k];
                const DevMaterial& m = sc->materials[sc->obj_material[A.obj[(size_t)k * cap + h]]];
                const double* ph = A.phong + (size_t)k * 3 * cap;
                const d3 phong = mk3(ph[h], ph[h + cap], ph[h + 2 * cap]);
                val = add3(phong, mul3(val, ld3(m.kr)));     // raytracer.cpp:542-543
            }
        }
        A.frame[A.pixel_of_ray[A.parent[h0]]] =
            quantise(val.x) | (quantise(val.y) << 8) | (quantise(val.z) << 16) | 0xff000000u;
    }
}

// Adds one pass's queue lengths to the running counters.
__global__ void k_count(const DevScene* __restrict__ sc, const uint32_t* __restrict__ counts, int levels,
                        DevCounters* __restrict__ c, uint32_t* hint) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    count_pass(sc, counts, levels, c, hint);
}

