// The trace loop as a wavefront pipeline.  One "level" is one generation of
// rays (level 0 = camera rays, level k = k-th reflection).  Per level:
//   k_closest  nearest hit of every ray in the level's queue (brute force)
//   k_hit      compacts the rays that hit something (ballot + one atomic per
//              wave), computes hit point, normal and reflection, and writes
//              the next level's ray queue
//   k_anyhit   one shadow ray per (hit, light)
//   k_phong    Phong sum over the unshadowed lights
// and after the last level k_fold walks the levels deepest-first to evaluate
//   phong_0 + kr_0 * (phong_1 + kr_1 * (...))            (raytracer.cpp:542-543)
// in the reference's order.  Queue lengths live in device memory; nothing
// is read back by the host between kernels.
//
// k_closest / k_anyhit scan every primitive for every ray.  Regular triangles
// go through an FP32 filter (16 floats per triangle, wave-uniform, so the
// compiler fetches them with scalar loads and they never occupy vector
// registers); pairs the filter cannot rule out are queued per lane in LDS and
// re-evaluated in the reference's mixed double/float arithmetic, so every
// accept/reject decision and every hit distance is the reference's own.
#pragma once

#include "rt_device.h"

#define RT_BLOCK 256
#define RT_WAVE 64
#define RT_CAND_CAP 64          // per-lane candidate slots in LDS (shared by the lane's rays)
#define RT_BATCH 8              // triangles filtered between two capacity checks
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 1     // minimum waves per SIMD the trace kernels are compiled for
#endif
#define RT_RPL 4                // rays per lane in the trace kernels
#define RT_SLOT_SHIFT 14        // candidate entry (u16) = triangle index relative to a chunk base | ray slot << 14
#define RT_CHUNK (1 << RT_SLOT_SHIFT)

struct RayQueue {               // structure of arrays, one entry per ray
    double *ox, *oy, *oz, *dx, *dy, *dz;
};

struct LevelStore {             // one entry per hit of a level
    int32_t* parent;            // index of the ray in this level's queue (= hit index one level up, or the pixel)
    int32_t* obj;               // Scene::objects index that was hit
    double* phong;              // [3][cap] phongShade result
    double* child;              // [3][cap] value returned by the reflected ray (0 when it misses)
};

struct HitAux {                 // per hit of the current level only
    double *nx, *ny, *nz;       // shading normal
    double *vx, *vy, *vz;       // the last object's transformed ray direction (raytracer.cpp:518-520)
};

struct PassDesc {               // rows traced by one pass, see rt_render_tiles
    int32_t y_start, tile_rows, stride_rows, q0;   // pass row q -> y = y_start + (q / tile_rows) * stride_rows + q % tile_rows
};

struct DevCounters {
    unsigned long long rays_primary, rays_reflect, rays_shadow, hits;
    unsigned long long closest_tests, anyhit_tests, exact_tests;
};

// ---------------------------------------------------------------------------
// Camera::generateRay (raytracer.cpp:441-447) for the pixels of a pass, plus the
// black background of pixels whose ray hits nothing (raytracer.cpp:546).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(RT_BLOCK)
k_primary(const DevScene* __restrict__ sc, PassDesc pd, int n_pix, RayQueue q,
          unsigned long long* __restrict__ keys, int32_t* __restrict__ pixel_of_ray,
          uint32_t* __restrict__ frame, uint32_t* __restrict__ counts) {
    int W = sc->width, H = sc->height;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += gridDim.x * blockDim.x) {
        int qrow = pd.q0 + p / W, x = p % W;
        int y = pd.y_start + (qrow / pd.tile_rows) * pd.stride_rows + qrow % pd.tile_rows;
        float u = (float)x / (float)W;
        float v = (float)y / (float)H;
        d3 ll = ld3(sc->cam_ll), lr = ld3(sc->cam_lr), ul = ld3(sc->cam_ul), ur = ld3(sc->cam_ur);
        d3 bottom = add3(scl3(ll, (double)(1 - u)), scl3(lr, (double)u));
        d3 top = add3(scl3(ul, (double)(1 - u)), scl3(ur, (double)u));
        d3 c = add3(scl3(bottom, (double)(1 - v)), scl3(top, (double)v));
        d3 eye = ld3(sc->cam_eye);
        d3 dir = sub3(c, eye);
        q.ox[p] = eye.x; q.oy[p] = eye.y; q.oz[p] = eye.z;
        q.dx[p] = dir.x; q.dy[p] = dir.y; q.dz[p] = dir.z;
        keys[p] = RT_MISS_KEY;
        int pix = (H - 1 - y) * W + x;          // row 0 of the frame is the top scanline (raytracer.cpp:225)
        pixel_of_ray[p] = pix;
        frame[pix] = 0xff000000u;               // R,G,B = 0, alpha = 255
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = (uint32_t)n_pix;
}

// ---------------------------------------------------------------------------
// Brute-force scan.  MODE 0: nearest hit; MODE 1: any hit (shadow).
// ---------------------------------------------------------------------------
struct TraceArgs {
    RayQueue rays;                       // MODE 0: the level's rays.  MODE 1: origins = hit points
    const uint32_t* n_items;             // MODE 0: number of rays; MODE 1: number of hits (shadow rays = hits * n_lights)
    unsigned long long* keys;            // MODE 0: per ray, atomicMin of (metric bits, object index)
    const int32_t* source_obj;           // MODE 1: object each hit lies on (excluded, raytracer.cpp:553)
    uint32_t* occluded;                  // MODE 1: per hit, one bit per light
    int32_t occl_words;                  // MODE 1: 32-bit words per hit
    int32_t force_exact;                 // 1: bypass the FP32 filter
    DevCounters* counters;
};

// World-space ray number `ray` of a launch.  MODE 0: entry `ray` of the level's queue.
// MODE 1: shadow ray of hit ray / n_lights towards light ray % n_lights (Light::lightRay,
// raytracer.cpp:108-119); *source = the object the hit lies on.
template <int MODE>
__device__ __forceinline__ void load_world_ray(const TraceArgs& A, const DevScene* __restrict__ sc,
                                               unsigned long long ray, int n_lights, d3* pos, d3* vec, int* source) {
    if (MODE == 0) {
        *pos = mk3(A.rays.ox[ray], A.rays.oy[ray], A.rays.oz[ray]);
        *vec = mk3(A.rays.dx[ray], A.rays.dy[ray], A.rays.dz[ray]);
        *source = -1;
    } else {
        const unsigned int hit = (unsigned int)(ray / n_lights), light = (unsigned int)(ray % n_lights);
        *pos = mk3(A.rays.ox[hit], A.rays.oy[hit], A.rays.oz[hit]);
        const DevLight& lt = sc->lights[light];
        d3 lv = lt.is_point ? sub3(ld3(lt.pos), *pos) : neg3(ld3(lt.pos));
        *vec = nrm3(lv);
        *source = A.source_obj[hit];
    }
}

// Polygon::transformRay (raytracer.cpp:251-259); X = {M, M^-1, M^-T}.
__device__ __forceinline__ void to_object_space(const double* X, int identity, d3* pos, d3* vec) {
    if (!identity) {
        d3 p = xf_point(X + 16, *pos), v = xf_dir(X + 16, *vec);
        *pos = p; *vec = v;
    }
}

// One primitive tested in reference precision; folds the result into the ray's running
// answer.  MODE 0: (metric, object) with the reference's strict '<' and lowest-index tie
// rule (raytracer.cpp:525).  MODE 1: any hit other than the source (raytracer.cpp:553;
// a NaN result counts as a hit there because NaN != MAXFLOAT).
template <int MODE>
__device__ __forceinline__ void fold_result(float t, int obj, d3 pos, d3 vec, const double* X, int identity,
                                            int source, float* best_d, int* best_obj, bool* occluded) {
    if (MODE == 0) {
        d3 world;
        float d = hit_metric(X, identity, pos, vec, t, &world);
        if (t != RT_MISS_T && (d < *best_d || (d == *best_d && *best_obj >= 0 && obj < *best_obj))) {
            *best_d = d; *best_obj = obj;
        }
    } else {
        if (t != RT_MISS_T && obj != source) *occluded = true;
    }
}

// The FP32 filter: false only when the reference-precision test of this (ray, triangle)
// pair is certain to miss.  f = {n, d0 | S, s0 | T, t0 | U, u0}: the triangle's plane and its
// three barycentric coordinates as affine functions with their margins folded in.
// The triangle's 16 floats arrive in scalar registers.
__device__ __forceinline__ bool filter_keep(const v16f f, float px, float py, float pz, float vx, float vy, float vz,
                                            const FilterParams fp) {
    float b = f[0] * vx;
    b = __builtin_fmaf(f[1], vy, b);
    b = __builtin_fmaf(f[2], vz, b);
    float a = __builtin_fmaf(-f[0], px, f[3]);
    a = __builtin_fmaf(-f[1], py, a);
    a = __builtin_fmaf(-f[2], pz, a);
    const float r = a * __builtin_amdgcn_rcpf(b);
    const float hx = __builtin_fmaf(r, vx, px);
    const float hy = __builtin_fmaf(r, vy, py);
    const float hz = __builtin_fmaf(r, vz, pz);
    const float s = __builtin_fmaf(f[4], hx, __builtin_fmaf(f[5], hy, __builtin_fmaf(f[6], hz, f[7])));
    const float t = __builtin_fmaf(f[8], hx, __builtin_fmaf(f[9], hy, __builtin_fmaf(f[10], hz, f[11])));
    const float u = __builtin_fmaf(f[12], hx, __builtin_fmaf(f[13], hy, __builtin_fmaf(f[14], hz, f[15])));
    const float m = __builtin_fminf(__builtin_fminf(s, t), u);
    // Every comparison is false for NaN, so NaN never rejects.  A ray too parallel to the
    // plane (|b| below the reference's grazing threshold) needs no test of its own: the
    // reference then misses unless the origin lies on the plane, which the last term keeps.
    const bool reject = (m < 0.0f) | (r < fp.r_lo);
    return (!reject) | (fabsf(a) <= fp.a_zero);
}

template <int MODE, int RPL>
__global__ void __launch_bounds__(RT_BLOCK, RT_TRACE_MIN_WAVES)
k_trace(const DevScene* __restrict__ sc, const v16f* __restrict__ filt_global, const TriExact* __restrict__ exact,
        TraceArgs A) {
    // The filter records are never written while a kernel runs: reading them through the
    // constant address space lets the compiler use scalar loads (s_load_dwordx16).
    const cv16f* filt = (const cv16f*)filt_global;
    __shared__ unsigned short cand[RT_CAND_CAP * RT_BLOCK];
    const int lane = threadIdx.x & (RT_WAVE - 1);
    // threadIdx.x / 64 is the same in every lane of a wave; say so, so that everything
    // derived from it (task, triangle index) stays in scalar registers.
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / RT_WAVE);
    unsigned short* my_cand = cand + wave_in_block * RT_WAVE + lane;   // entry k at my_cand[k * RT_BLOCK]
    const int n_waves = gridDim.x * (RT_BLOCK / RT_WAVE);
    const int wave_id = blockIdx.x * (RT_BLOCK / RT_WAVE) + wave_in_block;
    constexpr int RAYS_PER_TASK = RT_WAVE * RPL;

    const int n_lights = sc->n_lights;
    // the queue length is the same for every lane: keep it (and all loop bounds derived from it) scalar
    const unsigned int n_hits = __builtin_amdgcn_readfirstlane(*A.n_items);
    const unsigned long long n_rays = (MODE == 0) ? n_hits : (unsigned long long)n_hits * n_lights;
    const unsigned int n_ray_blocks = (unsigned int)((n_rays + RAYS_PER_TASK - 1) / RAYS_PER_TASK);
    const int NR = sc->n_regular, NI = sc->n_irregular, NS = sc->n_spheres;
    // Split the triangle range over several waves when there are too few rays to fill the chip.
    int max_split = (NR + 2047) / 2048;
    if (max_split < 1) max_split = 1;
    int split = 1;
    if (n_ray_blocks > 0 && n_ray_blocks < 2u * n_waves) {
        split = (2 * n_waves + n_ray_blocks - 1) / n_ray_blocks;
        if (split > max_split) split = max_split;
    }
    const unsigned int n_tasks = n_ray_blocks * (unsigned int)split;   // < 2^31: n_ray_blocks * split <= ~4 * n_waves or n_ray_blocks
    const float eps = (MODE == 0) ? sc->eps : sc->shadow_eps;
    const FilterParams fp = (MODE == 0) ? sc->fp_closest : sc->fp_shadow;
    const float ob = sc->origin_bound, vb = sc->dir_len2_bound;
    const int n_groups = (MODE == 0) ? sc->n_groups : 1;
    unsigned int exact_total = 0;

    for (unsigned int task = wave_id; task < n_tasks; task += n_waves) {
        // integer division goes through the vector ALU; its result is still wave-uniform
        const int part = __builtin_amdgcn_readfirstlane((int)(task / n_ray_blocks));
        const unsigned int rb = task - (unsigned int)part * n_ray_blocks;
        const unsigned long long ray0 = (unsigned long long)rb * RAYS_PER_TASK + lane;   // slot r: ray0 + r * 64
        float best_d[RPL];
        int best_obj[RPL];
        bool done[RPL], invalid[RPL];     // done: slot needs no more tests (no ray, or already occluded)
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            best_d[r] = RT_MISS_T; best_obj[r] = -1;
            invalid[r] = ray0 + (unsigned long long)r * RT_WAVE >= n_rays;
            done[r] = invalid[r];
        }

        // ---- spheres and irregular triangles: always in reference precision (first part only) ----
        if (part == 0 && NS + NI > 0) {
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                if (done[r]) continue;
                d3 wpos, wvec;
                int source;
                load_world_ray<MODE>(A, sc, ray0 + (unsigned long long)r * RT_WAVE, n_lights, &wpos, &wvec, &source);
                bool occ = false;
                for (int s = 0; s < NS; s++) {
                    const SphereRec& S = sc->spheres[s];
                    const double* X = sc->xforms + 48 * (size_t)S.xf;
                    d3 pos = wpos, vec = wvec;
                    const int ident = (MODE == 0) ? S.identity : 1;   // shadow rays are never transformed (raytracer.cpp:553)
                    to_object_space(X, ident, &pos, &vec);
                    float t = sph_intersect_exact(S, pos, vec, eps);
                    fold_result<MODE>(t, S.obj, pos, vec, X, ident, source, &best_d[r], &best_obj[r], &occ);
                    exact_total++;
                }
                for (int g = 0; g < n_groups; g++) {
                    int ibeg = NR, iend = NR + NI, ident = 1;
                    const double* X = sc->xforms;
                    if (MODE == 0) {
                        const Group& G = sc->groups[g];
                        ibeg = G.ibeg; iend = G.iend; ident = G.identity;
                        X = sc->xforms + 48 * (size_t)G.xf;
                    }
                    d3 pos = wpos, vec = wvec;
                    to_object_space(X, ident, &pos, &vec);
                    for (int i = ibeg; i < iend; i++) {
                        float t = tri_intersect_exact(exact[i], pos, vec, eps);
                        fold_result<MODE>(t, exact[i].obj, pos, vec, X, ident, source, &best_d[r], &best_obj[r], &occ);
                        exact_total++;
                    }
                }
                if (MODE == 1 && occ) done[r] = true;
            }
        }

        // ---- regular triangles, group by group ----
        for (int g = 0; g < n_groups; g++) {
            int fbeg = 0, fend = NR, ident = 1;
            const double* X = sc->xforms;
            if (MODE == 0) {
                // group records are the same for every lane: keep the bounds in scalar registers
                const Group& G = sc->groups[g];
                fbeg = __builtin_amdgcn_readfirstlane(G.fbeg); fend = __builtin_amdgcn_readfirstlane(G.fend);
                ident = __builtin_amdgcn_readfirstlane(G.identity);
                X = sc->xforms + 48 * (size_t)__builtin_amdgcn_readfirstlane(G.xf);
            }
            // this wave's share of the group's regular triangles
            const int glen = fend - fbeg;
            const int per_part = __builtin_amdgcn_readfirstlane((glen + split - 1) / split);
            const int pbeg = fbeg + part * per_part;
            const int pend = min(fend, pbeg + per_part);
            if (pbeg >= pend) continue;

            // FP32 copies of the rays in this group's object space
            float px[RPL], py[RPL], pz[RPL], vx[RPL], vy[RPL], vz[RPL];
            bool force[RPL];              // test everything exactly for this slot
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                d3 pos = mk3(0, 0, 0), vec = mk3(0, 0, 1);
                int source;
                if (!invalid[r]) {
                    load_world_ray<MODE>(A, sc, ray0 + (unsigned long long)r * RT_WAVE, n_lights, &pos, &vec, &source);
                    to_object_space(X, ident, &pos, &vec);
                }
                px[r] = (float)pos.x; py[r] = (float)pos.y; pz[r] = (float)pos.z;
                vx[r] = (float)vec.x; vy[r] = (float)vec.y; vz[r] = (float)vec.z;
                // Outside the bounds the margins were derived for (or NaN): no filtering.
                const bool trusted = fabsf(px[r]) <= ob && fabsf(py[r]) <= ob && fabsf(pz[r]) <= ob &&
                                     (vx[r] * vx[r] + vy[r] * vy[r] + vz[r] * vz[r]) <= vb;
                force[r] = A.force_exact || !trusted;
            }

            for (int cbeg = pbeg; cbeg < pend; cbeg += RT_CHUNK) {
                const int cend = min(cbeg + RT_CHUNK, pend);
                int n_cand = 0;
                for (int j0 = cbeg; j0 < cend; j0 += RT_BATCH) {
                    const int jn = min(j0 + RT_BATCH, cend);
                    if (MODE == 1) {
                        bool all_done = true;
#pragma unroll
                        for (int r = 0; r < RPL; r++) all_done = all_done && done[r];
                        if (!__any(!all_done)) break;
                    }
                    // The record of triangle j+1 is requested before triangle j is evaluated
                    // (the array carries one padding record, so j+1 is always readable).
                    v16f cur = filt[j0];
#pragma unroll 2
                    for (int j = j0; j < jn; j++) {
                        const v16f nxt = filt[j + 1];
                        bool c[RPL], any = false;
#pragma unroll
                        for (int r = 0; r < RPL; r++) {
                            c[r] = (filter_keep(cur, px[r], py[r], pz[r], vx[r], vy[r], vz[r], fp) | force[r]) & !done[r];
                            any |= c[r];
                        }
                        // one (rarely taken) region per triangle for all of the lane's rays
                        if (any) {
                            // The entry is built from a SCALAR register here, inside the branch;
                            // otherwise the compiler moves the whole triangle counter (and with it
                            // the address of every record load) to the vector ALU.
                            unsigned int entry;
                            asm volatile("v_mov_b32 %0, %1" : "=v"(entry) : "s"(j - cbeg));
#pragma unroll
                            for (int r = 0; r < RPL; r++) {
                                if (c[r]) {
                                    my_cand[n_cand * RT_BLOCK] = (unsigned short)(entry | (r << RT_SLOT_SHIFT));
                                    n_cand++;
                                }
                            }
                        }
                        cur = nxt;
                    }
                    // Drain when some lane could overflow during the next batch, and at the end of the chunk.
                    const bool last = jn >= cend;
                    if (__any(n_cand > RT_CAND_CAP - RT_BATCH * RPL) || (last && __any(n_cand > 0))) {
                        unsigned int dmask = 0;
#pragma unroll
                        for (int r = 0; r < RPL; r++) dmask |= done[r] ? (1u << r) : 0u;
                        for (int k = 0; __any(k < n_cand); k++) {
                            if (k >= n_cand) continue;
                            const unsigned int e = my_cand[k * RT_BLOCK];
                            const int r = (int)(e >> RT_SLOT_SHIFT);
                            if ((dmask >> r) & 1u) continue;
                            const int j = cbeg + (int)(e & ((1u << RT_SLOT_SHIFT) - 1u));
                            // the ray in reference precision, re-read and re-transformed on demand
                            d3 pos, vec;
                            int source;
                            load_world_ray<MODE>(A, sc, ray0 + (unsigned long long)r * RT_WAVE, n_lights, &pos, &vec, &source);
                            to_object_space(X, ident, &pos, &vec);
                            const TriExact& T = exact[j];
                            const float t = tri_intersect_exact(T, pos, vec, eps);
                            exact_total++;
                            bool occ = false;
#pragma unroll
                            for (int rr = 0; rr < RPL; rr++)
                                if (rr == r) fold_result<MODE>(t, T.obj, pos, vec, X, ident, source, &best_d[rr], &best_obj[rr], &occ);
                            if (MODE == 1 && occ) dmask |= 1u << r;
                        }
#pragma unroll
                        for (int r = 0; r < RPL; r++) done[r] = (dmask >> r) & 1u;
                        n_cand = 0;
                    }
                }
            }
        }

        // ---- publish ----
#pragma unroll
        for (int r = 0; r < RPL; r++) {
            if (invalid[r]) continue;
            const unsigned long long ray = ray0 + (unsigned long long)r * RT_WAVE;
            if (MODE == 0) {
                if (best_obj[r] >= 0) atomicMin(&A.keys[ray], make_key(best_d[r], best_obj[r]));
            } else if (done[r]) {
                const unsigned int hit = (unsigned int)(ray / n_lights), light = (unsigned int)(ray % n_lights);
                atomicOr(&A.occluded[(size_t)hit * A.occl_words + (light >> 5)], 1u << (light & 31));
            }
        }
    }
    // one counter update per wave
    for (int off = 32; off > 0; off >>= 1) exact_total += __shfl_xor(exact_total, off);
    if (lane == 0 && exact_total) atomicAdd(&A.counters->exact_tests, (unsigned long long)exact_total);
}

// ---------------------------------------------------------------------------
// After k_closest: the part of Scene::traceRay behind the object loop
// (raytracer.cpp:531-538) for every ray that hit, compacted.
// ---------------------------------------------------------------------------
struct HitArgs {
    const DevScene* sc;
    RayQueue cur, next;                  // next[h] = (hit point, reflected direction)
    const unsigned long long* keys;
    unsigned long long* next_keys;
    const uint32_t* n_rays;
    uint32_t* n_hits;                    // zeroed by the host before the launch
    LevelStore lvl;
    HitAux aux;
    uint32_t* occluded;
    int32_t occl_words;
};

__global__ void __launch_bounds__(RT_BLOCK)
k_hit(HitArgs A) {
    const DevScene* __restrict__ sc = A.sc;
    const unsigned int n = *A.n_rays;
    const int lane = threadIdx.x & (RT_WAVE - 1);
    const unsigned int stride = gridDim.x * blockDim.x;
    const unsigned int n_round = (n + stride - 1) / stride * stride;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        unsigned long long key = (i < n) ? A.keys[i] : RT_MISS_KEY;
        const bool hit = key != RT_MISS_KEY;
        // wave-level compaction: one atomic per wave
        const unsigned long long ballot = __ballot(hit);
        unsigned int base = 0;
        if (lane == 0 && ballot) base = atomicAdd(A.n_hits, (unsigned int)__popcll(ballot));
        base = __shfl(base, 0);
        if (!hit) continue;
        const unsigned int h = base + (unsigned int)__popcll(ballot & ((1ull << lane) - 1ull));
        const int obj = (int)(unsigned int)(key & 0xffffffffu);
        const int xf = sc->obj_transform[obj];
        const double* X = sc->xforms + 48 * (size_t)xf;
        const int ident = sc->xf_identity[xf];
        d3 wpos = mk3(A.cur.ox[i], A.cur.oy[i], A.cur.oz[i]);
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
        A.next_keys[h] = RT_MISS_KEY;
        A.aux.nx[h] = nw.x; A.aux.ny[h] = nw.y; A.aux.nz[h] = nw.z;
        A.aux.vx[h] = rv.x; A.aux.vy[h] = rv.y; A.aux.vz[h] = rv.z;
        A.lvl.parent[h] = (int32_t)i;
        A.lvl.obj[h] = obj;
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
        A.lvl.child[h] = 0.0; A.lvl.child[h + (size_t)A.cap] = 0.0; A.lvl.child[h + 2 * (size_t)A.cap] = 0.0;
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

// Adds one pass's queue lengths to the running counters.
__global__ void k_count(const DevScene* __restrict__ sc, const uint32_t* __restrict__ counts, int levels,
                        DevCounters* __restrict__ c) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    unsigned long long prims = (unsigned long long)sc->n_regular + sc->n_irregular + sc->n_spheres;
    unsigned long long rays = 0, hits = 0;
    for (int k = 0; k < levels; k++) { rays += counts[k]; hits += counts[k + 1]; }
    c->rays_primary += counts[0];
    c->rays_reflect += rays - counts[0];
    c->rays_shadow += hits * (unsigned long long)sc->n_lights;
    c->hits += hits;
    c->closest_tests += rays * prims;
    c->anyhit_tests += hits * (unsigned long long)sc->n_lights * prims;
}
