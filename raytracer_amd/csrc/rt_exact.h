// The reference-precision kernels of a trace launch: the listed pairs (k_exact_pairs) and everything the filter
// does not cover (k_brute).
#pragma once

#include "rt_trace_args.h"

// One thread per listed (ray, triangle) pair.  What a thread does per pair is a chain of dependent loads with some
// double-precision arithmetic behind it, and the kernel goes as fast as it has such chains in flight: compiled for
// eight waves per SIMD (64 registers; at 72 and seven waves the camera rays' 17 M pairs of a 4K frame took 0.29 ms,
// now 0.22).
#ifndef RT_EXACT_MIN_WAVES
#define RT_EXACT_MIN_WAVES 8
#endif
template <int MODE>
__global__ void __launch_bounds__(RT_BLOCK, RT_EXACT_MIN_WAVES)
k_exact_pairs(const DevScene* __restrict__ sc, const TriExact* __restrict__ exact, TraceArgs A) {
    // the sub-lists one after the other: first[l] = entries in the sub-lists before l
    __shared__ unsigned int first[RT_PAIR_LISTS + 1];
    static_assert(RT_PAIR_LISTS == RT_WAVE, "one lane of the block's first wave per sub-list");
    if (threadIdx.x < RT_WAVE) {
        unsigned long long listed = 0;
        if (threadIdx.x < A.pair_lists) listed = A.pair_count[16 * threadIdx.x];
        unsigned int sum = listed < A.pair_seg ? (unsigned int)listed : A.pair_seg;
        for (int off = 1; off < RT_WAVE; off <<= 1) {           // inclusive prefix over the lanes
            const unsigned int up = (unsigned int)__shfl_up((int)sum, off);
            if ((int)threadIdx.x >= off) sum += up;
        }
        first[threadIdx.x + 1] = sum;
        if (threadIdx.x == 0) first[0] = 0;
    }
    __syncthreads();
    const unsigned int n = first[RT_PAIR_LISTS];
    const unsigned int n_lights = sc->n_lights;
    const unsigned int n_hits = *A.n_items;
    const unsigned int n_rays = (MODE == 0) ? n_hits : n_hits * n_lights;
    const unsigned int n_regular = (unsigned int)sc->n_regular;
    const float eps = (MODE == 0) ? sc->eps : sc->shadow_eps;
    // Entry i of the walk over the sub-lists.
    // (a thread's entries lie in ascending sub-lists: the search goes on from the last one)
    unsigned int l = 0;
    auto entry = [&](const unsigned int i) -> unsigned long long {
        while (first[l + 1] <= i) l++;
        return A.pairs[(size_t)l * A.pair_seg + (i - first[l])];
    };
    // The entry, then its ray and its record (and the record's group, if that has a transform): the next entry is
    // requested while this one is worked on.
    const unsigned int stride = gridDim.x * blockDim.x;
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long pr_next = (i < n) ? entry(i) : 0ull;
    for (; i < n; i += stride) {
        const unsigned long long pr = pr_next;
        if (i + stride < n) pr_next = entry(i + stride);
        const unsigned int ray = (unsigned int)(pr >> 32), slot = (unsigned int)pr;
        // Every entry below n was written by this launch's producers; the check only keeps a
        // producer bug from turning into an out-of-range access.
        if (ray >= n_rays || slot >= n_regular) continue;
        const TriExact& T = exact[slot];
        // (no scan lists a hole; asked only at the end, so that this load does not stand in front of the others)
        const bool is_hole = T.hole != 0;
        int ident = 1;
        const double* X = sc->xforms;
        if (MODE == 0 && !(T.grp & RT_GRP_IDENTITY)) {
            const Group& G = sc->groups[T.grp];
            ident = 0;
            X = sc->xforms + 48 * (size_t)G.xf;
        }
        d3 pos, vec;
        int source;
        load_world_ray<MODE>(A, sc, ray, n_hits, &pos, &vec, &source);
        to_object_space(X, ident, &pos, &vec);
        bool q7 = false;
        const float t = tri_intersect_exact(T, pos, vec, eps, &q7);
        if (is_hole) continue;
        if (q7) atomicAdd(&A.counters->in_plane_origin_hits, 1ull);
        publish_result<MODE>(A, ray, n_hits, t, T.obj, pos, vec, X, ident, source);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&A.counters->exact_tests, (unsigned long long)n);
}

// One ray per lane, every primitive the filter does not cover, in reference arithmetic.
template <int MODE>
__device__ __forceinline__ void brute_scan(const DevScene* __restrict__ sc, const TriExact* __restrict__ exact, const TraceArgs& A) {
    const int lane = threadIdx.x & (RT_WAVE - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / RT_WAVE);
    const unsigned int n_waves = gridDim.x * (RT_BLOCK / RT_WAVE);
    const unsigned int wave_id = blockIdx.x * (RT_BLOCK / RT_WAVE) + wave_in_block;
    const unsigned int n_lights = sc->n_lights;
    const unsigned int n_hits = __builtin_amdgcn_readfirstlane(*A.n_items);
    const unsigned int n_rays = (MODE == 0) ? n_hits : n_hits * n_lights;
    const unsigned int n_ray_blocks = (n_rays + RT_WAVE - 1) / RT_WAVE;
    if (n_ray_blocks == 0) return;
    // (TraceArgs::spheres: 1 = this launch asks the spheres alone, 2 = an earlier kernel of the launch has asked them)
    const int what = A.spheres;
    const int NR = sc->n_regular, NI = (what == 1) ? 0 : sc->n_irregular, NS = (what == 2) ? 0 : sc->n_spheres;
    const int n_void = (what == 1) ? 0 : sc->n_void_normal;
    // A triangle whose unit normal is not a number (zero area: 0 / 0 in raytracer.cpp:313) gives Triangle::intersection
    // the same run for every ray: a, b and r are not numbers, every comparison of raytracer.cpp:319-355 is false, and
    // it returns r.  As a nearest hit that never counts (its distance is not below MAXFLOAT, raytracer.cpp:525); for a
    // shadow ray it always does (NaN != MAXFLOAT, raytracer.cpp:553; the source of a shadow ray is a nearest hit, so
    // never such a triangle).  One such triangle in the scene therefore shadows every ray, and none is tested.
    const bool all_shadowed = MODE == 1 && n_void > 0;
    // objects that every ray takes to the reference's arithmetic (a shadow ray already in shadow takes none)
    const int n_each_ray = (all_shadowed && !A.scan_all) ? 0 : NS + NI - n_void;
    // nothing but flagged rays could need this kernel, and no ray is flagged: done (one scalar load)
    if (n_each_ray == 0 && !all_shadowed && !A.scan_all &&
        (what == 1 || __builtin_amdgcn_readfirstlane(*(const volatile unsigned int*)A.any_flagged) == 0u)) return;
    int max_split = (NR + 255) / 256;
    if (max_split < 1) max_split = 1;
    int split = 1;
    if (what != 1 && n_ray_blocks < 2u * n_waves) {
        split = (int)((2u * n_waves + n_ray_blocks - 1) / n_ray_blocks);
        if (split > max_split) split = max_split;
    }
    const unsigned int n_tasks = n_ray_blocks * (unsigned int)split;
    const float eps = (MODE == 0) ? sc->eps : sc->shadow_eps;
    const int n_groups = (what == 1) ? 0 : (MODE == 0) ? sc->n_groups : 1;
    unsigned int n_exact = 0, n_q7 = 0;       // (n_q7: pairs that took the in-plane-origin branch, raytracer.cpp:319-321)

    for (unsigned int task = wave_id; task < n_tasks; task += n_waves) {
        const int part = __builtin_amdgcn_readfirstlane((int)(task / n_ray_blocks));
        const unsigned int rb = task - (unsigned int)part * n_ray_blocks;
        const unsigned int ray = rb * RT_WAVE + lane;
        const bool valid = ray < n_rays;
        const bool scan = what != 1 && valid && (A.scan_all || A.ray_flags[ray] != 0);
        const bool any_scan = __any(scan);
        if (part == 0 && !A.scan_all && any_scan) {
            const unsigned long long m = __ballot(scan);
            if (lane == 0) atomicAdd(&A.counters->flagged_rays, (unsigned long long)__popcll(m));
        }
        if (all_shadowed && part == 0 && valid) {
            unsigned int hit, light;
            shadow_ray_index(ray, n_hits, &hit, &light);
            atomicOr(&A.occluded[(size_t)hit * A.occl_words + (light >> 5)], 1u << (light & 31));
        }
        // A shadow ray that the launch's earlier kernels (or an earlier object of this loop) found something in the way of
        // asks no further object: traceShadowRay returns at the first one (raytracer.cpp:553-555).  (A bit another wave
        // sets meanwhile may be missed; the ray then asks an object too many.)
        bool live = valid;
        if (MODE == 1 && !A.scan_all && part == 0 && valid && n_each_ray > 0) {
            unsigned int hit, light;
            shadow_ray_index(ray, n_hits, &hit, &light);
            live = ((A.occluded[(size_t)hit * A.occl_words + (light >> 5)] >> (light & 31)) & 1u) == 0u;
        }
        if (!any_scan && (part != 0 || n_each_ray == 0 || !__any(live))) continue;
        d3 wpos = mk3(0, 0, 0), wvec = mk3(0, 0, 1);
        int source = -1;
        if (live || scan) load_world_ray<MODE>(A, sc, ray, n_hits, &wpos, &wvec, &source);
        if (part == 0 && NS > 0) {
            // an FP32 pre-test decides most (ray, sphere) pairs; the rest take the reference's arithmetic
            const float fpx = (float)wpos.x, fpy = (float)wpos.y, fpz = (float)wpos.z;
            const float fvx = (float)wvec.x, fvy = (float)wvec.y, fvz = (float)wvec.z;
            const float p1 = fabsf(fpx) + fabsf(fpy) + fabsf(fpz), v1 = fabsf(fvx) + fabsf(fvy) + fabsf(fvz);
            // With more than a few spheres the wave first drops those none of its rays can meet: the rays of a wave are
            // neighbours in the frame (or the shadow rays of neighbours towards one light) and share a narrow cone; sphere
            // s is judged by lane s, 64 at a time.
            const bool cull = A.sphere_cull_min > 0 && NS >= A.sphere_cull_min && !A.scan_all;
            RayCone cone;
            if (cull) cone = ray_cone_of_wave(live, fpx, fpy, fpz, fvx, fvy, fvz, p1);
            for (int base = 0; base < NS; base += RT_WAVE) {
                unsigned long long todo = ~0ull;
                if (cull && cone.usable) {
                    bool keep = false;
                    if (base + lane < NS) {
                        const SphereFilt& F = sc->sphere_filt[base + lane];
                        // (shadow rays meet the untransformed sphere, raytracer.cpp:553)
                        keep = (MODE == 0) ? ball_in_reach(cone, F.wc[0], F.wc[1], F.wc[2], F.wr, eps >= 0.0f)
                                           : ball_in_reach(cone, F.c[0], F.c[1], F.c[2], sqrtf(F.r2) * 1.000001f, eps >= 0.0f);
                    }
                    todo = __ballot(keep);
                }
                if (NS - base < RT_WAVE) todo &= (1ull << (NS - base)) - 1ull;
#ifdef RT_CHECK_SPHERE_CULL
                // Diagnostic build (brute-force mode, where the LBVH's two counters are idle): every dropped (ray, sphere)
                // pair is counted in bvh_leaf_tests and takes the reference's arithmetic all the same; one that hits is
                // counted in bvh_node_tests, which must stay at zero.
                {
                    unsigned long long dropped = ~todo;
                    if (NS - base < RT_WAVE) dropped &= (1ull << (NS - base)) - 1ull;
                    unsigned int n_dropped = 0, n_wrong = 0;
                    while (dropped != 0ull) {
                        const int s = base + (int)__builtin_ctzll(dropped);
                        dropped &= dropped - 1ull;
                        if (!live) continue;
                        const SphereRec& S = sc->spheres[s];
                        d3 pos = wpos, vec = wvec;
                        to_object_space(sc->xforms + 48 * (size_t)S.xf, (MODE == 0) ? S.identity : 1, &pos, &vec);
                        n_dropped++;
                        if (sph_intersect_exact(S, pos, vec, eps) != RT_MISS_T) n_wrong++;
                    }
                    if (n_dropped) atomicAdd(&A.counters->bvh_leaf_tests, (unsigned long long)n_dropped);
                    if (n_wrong) atomicAdd(&A.counters->bvh_node_tests, (unsigned long long)n_wrong);
                }
#endif
                while (todo != 0ull) {
                    const int s = base + (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    if (MODE == 1 && !__any(live)) break;
                    if (!live) continue;
                    const SphereFilt& F = sc->sphere_filt[s];
                    if (!A.scan_all && !sphere_may_hit(F, MODE == 0 && !F.identity, fpx, fpy, fpz, fvx, fvy, fvz, p1, v1, eps)) continue;
                    const SphereRec& S = sc->spheres[s];
                    const double* X = sc->xforms + 48 * (size_t)S.xf;
                    d3 pos = wpos, vec = wvec;
                    const int ident = (MODE == 0) ? S.identity : 1;   // shadow rays are never transformed (raytracer.cpp:553)
                    to_object_space(X, ident, &pos, &vec);
                    const float t = sph_intersect_exact(S, pos, vec, eps);
                    publish_result<MODE>(A, ray, n_hits, t, S.obj, pos, vec, X, ident, source);
                    n_exact++;
                    if (MODE == 1 && !A.scan_all && t != RT_MISS_T && S.obj != source) live = false;
                }
            }
        }
        for (int g = 0; g < n_groups; g++) {
            int fbeg = 0, fend = NR, ibeg = NR, iend = NR + NI, ident = 1;
            const double* X = sc->xforms;
            if (MODE == 0) {
                const Group& G = sc->groups[g];
                fbeg = __builtin_amdgcn_readfirstlane(G.fax[0]); fend = __builtin_amdgcn_readfirstlane(G.fax[3]);
                ibeg = __builtin_amdgcn_readfirstlane(G.ibeg); iend = __builtin_amdgcn_readfirstlane(G.iend);
                ident = __builtin_amdgcn_readfirstlane(G.identity);
                X = sc->xforms + 48 * (size_t)__builtin_amdgcn_readfirstlane(G.xf);
            }
            d3 pos = wpos, vec = wvec;
            to_object_space(X, ident, &pos, &vec);
            if (part == 0 && live) {
                for (int i = ibeg; i < iend; i++) {
                    if (exact[i].hole == 2) continue;             // (normal not a number: answered for all rays at once, above)
                    bool q7 = false;
                    const float t = tri_intersect_exact(exact[i], pos, vec, eps, &q7);
                    if (q7) n_q7++;
                    publish_result<MODE>(A, ray, n_hits, t, exact[i].obj, pos, vec, X, ident, source);
                    n_exact++;
                    if (MODE == 1 && !A.scan_all && t != RT_MISS_T && exact[i].obj != source) break;
                }
            }
            if (any_scan) {
                const int glen = fend - fbeg;
                const int per_part = __builtin_amdgcn_readfirstlane((glen + split - 1) / split);
                const int pbeg = fbeg + part * per_part;
                const int pend = min(fend, pbeg + per_part);
                // The whole wave works on one flagged ray at a time, 64 triangles per step: a
                // handful of flagged rays must not turn into a long single-lane tail.
                unsigned long long todo = __ballot(scan);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const d3 bpos = mk3(__shfl(pos.x, src), __shfl(pos.y, src), __shfl(pos.z, src));
                    const d3 bvec = mk3(__shfl(vec.x, src), __shfl(vec.y, src), __shfl(vec.z, src));
                    const unsigned int bray = (unsigned int)__shfl((int)ray, src);
                    const int bsource = __shfl(source, src);
                    for (int j = pbeg + lane; j < pend; j += RT_WAVE) {
                        if (exact[j].hole) continue;
                        bool q7 = false;
                        const float t = tri_intersect_exact(exact[j], bpos, bvec, eps, &q7);
                        if (q7) n_q7++;
                        publish_result<MODE>(A, bray, n_hits, t, exact[j].obj, bpos, bvec, X, ident, bsource);
                        n_exact++;
                    }
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { n_exact += __shfl_xor(n_exact, off); n_q7 += __shfl_xor(n_q7, off); }
    if (lane == 0 && n_exact) atomicAdd(&A.counters->exact_tests, (unsigned long long)n_exact);
    if (lane == 0 && n_q7) atomicAdd(&A.counters->in_plane_origin_hits, (unsigned long long)n_q7);
}

// The last kernel of a trace launch: brute_scan, and the housekeeping that spares every launch a fill in front of it.
// Launches of one kind follow each other on a stream and take three headers and two flag arrays in turn; this launch
// clears the flags of the launch before it, if that one set any (all blocks, a slice each), and the header of the launch
// after it (block 0) -- memory that no kernel of this launch or of a launch still running touches otherwise.
// The spheres alone, ahead of the launch's other kernels (TraceArgs::spheres = 1): a shadow ray that a sphere is in the
// way of then asks no triangle (traceShadowRay returns at the first object in the way, raytracer.cpp:553-555).
template <int MODE>
__global__ void __launch_bounds__(RT_BLOCK)
k_spheres_first(const DevScene* __restrict__ sc, const TriExact* __restrict__ exact, TraceArgs A) {
    brute_scan<MODE>(sc, exact, A);
}

template <int MODE>
__global__ void __launch_bounds__(RT_BLOCK)
k_brute(const DevScene* __restrict__ sc, const TriExact* __restrict__ exact, TraceArgs A) {
    brute_scan<MODE>(sc, exact, A);
    if (__builtin_amdgcn_readfirstlane(A.prev_words[1]) != 0u) {
        const size_t n_prev = A.prev_words[2];
        uint4* wide = (uint4*)A.prev_ray_flags;                     // (a flag array starts at a multiple of 256 bytes and ends at one)
        for (size_t k = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < (n_prev + 15) / 16; k += (size_t)gridDim.x * RT_BLOCK)
            wide[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (blockIdx.x == 0) {
        for (unsigned int k = threadIdx.x; k < RT_FLAGS_HEADER / 8; k += RT_BLOCK) A.next_header[k] = 0ull;
        if (threadIdx.x == 0) {
            const unsigned int n_hits = *A.n_items;
            A.task_counter[2] = (MODE == 0) ? n_hits : n_hits * (unsigned int)sc->n_lights;
        }
    }
}
