// What a trace launch is given (TraceArgs), the executed-work counters of the scans, and the pieces every trace
// kernel shares: a ray of the launch in world space, its object space, the merge of a test result.
#pragma once

#include "rt_queue.h"

// ---------------------------------------------------------------------------
// Brute-force scan of every primitive for every ray.
// MODE 0: nearest hit (Scene::traceRay's object loop, raytracer.cpp:519-530).
// MODE 1: any hit (Scene::traceShadowRay, raytracer.cpp:551-558).
//
//   k_filter       FP32 test of every (ray, regular triangle) pair; pairs that cannot be
//                  ruled out go to a pair list in HBM (staged per lane in LDS).  This is the
//                  kernel that does the O(rays x triangles) work.
//   k_exact_pairs  the listed pairs again, in the reference's mixed double/float arithmetic;
//                  results are merged per ray with atomicMin / atomicOr.
//   k_brute        everything that never goes through the filter, in reference arithmetic:
//                  spheres, irregular triangles, and ALL triangles for rays flagged by
//                  k_filter (origin outside the bounds the filter margins were derived for,
//                  NaN, pair list full) or when the filter is switched off.
// ---------------------------------------------------------------------------
struct TraceArgs {
    RayQueue rays;                       // MODE 0: the level's rays.  MODE 1: origins = hit points
    int32_t beside_others;               // the launch runs beside the launches of other render calls (RT_OPT_FRAMES_IN_FLIGHT > 1)
    const uint32_t* n_items;             // MODE 0: number of rays; MODE 1: number of hits (shadow rays = hits * n_lights)
    unsigned long long* keys;            // MODE 0: per ray, atomicMin of (metric bits, object index)
    const int32_t* source_obj;           // MODE 1: object each hit lies on (excluded, raytracer.cpp:553)
    uint32_t* occluded;                  // MODE 1: per hit, one bit per light
    int32_t occl_words;                  // MODE 1: 32-bit words per hit
    int32_t scan_all;                    // k_brute: 1 = test every triangle for every ray (filter switched off)
    int32_t camera_rays;                 // MODE 0: 1 = the launch's rays are the pass's camera rays (origin = the eye)
    uint8_t* ray_flags;                  // per ray: 1 = k_brute must scan every triangle for it
    unsigned long long* pairs;           // (ray << 32) | exact-record slot
    // The pair list is cut into pair_lists (a power of two, at most RT_PAIR_LISTS) equal sub-lists of pair_seg
    // entries, each with a fill counter on a 128-byte line of its own (pair_count[16 * l]): a flush picks one by a
    // hash of its rays and its chunk.  (One list with one counter had every flush of every wave of a launch
    // queue at one address of one L2 channel: 0.26 ms of the 0.66 ms camera-ray scan of a 4K frame.)
    unsigned long long* pair_count;      // per sub-list: pairs appended so far (may run past pair_seg: the surplus is dropped)
    unsigned int pair_lists, pair_seg;
    unsigned int* task_counter;          // k_filter: next task to hand out (zeroed before the launch)
    unsigned int* any_flagged;           // set to 1 by whoever sets a ray flag (zeroed before the launch): lets k_brute
                                         // leave at once when no ray is flagged and the scene has nothing else for it
    // k_brute, the last kernel of a launch, prepares the launches behind it: prev_words = {task counter, "some ray was
    // flagged", number of rays} of the launch before this one, whose flags (prev_ray_flags) are cleared if it set any;
    // next_header: the header the next launch will use, cleared.  task_counter[2] receives this launch's number of rays.
    const unsigned int* prev_words;
    uint8_t* prev_ray_flags;
    unsigned long long* next_header;
    unsigned int pair_cap;
    DevCounters* counters;
    // Per regular-triangle constants that let k_filter skip work shared by all rays of a launch:
    const float* shared_origin_a;        // MODE 0, camera rays only (all start at the eye): d0 - n.eye per slot; else null
    const float* shared_dir_rinv;        // MODE 1: [row][n_regular + 1] of 1 / (n.dir) for directional lights; may be null
    const int32_t* light_dir_row;        // MODE 1: per light, its row in shared_dir_rinv or -1 (point light)
    // Bounding cones (see filter_range): per regular triangle (unit axis from the shared point to
    // the centre of the triangle's bounding sphere, cosine of the half-angle).
    const float* cone_camera;            // MODE 0, camera rays: [n_regular + 8][4]; else null
    // MODE 1, per light l (row l of each array; null = no bounding tests):
    const float* bound_lights;           // [n_lights][n_regular + 8][4]  point light: cone from the light (axis, cos);
                                         //   directional light: (-2 cu, -2 cv, rho^2 - cu^2 - cv^2, 0) in the light's plane
    const double* light_frame;           // [n_lights][8]  directional light: e1 (3), e2 (3), -O.e1, -O.e2
    // MODE 0, rays with nothing in common (null = no bounding test):
    const float* sphere_any;             // [n_regular + 8][4]  bounding sphere: centre, rho^2 - |centre|^2 (+ rounding slack)
    const float* plane_any;              // [n_regular + 8][4]  the triangle's plane (n, d0), a compact copy of the filter record's head
    // One level up: the regular triangles in clusters of 64 consecutive slots (every axis range of every
    // group starts at a multiple of 64), one record per cluster in the layout of the per-triangle table of
    // the same kind, around the smallest sphere found around the cluster's bounding spheres.
    const float* cl_camera;              // [n_clusters + 1][4]   like cone_camera
    const float* cl_lights;              // [n_lights][n_clusters + 1][4]   like bound_lights
    // Per light, the slots of the triangles whose plane a shadow ray towards that light can graze at all
    // (in ascending order): the light lies close to the plane (point light) or nearly in it (directional
    // light).  Only these can take the in-plane-origin branch of Triangle::intersection (raytracer.cpp:319).
    const int32_t* graze_slots;          // all lights' lists, one after the other
    const int32_t* graze_start;          // [n_lights][n_clusters + 1]: first entry of the light's list with slot >= 64 * cluster (index into graze_slots)
    const float* cl_sphere;              // [n_clusters + 1][4]   centre, radius
    const float* cl_normal;              // [n_clusters + 1][4]   unit mean normal N, sine of the largest angle between N and a member's normal
    int32_t n_clusters;
    // FP32 scan copies (see RayQueue::scan).  MODE 0: rays.scan may be used (every group's transform is the
    // identity).  MODE 1: per (light, hit) of the launch, [light * shadow_stride + hit]: point light -- unit
    // direction to the light and the distance (rounded up); directional light -- the origin in the light's
    // plane and the square of that point; null: none kept.  The origins are rays.scan's.
    int32_t scan_nearest;
    const float4* shadow_scan;
    uint32_t shadow_stride;
    unsigned long long* work;            // k_filter<..., WORK = true>: [RT_WORK_KINDS][RT_WORK_FIELDS] executed-work counters
    // 1: the bounded instances ask ONLY the in-plane-origin question (raytracer.cpp:319-321) -- no reach question at
    // any level -- and list the pairs it names: the complement of a box hierarchy, which finds every hit but that one
    // (a hit "at the ray's own origin" on a triangle whose box the ray never reaches).  Launched behind the LBVH
    // traversal; what it lists goes through k_exact_pairs like any other pair.
    int32_t in_plane_only;
    // 0: the in-plane-origin question is not asked at all -- nothing is named for it, no pair is listed for it.  Right only
    // when it is known that no ray of the frame takes that branch of Triangle::intersection (rt_ctx::plane_guard).
    int32_t plane_guard;
    // k_brute drops spheres for a whole wave at a time (see ray_cone_of_wave) in scenes with at least this many; 0: never.
    int32_t sphere_cull_min;
    // 0: k_brute asks the spheres with everything else; 1: this kernel asks them alone (k_spheres_first); 2: that kernel ran
    // ahead of this launch's others, which skip the spheres and the shadow rays it found something in the way of.
    int32_t spheres;
};

// Executed work of the scans, counted per wave in registers and added to TraceArgs::work at the
// end of every task (only in the WORK instances of k_filter; the others carry none of this).
// Kinds of scan: 0 nearest hit without bounding tests, 1 camera rays, 2 shadow rays towards a
// point light, 3 towards a directional light, 4 reflection rays, 5 shadow rays without
// bounding tests.
#define RT_WORK_KINDS 6
#define RT_WORK_FIELDS 11
enum {
    RT_WK_TASKS = 0,        // tasks (block of rays x part of the triangle range)
    RT_WK_GROUPS,           // groups of 64 triangles judged against the task's bundle(s) of rays
    RT_WK_GROUP_TESTS,      // ... (bundle, triangle) questions asked in them: one per lane with a triangle and per bundle
    RT_WK_GROUPS_LIVE,      // groups with a triangle some bundle can reach (or the plane guard names)
    RT_WK_BATCHES,          // batches of 4 triangles whose per-ray bounding tests were evaluated
    RT_WK_RAY_BOUND_TESTS,  // per-ray bounding tests executed in them (64 lanes x rays of the lane that took part x 4)
    RT_WK_BATCHES_TESTED,   // batches in which some lane asked for a triangle test
    RT_WK_TRI_TESTS,        // FP32 triangle tests executed: (wave, triangle) executions x 64 lanes x rays per lane
    RT_WK_TRI_WANTED,       // ... of which in lanes that had asked for the triangle (all of them without bounding tests)
    RT_WK_LISTED,           // pairs listed for the reference-precision re-test
    RT_WK_CLUSTERS          // clusters of 64 triangles judged (one lane each, one level above the groups)
};
struct WorkCount {
    unsigned int v[RT_WORK_FIELDS];
#ifdef RT_PHASE_CLOCK
    unsigned int t[8];      // shader cycles by phase of a task (see k_filter)
#endif
};
// Diagnostic build (-DRT_PHASE_CLOCK): the counting instances of the scan kernels report, in place of the
// executed-work counts, the shader cycles (s_memtime) a wave spends in each phase of its tasks, summed over
// waves: field 1 waiting for the task number, 2 loading the task's rays and building their bundles, 3 judging
// clusters and groups, 4 the batch loop (per-ray bounding tests, triangle tests) without 5 the flushes of the
// staged candidates, 6 whole tasks, 7 the waves' lifetimes; field 0 stays the number of tasks.
#ifdef RT_PHASE_CLOCK
#define RT_CLOCK() ((unsigned int)__builtin_readcyclecounter())
#endif

// Shadow rays are numbered light-major: ray = light * n_hits + hit, so that rays towards one
// light are neighbours (a block of rays towards a directional light shares its direction).
__device__ __forceinline__ void shadow_ray_index(unsigned int ray, unsigned int n_hits, unsigned int* hit, unsigned int* light) {
    const unsigned int l = ray / n_hits;
    *light = l;
    *hit = ray - l * n_hits;
}

// World-space ray number `ray` of a launch.  MODE 0: entry `ray` of the level's queue.
// MODE 1: shadow ray of one hit towards one light (Light::lightRay, raytracer.cpp:108-119);
// *source = the object the hit lies on.
template <int MODE>
__device__ __forceinline__ void load_world_ray(const TraceArgs& A, const DevScene* __restrict__ sc,
                                               unsigned int ray, unsigned int n_hits, d3* pos, d3* vec, int* source) {
    if (MODE == 0) {
        // (camera rays all start at the eye, which is what k_primary stored for each of them: not fetched)
        *pos = A.camera_rays ? ld3(sc->cam_eye) : mk3(A.rays.ox[ray], A.rays.oy[ray], A.rays.oz[ray]);
        *vec = mk3(A.rays.dx[ray], A.rays.dy[ray], A.rays.dz[ray]);
        *source = -1;
    } else {
        unsigned int hit, light;
        shadow_ray_index(ray, n_hits, &hit, &light);
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

// Merges one reference-precision test result into the ray's global answer.
// MODE 0: (metric, object) with the reference's strict '<' against MAXFLOAT and its
// lowest-index tie rule (raytracer.cpp:516, 525) -- the 64-bit key orders exactly that way.
// MODE 1: any hit other than the source (raytracer.cpp:553; a NaN result counts as a hit
// there because NaN != MAXFLOAT).
template <int MODE>
__device__ __forceinline__ void publish_result(const TraceArgs& A, unsigned int ray, unsigned int n_hits, float t,
                                               int obj, d3 pos, d3 vec, const double* X, int identity, int source) {
    if (MODE == 0) {
        d3 world;
        float d = hit_metric(X, identity, pos, vec, t, &world);
        if (t != RT_MISS_T && d < RT_MISS_T) atomicMin(&A.keys[ray], make_key(d, obj));
    } else {
        if (t != RT_MISS_T && obj != source) {
            unsigned int hit, light;
            shadow_ray_index(ray, n_hits, &hit, &light);
            atomicOr(&A.occluded[(size_t)hit * A.occl_words + (light >> 5)], 1u << (light & 31));
        }
    }
}

