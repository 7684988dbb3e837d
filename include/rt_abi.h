/* rt_abi.h -- C ABI of the MI355X trace loop.
 *
 * The library behind this header (raytracer_amd/csrc -> libraytracer_amd.so)
 * replaces exactly one call of the reference program: `scene->render()` in
 * main (raytracer.cpp:857), i.e. Scene::render / traceRay / traceShadowRay /
 * phongShade (raytracer.cpp:497-581) together with the primitive methods they
 * reach (raytracer.cpp:251-272, 307-373, 389-419), Camera::generateRay
 * (raytracer.cpp:441-447), the Light methods (raytracer.cpp:108-128) and the
 * pixel quantisation of Film::writeToFilm (raytracer.cpp:174-203).  Parsing
 * stays on the host and so does PNG encoding.
 *
 * Everything crossing the boundary is plain C: pointers, sizes, ints.  All
 * calls return 0 on success or a negative rt_status; rt_last_error() gives
 * the message of the calling thread's last failure.  The library never
 * exits or aborts the process and has no CPU fallback: with no usable GPU
 * rt_upload_scene fails with RT_ERR_DEVICE.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,     /* bad argument / malformed scene description */
    RT_ERR_DEVICE = -2,  /* a HIP call failed (no GPU, out of memory, launch failure) */
    RT_ERR_RANGE = -3,   /* tile or buffer outside the frame */
    RT_ERR_STATE = -4    /* call not valid in the current state */
} rt_status;

/* Material (raytracer.cpp:44-64); sp holds the reference's float exactly. */
typedef struct rt_material {
    double ka[3], kd[3], ks[3], kr[3];
    double sp;
} rt_material;

/* Polygon::transform / inverseTransform / inverseTranspose (raytracer.cpp:241-250),
 * row-major.  The caller computes inv and inv_t with the reference's own
 * mat4::inverse()/transpose() (algebra3.h:1130, 1123) so the bits agree. */
typedef struct rt_transform {
    double m[16], inv[16], inv_t[16];   /* m and inv must be affine: last row 0 0 0 1 (else RT_ERR_ARG) */
} rt_transform;

/* Light (raytracer.cpp:86-106).  For a directional light `pos` is the
 * direction the light travels in (lightVec = -pos, raytracer.cpp:112). */
typedef struct rt_light {
    double color[3], pos[3];
    int32_t falloff;   /* 0, 1 or 2: intensity = color / dist^falloff (raytracer.cpp:121-128) */
    int32_t is_point;
} rt_light;

/* Flat description of a parsed Scene (raytracer.cpp:457-466).  Objects keep
 * the order of Scene::objects: nearest-hit ties go to the lowest index
 * (raytracer.cpp:525) and the shading ray is the LAST object's transformed
 * ray (raytracer.cpp:518-520, 536, 571), so the order is part of the input.
 * All pointers are host pointers owned by the caller; rt_upload_scene copies.
 */
typedef struct rt_scene_desc {
    uint32_t struct_size;      /* = sizeof(rt_scene_desc) */
    int32_t width, height;     /* Scene::width/height (raytracer.cpp:465) */
    int32_t max_depth;         /* Scene::maxDepth: up to max_depth+1 levels (raytracer.cpp:477, 539) */
    float eps;                 /* Scene::eps, 0.05f in main (raytracer.cpp:849) */
    float shadow_eps;          /* 0.1f in intersects() (raytracer.cpp:308, 390) */
    double cam_eye[3], cam_ll[3], cam_lr[3], cam_ul[3], cam_ur[3]; /* Camera (raytracer.cpp:425) */
    double ambient[3];         /* Scene::ambientLight.color (raytracer.cpp:463) */
    int32_t n_objects, n_materials, n_transforms, n_lights;
    const int32_t* obj_kind;      /* [n_objects] 0 = Triangle, 1 = Sphere */
    const int32_t* obj_material;  /* [n_objects] index into materials */
    const int32_t* obj_transform; /* [n_objects] index into transforms */
    const int32_t* obj_interp;    /* [n_objects] Triangle::needInterpolate (raytracer.cpp:281) */
    const double* obj_geom;       /* [n_objects][9] triangle: vertex1,vertex2,vertex3;
                                     sphere: center[3], radius (a float value), 5 unused */
    const double* obj_normals;    /* [n_objects][9] normal1..3 where obj_interp != 0; may be NULL
                                     when no object interpolates */
    const rt_material* materials;
    const rt_transform* transforms;
    const rt_light* lights;
} rt_scene_desc;

/* Options for rt_set_option. */
typedef enum rt_option {
    RT_OPT_EXACT_ONLY = 1,   /* 1: skip the FP32 candidate filter, test every primitive in
                                reference precision (slow; used to validate the filter) */
    RT_OPT_PROFILE = 2,      /* 1: bracket every trace-kernel launch with HIP events; the sums
                                appear in rt_stats */
    RT_OPT_PASS_PIXELS = 3,  /* pixels traced per internal pass (bounds the workspace) */
    RT_OPT_USE_BVH = 4,      /* 1: build (once) and use the LBVH in place of the brute-force triangle scan.  Same
                                frames, same nearest hits.  The in-plane-origin branch of Triangle::intersection
                                (raytracer.cpp:319-321: a ray that starts exactly in a triangle's plane and grazes it
                                "hits" at its own origin wherever the triangle is), which no box hierarchy can find, is
                                covered by a scan of its own behind the traversals where the frame needs it
                                (RT_OPT_IN_PLANE_ORIGIN).  Blocks (build, and the calibration frame of that option if it is still due).
                                The traversal walks four-wide nodes (each binary node with its internal children
                                replaced by theirs); 2: walk the binary tree itself (same frames, ~15 % slower) */
    RT_OPT_FILTER_FORM = 5,  /* form of the FP32 candidate filter: 0 = chosen from the scene's own numbers
                                (default; chosen again when RT_OPT_BOUNDING_TESTS changes: with bounding tests few
                                pairs reach the triangle test and per-test margins pay as soon as the static ones
                                are a few per cent of a triangle's width), 1 = static worst-case margins (cheapest
                                test), 2 = per-test margins (one more instruction per test; fewer pairs listed for
                                the reference-precision re-test).  Frames are identical in every form; blocks until
                                queued work has finished */
    RT_OPT_SHARED_CONSTANTS = 6, /* 1 (default): the brute-force filter reads host-computed per-triangle constants
                                where all rays of a launch share their origin (camera rays) or their direction
                                (shadow rays towards a directional light); 0: every ray is treated alike */
    RT_OPT_OVERLAP = 7,      /* 1 (default): a level's shadow-ray launches run on an internal stream beside the
                                next level's nearest-hit launches (joined back into the caller's stream before
                                the call's last kernel); 0: one stream.  Off while RT_OPT_PROFILE is on */
    RT_OPT_BOUNDING_TESTS = 8, /* 1 (default): the brute-force scan first asks whether any ray of a lane can reach
                                the triangle's bounding sphere and runs the triangle test only then: 2-3
                                multiply-adds per ray where all rays of a launch pass through one point (camera
                                rays: the eye; shadow rays: a point light) or are parallel (a directional
                                light) -- these need RT_OPT_SHARED_CONSTANTS --, 12 for rays with nothing in
                                common (reflection rays); 2: only the former; 0: every pair takes the triangle
                                test.  Frames are identical in all three */
    RT_OPT_WORK_COUNTERS = 9, /* 1: the brute-force scans count the work they execute (rt_stats.scan_work) in
                                 instances of the scan kernels compiled for it; 0 (default): the scans carry no
                                 counting code.  Frames are identical */
    RT_OPT_SCAN_COPIES = 10,  /* 1: the kernels that make a ray also write the FP32 values the bounded scans hold
                                 for it (36 bytes per ray, 16 per shadow ray for up to 8 lights), and the scans
                                 load those instead of converting the ray's doubles again for every part of the
                                 triangle range; 0 (default): the scans convert.  Same bits either way: frames,
                                 pair lists and counters are identical.  Measured on the 4K teapot frame: the
                                 scans 1-3 % faster, the frame 1.5-2.5 % slower (the extra stores) -- hence off */
    RT_OPT_IN_PLANE_ORIGIN = 12, /* Triangle::intersection (raytracer.cpp:319-321) lets a ray that starts exactly in a
                                 triangle's plane and grazes it "hit" that triangle at the ray's own origin, wherever in
                                 the plane the triangle lies.  Asking that question of every ray is what the bounded
                                 scans keep their naming machinery for (camera rays: the triangles in whose plane the
                                 eye may lie; shadow rays: the triangles a ray towards the light can graze; reflection
                                 rays: clusters of triangles whose planes can hold a ray's origin and a grazing
                                 direction), and in LBVH mode a scan of that kind behind every traversal (no box
                                 hierarchy finds such a hit); it costs a fifth of the benchmark frame by brute force and
                                 doubles its LBVH frame, and almost no scene has a ray that takes the branch.
                                 2 (default): decided once per context -- camera, frame size and depth belong to the
                                 scene, so every render call traces the same rays: the first render call (or
                                 RT_OPT_USE_BVH) first traces one whole frame with the question asked everywhere
                                 (blocking; into the context's own frame buffer, its counts stay out of the
                                 statistics), and the question is asked from then on iff some pair of that frame took
                                 the branch.  Frames equal the reference's either way.  Until that frame has been
                                 traced the question is asked (rt_render_tile_noalloc and rt_primary_hits never
                                 start it).  1: always asked.  0: never -- frames then differ from the reference
                                 exactly where a ray takes that branch (in LBVH mode: on a triangle whose box it
                                 does not reach) */
    RT_OPT_BVH_PLANE_GUARD = 12, /* (earlier name of RT_OPT_IN_PLANE_ORIGIN) */
    RT_OPT_FRAMES_IN_FLIGHT = 11 /* 1 (default): a render call's work runs on the caller's stream (and the internal
                                 stream of RT_OPT_OVERLAP, joined back before the call's last kernel).  2..8: that
                                 many render calls may be in flight at once: call i runs on internal streams of its
                                 own with workspace i mod N, behind the work that was on the caller's stream at the
                                 time of the call, and is NOT joined back -- rt_join does that.  The short launches
                                 at the deep levels of one frame then run beside the large launches of the next
                                 (4K teapot frame: 6.0 -> 4.7 ms per frame at 4; one rank's eighth of it: 2.0 ->
                                 1.1 ms).  Pixels are identical; calls that write the same pixels land in call
                                 order.  N workspaces are held, and every call in flight gets a hardware queue of
                                 its own (found out with a ~10 ms probe the first time the option is used); the
                                 caller's stream shares one unless the process has more queues than calls in
                                 flight (GPU_MAX_HW_QUEUES, 4 by default, read by HIP at its first call) -- what
                                 is enqueued on it then waits behind that call's kernels.  RT_OPT_PROFILE and RT_OPT_WORK_COUNTERS renders
                                 run as with 1.  Blocks until queued work has finished */
} rt_option;

/* Executed work of the brute-force scans of one kind of ray (RT_OPT_WORK_COUNTERS).  A scan hands
 * out tasks (a block of 256 or 512 rays x a part of the triangle range); a task judges clusters
 * of 64 triangles against what bounds its rays together, then the triangles of the clusters that
 * pass (a "group": one triangle per lane), evaluates per-ray bounding tests for the batches of 4
 * triangles that survive, and runs the FP32 triangle test where a lane asks for it. */
typedef struct rt_scan_work {
    uint64_t tasks;
    uint64_t groups;            /* groups of 64 triangles judged */
    uint64_t group_tests;       /* (bundle of rays, triangle) bounding questions asked in them */
    uint64_t groups_live;       /* groups with a triangle that a bundle can reach */
    uint64_t batches;           /* batches of 4 triangles whose per-ray bounding tests were evaluated */
    uint64_t ray_bound_tests;   /* per-ray bounding tests executed (whole waves: 64 lanes x rays per lane x 4) */
    uint64_t batches_tested;    /* batches in which some lane asked for a triangle test */
    uint64_t tri_tests;         /* FP32 triangle tests executed (whole waves: 64 lanes x rays per lane) */
    uint64_t tri_tests_wanted;  /* ... of which in lanes that had asked for the triangle */
    uint64_t listed;            /* pairs listed for the reference-precision re-test */
    uint64_t clusters;          /* clusters of 64 triangles judged, one level above the groups */
} rt_scan_work;

enum { RT_SCAN_PLAIN_NEAREST = 0,  /* nearest hit, every pair takes the triangle test */
       RT_SCAN_CAMERA = 1, RT_SCAN_SHADOW_POINT = 2, RT_SCAN_SHADOW_DIRECTIONAL = 3, RT_SCAN_REFLECTION = 4,
       RT_SCAN_PLAIN_SHADOW = 5, RT_SCAN_KINDS = 6 };

/* Counters since rt_upload_scene or the last rt_reset_stats. */
typedef struct rt_stats {
    uint64_t rays_primary;   /* traceRay calls from render() (raytracer.cpp:504) */
    uint64_t rays_reflect;   /* recursive traceRay calls (raytracer.cpp:543) */
    uint64_t rays_shadow;    /* traceShadowRay calls (raytracer.cpp:567) */
    uint64_t hits;           /* traceRay calls that found an object */
    uint64_t closest_tests;  /* ray x primitive pairs scanned for nearest hits */
    uint64_t anyhit_tests;   /* ray x primitive pairs scanned for shadow rays */
    uint64_t exact_tests;    /* pairs evaluated in reference precision */
    uint64_t bvh_node_tests; /* RT_OPT_USE_BVH: ray-box tests executed */
    uint64_t bvh_leaf_tests; /* RT_OPT_USE_BVH: triangles reached at the leaves */
    uint64_t in_plane_origin_hits; /* listed pairs that took the a == 0 branch of Triangle::intersection
                                      (raytracer.cpp:319-321), the one case a box hierarchy cannot find */
    uint64_t flagged_rays;   /* rays that bypassed the filter / hierarchy and were scanned in full */
    double closest_ms;       /* RT_OPT_PROFILE: summed duration of the nearest-hit filter kernel launches */
    double anyhit_ms;        /* RT_OPT_PROFILE: summed duration of the shadow filter kernel launches */
    double exact_ms;         /* RT_OPT_PROFILE: summed duration of the reference-precision kernels */
    uint32_t closest_launches, anyhit_launches, exact_launches;
    int32_t n_triangles, n_spheres, n_groups, n_irregular;
    int32_t filter_form;     /* 1 = static margins, 2 = per-test margins (see RT_OPT_FILTER_FORM) */
    double camera_ms;        /* RT_OPT_PROFILE: the part of closest_ms spent in launches of camera rays with
                                RT_OPT_BOUNDING_TESTS on (the others scan rays that have nothing in common) */
    uint32_t camera_launches;
    uint32_t bvh_plane_guard; /* 1: the in-plane-origin question is asked (RT_OPT_IN_PLANE_ORIGIN; in LBVH mode: the scan
                                 behind the traversals runs) */
    rt_scan_work scan_work[RT_SCAN_KINDS];   /* RT_OPT_WORK_COUNTERS */
    double scan_ms[RT_SCAN_KINDS];           /* RT_OPT_PROFILE: summed duration of the scan launches by kind (a shadow
                                                launch holds both kinds of light: booked under the kind of light 0) */
    uint32_t scan_launches[RT_SCAN_KINDS];
    /* RT_OPT_FRAMES_IN_FLIGHT > 1, once a render call has set the calls' streams up: how they were dealt onto the
       process's hardware queues (found by running short kernels on pairs of streams and comparing the device clock
       intervals they note: streams on one queue run one after the other). */
    uint32_t lanes;                       /* calls that can be in flight (0: none set up) */
    uint32_t lane_hw_queues_found;        /* different hardware queues found among the streams tried */
    int32_t lanes_share_a_queue;          /* 1: two of the calls' streams sit on one queue (they do not overlap) */
    int32_t caller_stream_shares_a_queue; /* 1: the caller's stream sits on a queue a call in flight uses (what is enqueued
                                             on it waits behind that call's kernels); 0: it has a queue to itself;
                                             -1: not known (no lanes yet, or the stream was capturing) */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* Copy the scene to the current HIP device, precompute the per-primitive
 * constants there and allocate the frame (width*height RGBA8) and the
 * workspace.  Replaces nothing in the reference: it is the hand-over of the
 * object graph built by parseSceneFile (raytracer.cpp:705) that
 * Scene::render reads in place (raytracer.cpp:497). */
int rt_upload_scene(rt_ctx** out, const rt_scene_desc* desc);

int rt_set_option(rt_ctx* ctx, int option, int64_t value);

/* Trace film rows [y0, y0+rows) -- y = 0 is the BOTTOM scanline, as in
 * Sampler::generateSample / Camera::generateRay (raytracer.cpp:153, 441) --
 * and store the quantised pixels (raytracer.cpp:174-203) into the frame.
 * Asynchronous: work is enqueued on `stream` (a hipStream_t; NULL = the
 * default stream).  Replaces the body of Scene::render for those rows
 * (raytracer.cpp:502-511).  Calls on one ctx must not overlap. */
int rt_render_tile(rt_ctx* ctx, int y0, int rows, void* stream);

/* The interleaved form: tiles k = first_tile, first_tile+tile_step, ... where
 * tile k is film rows [k*tile_rows, min((k+1)*tile_rows, height)); one rank of
 * an N-GPU job calls it with first_tile = rank, tile_step = N. */
int rt_render_tiles(rt_ctx* ctx, int first_tile, int tile_rows, int tile_step, void* stream);

/* With RT_OPT_FRAMES_IN_FLIGHT > 1 (a no-op otherwise): makes `stream` wait -- on the device, the
 * call does not block the host -- for every render call of this ctx issued so far except the
 * `keep_latest` most recent ones (0: all of them).  What is enqueued on `stream` afterwards sees
 * those calls' pixels; a caller that pipelines frames binds one frame buffer per call in flight
 * (rt_bind_frame takes effect for the calls after it) and hands frame i on while frame i+1 is
 * traced.  rt_download, rt_get_stats, rt_reset_stats and rt_free wait for everything by
 * themselves.  No counterpart in the reference, whose render() returns when the film is full
 * (raytracer.cpp:497-511). */
int rt_join(rt_ctx* ctx, void* stream, int keep_latest);

/* rt_render_tile for callers that cannot allocate device memory at the time of the call (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the same
 * kernels, with the workspace (ray queues, keys, per-level hit records, pair lists) taken from a
 * static array of 64 MiB that the library holds on every GPU it runs on from load to process
 * exit, so the call itself performs no hipMalloc / hipFree, creates no stream and no event, and
 * runs every launch on `stream`.  No counterpart in the reference.  The tile is traced in one
 * pass and must fit the array: rows * width * (about 0.4 KB + 56 B per level + 0.1 KB per light);
 * a larger tile fails with RT_ERR_RANGE and a message that names the largest row count that
 * fits -- the caller splits the frame.  Results are identical to rt_render_tile's.  All calls on
 * one GPU share that GPU's array (also across contexts): calls that overlap in time race. */
int rt_render_tile_noalloc(rt_ctx* ctx, int y0, int rows, void* stream);

/* Use caller-owned device memory (width*height*4 bytes) as the frame, e.g. a
 * tensor that a collective will read.  NULL returns to the internal frame. */
int rt_bind_frame(rt_ctx* ctx, void* device_rgba, size_t bytes);

/* Wait for the ctx's work and copy the whole frame to host memory in the
 * layout Film::writeFile hands to lodepng::encode (raytracer.cpp:212-231):
 * RGBA8, alpha 255, row 0 = TOP scanline (film row height-1).  Rows never
 * rendered are black. */
int rt_download(rt_ctx* ctx, uint8_t* rgba, size_t bytes);

/* Known-answer hook for parity tests: the nearest hit of the camera ray of every pixel of film
 * rows [y0, y0 + rows), row-major from row y0 (the object loop of Scene::traceRay,
 * raytracer.cpp:516-529, and nothing after it).  keys[i] = (bit pattern of the float metric d
 * << 32) | index of the object in Scene::objects order; a miss is 0x7f7fffffffffffff.  Honours
 * the options (filter form, LBVH, exact-only).  Blocks; runs on the default stream; leaves those
 * rows of the frame black. */
int rt_primary_hits(rt_ctx* ctx, int y0, int rows, uint64_t* keys, size_t count);

/* Known-answer hook beyond the camera rays, for every pixel of film rows [y0, y0 + rows), row-major from
 * row y0; any output pointer may be NULL.  With the camera ray's nearest hit (rt_primary_hits):
 *   point[3], refl[3]   the hit point and the reflected ray's unit direction (Scene::traceRay,
 *                       raytracer.cpp:531-538); zeros where the camera ray misses;
 *   occluded[words]     bit l of word l / 32: the shadow ray from the hit point towards light l is blocked
 *                       (Scene::traceShadowRay, raytracer.cpp:551-558, 565-567); words = max(1, (n_lights + 31) / 32);
 *   keys1               the nearest hit of the reflected ray (the object loop of the recursive traceRay call,
 *                       raytracer.cpp:543, 516-529), encoded as rt_primary_hits' keys; the miss key where
 *                       there is no reflected ray.
 * `count` is the capacity of the arrays in pixels.  Honours the options; blocks; default stream; leaves
 * those rows of the frame black. */
int rt_bounce_probe(rt_ctx* ctx, int y0, int rows, double* point, double* refl, uint32_t* occluded, uint64_t* keys1,
                    size_t count);

/* Test hook, host arithmetic only (needs no GPU and no ctx): the pixel that index p of the ray
 * queue of a pass stands for.  A pass of `rows` film rows of `width` pixels numbers its camera rays
 * in tile order (16 x 16 pixel tiles of four 8 x 8 blocks; what does not fill a tile in scanline
 * order), so that rays that are neighbours in the queue are neighbours on the film in both
 * directions; the reference's loops (raytracer.cpp:502-511) visit one scanline after the other and
 * nothing in them depends on that order.  *row in [0, rows) counts from the pass's first row, *x
 * in [0, width).  Returns RT_ERR_ARG when p is outside [0, width * rows). */
int rt_queue_pixel(int p, int width, int rows, int* row, int* x);

int rt_get_stats(rt_ctx* ctx, rt_stats* out);   /* waits for the ctx's work */
int rt_reset_stats(rt_ctx* ctx);

/* The device the ctx lives on: number of compute units and the engine clock in kHz as the HIP
 * runtime reports them (the measurement harness derives the FP32 vector peak from these:
 * CUs x 128 lanes x 2 x clock).  No counterpart in the reference. */
int rt_device_info(rt_ctx* ctx, int32_t* compute_units, int32_t* clock_khz);

void rt_free(rt_ctx* ctx);

const char* rt_last_error(void);

/* Serialised scene description ("RTSD", little endian), written by the host
 * loader (rth_save_scene) and by the reference-side binding; the layout is the
 * field order of rt_scene_desc with the arrays inlined:
 *   "RTSD", i32 version=1, i32 width, height, max_depth, f32 eps, shadow_eps,
 *   i32 n_objects, n_materials, n_transforms, n_lights,
 *   f64 cam_eye[3], cam_ll[3], cam_lr[3], cam_ul[3], cam_ur[3], ambient[3],
 *   i32 obj_kind[n], obj_material[n], obj_transform[n], obj_interp[n],
 *   (4 zero bytes if needed to reach a multiple of 8),
 *   f64 obj_geom[n][9], obj_normals[n][9], rt_material[n_materials],
 *   rt_transform[n_transforms], rt_light[n_lights].
 */

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
