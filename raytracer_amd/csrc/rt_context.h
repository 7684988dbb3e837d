// What a context holds: the host copy of the scene's records, the per-pass workspace, the lanes of render
// calls in flight, options and statistics.
#pragma once

// One triangle as the host keeps it: the reference-precision record and the filter record in
// both forms (static worst-case margins / per-test margins), see filter_z2.
struct HostTri {
    TriExact ex;
    float f_static[16], f_robust[16];
    bool ok_static, ok_robust;
    float margin_static;      // the u plane's static margin in barycentric units
    int xf, axis;
};

// Queue lengths of the last pass traced in a workspace, copied to pinned host memory behind that pass: the next
// pass of the same size launches its streaming kernels (k_hit, k_phong, k_fold, k_exact_pairs, the LBVH traversal ...)
// with grids sized for those lengths instead of for the largest queue the pass could have.  The frame's rays are the
// same from call to call, so the lengths repeat; a hint that is stale or absent costs time, never pixels (every such
// kernel walks its queue with a grid-stride loop).  A deep level holds a few per cent of the pixels: a full-size grid
// there is thousands of blocks that start, find nothing and leave, while the other calls in flight wait for the slots.
struct PassHint {
    int n_pix;                     // pixels of the pass the counts belong to (written by the host when the copy is enqueued)
    uint32_t counts[72];           // [levels + 2] queue lengths as the pass left them (all ones until a copy has landed)
};

// The per-pass workspace: queues, keys, per-level hit records, pair lists.  Every array is a
// slice of ONE block of device memory (the context's own allocation, or the library's static
// scratch array for the entry point that allocates nothing), cut by carve_workspace.
struct Workspace {
    size_t cap = 0;                // rays (pixels) per pass
    int levels = 0;                // max_depth + 1
    int occl_words = 1;
    double* rayq[2] = {nullptr, nullptr};              // [6][cap] each
    unsigned long long* keys[2] = {nullptr, nullptr};
    double* aux[2] = {nullptr, nullptr};               // [6][cap], by level parity
    uint32_t* occluded[2] = {nullptr, nullptr};        // [cap][occl_words], by level parity
    int32_t* pixel_of_ray = nullptr;
    int32_t *lvl_parent = nullptr, *lvl_obj = nullptr; // [levels][cap]
    double *lvl_phong = nullptr, *lvl_child = nullptr; // [levels][3][cap]
    uint32_t* counts = nullptr;                        // [levels + 2]
    uint32_t* hit_counts = nullptr;                    // [aux_grid]: k_hit_counts -> k_hit
    // Three launch headers (the pair list's fill counters, the task counter, ...) and two arrays of per-ray flags, which
    // the trace launches take in turn (RT_FLAGS_HEADER); one set per kind of query ([0] nearest hit, [1] shadow) so
    // that a level's shadow launches can run beside the next level's nearest-hit launches
    uint8_t* ray_flags[2] = {nullptr, nullptr};
    size_t ray_flags_n[2] = {0, 0};                    // bytes of the whole set
    size_t flags_stride[2] = {0, 0};                   // bytes from one flag array to the other
    mutable unsigned int launch_no[2] = {0, 0};        // trace launches of the kind enqueued so far
    mutable bool flags_clean[2] = {false, false};      // ... and the last one was enqueued to its end (else: one fill of the set)
    unsigned long long* pairs[2] = {nullptr, nullptr};
    size_t pair_cap[2] = {0, 0};
    // FP32 scan copies (RayQueue::scan, TraceArgs::shadow_scan); null when the workspace keeps none
    float* scan[2] = {nullptr, nullptr};               // [9][cap], beside rayq[i]
    float4* shadow_scan[2] = {nullptr, nullptr};       // [n_lights][cap], by level parity
    PassHint* hint = nullptr;                          // pinned host memory (null: none kept, e.g. the static workspace)
};

// What one render call in flight owns (RT_OPT_FRAMES_IN_FLIGHT > 1): a workspace, the two streams of a pass,
// the per-level events between them, the LBVH traversal's per-thread strips and per-block tallies.
struct Lane {
    Workspace ws;
    DevBuf<unsigned char> block;
    hipStream_t main = nullptr, side = nullptr;
    std::vector<hipEvent_t> ev_hit, ev_phong;
    hipEvent_t in = nullptr, done = nullptr;   // the caller's stream at the time of the call / the call's last kernel
    bool pending = false;                      // `done` was recorded and nobody has waited for all of it yet
    unsigned long long seq = 0;                // number of the render call that recorded `done`
    uint32_t* frame = nullptr;                 // the frame buffer that call writes
    int main_queue = -1, side_queue = -1;      // the hardware queues its streams were found on (side: -1 = none of its own)
    DevBuf<int> bvh_spill;
    DevBuf<unsigned long long> bvh_tally;
    PassHint* hint = nullptr;
};

struct rt_ctx {
    int device = 0;
    int n_cu = 0;
    DevScene hs{};                 // host copy (device pointers inside)
    DevBuf<DevScene> d_scene;
    DevBuf<v16f> tri_filt;
    DevBuf<TriExact> tri_exact;
    DevBuf<SphereRec> spheres;
    DevBuf<SphereFilt> sphere_filt;
    DevBuf<Group> groups;
    DevBuf<double> xforms;
    DevBuf<int32_t> xf_identity, obj_kind, obj_slot, obj_material, obj_transform, obj_interp;
    DevBuf<double> obj_geom, obj_normals;
    DevBuf<DevMaterial> materials;
    DevBuf<DevLight> lights;
    DevBuf<uint32_t> own_frame;
    uint32_t* frame = nullptr;     // own_frame.p or the caller's buffer
    DevBuf<DevCounters> counters;
    DevBuf<unsigned long long> work;   // [RT_WORK_KINDS][RT_WORK_FIELDS], see WorkCount
    int work_counters = 0;             // RT_OPT_WORK_COUNTERS
    int scan_copies = 0;               // RT_OPT_SCAN_COPIES (off: the scans gain 1-3 %, k_hit's extra stores cost the frame more)
    bool all_identity = false;         // every transform group is the identity (object space = world space)

    // workspace, sized for `ws.cap` rays per pass: one allocation, cut by carve_workspace
    Workspace ws;
    DevBuf<unsigned char> ws_block;
    PassHint* hint = nullptr;             // of the context's own workspace
    int grid_hints = 1;                   // RAYTRACER_AMD_GRID_HINTS=0 in the environment switches them off (A/B runs)
    hipStream_t side_stream = nullptr;    // the shadow/Phong half of a level runs here
    std::vector<hipEvent_t> ev_hit, ev_phong;   // per level: k_hit done / k_phong done
    int overlap = 1;                      // RT_OPT_OVERLAP
    // The scan launches of a call in flight beside others get half the blocks: two calls' scans are then resident side
    // by side on every CU instead of one filling the chip and the next taking over as it drains (four in flight:
    // 4.85 -> 4.70 ms per frame; a quarter of the blocks: 5.4 ms).
    int beside_grid_div = 2;
    int n_void_normal = 0;                       // triangles whose unit normal is not a number: one of them shadows every ray (see brute_scan)
    int sphere_cull_min = 4;                     // RAYTRACER_AMD_SPHERE_CULL=n: spheres from which k_brute drops them wave by wave (0: never; A/B, tests)
    bool lean_scans = true;                      // RAYTRACER_AMD_LEAN_SCANS=0 (read at upload): the full instances of the scans also when the in-plane-origin question is not asked (A/B, tests)
    bool clear_every_launch = false;             // RAYTRACER_AMD_CLEAR_EVERY_LAUNCH=1 (read at upload): a fill in front of every trace launch (A/B, tests)
    bool fold_by_level = false;                  // RAYTRACER_AMD_FOLD_BY_LEVEL=1: the level-by-level fold whatever the depth (A/B, tests)
    // RT_OPT_FRAMES_IN_FLIGHT: render call i runs in lanes[i % in_flight] (none of this is used at 1)
    int in_flight = 1;
    std::vector<Lane*> lanes;
    // The lanes' streams, created once and kept until rt_free: which hardware queue a stream lands on is decided
    // when it is first used, and streams that share a queue do not overlap -- an arrangement that overlaps is kept.
    std::vector<hipStream_t> lane_streams;
    std::vector<int> lane_stream_queue;   // per stream: the hardware queue it was found to sit on (numbered as found)
    int lane_queues = 0;                  // how many different ones
    int caller_queue = -1;                // the queue the caller's stream sat on when the lanes were made (-1: none of those found, or not known)
    bool caller_probed = false;           // ... it was tried against every queue found (false: the stream was capturing)
    Lane* cur = nullptr;                  // the lane of the call being enqueued (null: the context's own resources)
    // (`done` of the previous call in flight: kept for the bookkeeping of sync_all)
    hipEvent_t prev_done = nullptr;
    uint32_t* prev_frame = nullptr;
    unsigned long long seq = 0;           // render calls issued in lanes so far
    // LBVH tallies of lanes that have been released (rt_get_stats adds them; rt_reset_stats clears them)
    unsigned long long tally_carry[3] = {0, 0, 0};

    // host copies kept so that the filter form can be switched after the upload
    std::vector<HostTri> h_tris;          // in Scene::objects order (triangles only)
    std::vector<GroupFilter> h_q0, h_q1;  // per transform: nearest-hit / shadow thresholds
    std::vector<int32_t> h_slot;          // per object: slot of its record (spheres: fixed; triangles: per form)
    std::vector<int32_t> h_xf_ident;
    std::vector<double> h_xf;             // [n_transforms][48]
    int clock_khz = 0;                    // engine clock reported by the runtime
    // The form of the filter chosen from the scene's own numbers, per configuration: where every pair takes the
    // triangle test (no bounding tests) the static margins save a multiply-add per ray and triangle unless they are
    // useless; where bounding tests decide most pairs, few triangle tests are left and the per-test margins pay as
    // soon as the static ones list a few per cent of pairs too many for the reference-precision re-test.
    int auto_robust = 0;                  // ... without bounding tests
    int auto_robust_bounded = 0;          // ... with them (RT_OPT_BOUNDING_TESTS, the default)
    int form_option = 0;                  // RT_OPT_FILTER_FORM as last set: 0 = chosen here, 1 static, 2 per-test
    int robust = 0;                       // the form installed now

    // per-triangle constants shared by all rays of a launch (see filter_z2)
    DevBuf<float> eye_a;                  // [n_regular + 1]
    DevBuf<float> dir_rinv;               // [directional lights][n_regular + 1]
    DevBuf<int32_t> light_dir_row;        // [n_lights]
    DevBuf<float> cone_camera;            // [n_regular + 8][4]: bounding cone of each triangle seen from the eye
    DevBuf<float> bound_lights;           // [n_lights][n_regular + 8][4]: bounding cone / disc of each triangle per light
    DevBuf<double> light_frame;           // [n_lights][8]: plane coordinates of a directional light
    DevBuf<float> sphere_any, plane_any;  // [n_regular + 8][4] each: bounding sphere / plane of each triangle (rays with nothing in common)
    // the same one level up: clusters of 64 consecutive regular slots (see TraceArgs)
    DevBuf<float> cl_camera, cl_lights, cl_sphere, cl_normal;
    DevBuf<int32_t> graze_slots, graze_start;   // per light: the triangles a shadow ray can graze (see TraceArgs)
    int n_clusters = 0;
    // RT_OPT_IN_PLANE_ORIGIN: the in-plane-origin question (raytracer.cpp:319-321: a ray that starts exactly in a
    // triangle's plane and grazes it "hits" at its own origin, wherever in the plane the triangle is).  Asking it costs
    // the bounded scans their naming machinery (graze lists, cluster normals, every triangle in whose plane an origin
    // lies visited) and the LBVH a scan behind every traversal, and almost no scene has a ray that takes the branch.
    // Mode 2 (default) decides once per context: camera, frame size and depth belong to the scene, so every render call
    // traces the same rays; one whole frame traced with the question asked everywhere tells whether any pair of it
    // takes the branch at all, and only then is the question kept.  Until that frame has been traced it is asked.
    int plane_mode = 2;                   // RT_OPT_IN_PLANE_ORIGIN: 0 never asked, 1 always, 2 decided by a calibration frame
    bool plane_calibrated = false;
    unsigned long long plane_found = 0;   // pairs of the calibration frame that took the branch
    double plane_calibration_s = 0;
    bool calibrating = false;             // the calibration frame is being rendered (it must not start another)
    int n_eye_plane = 0;                  // regular triangles in whose plane the eye may lie (FP32 bound): the camera rays' candidates
    size_t n_graze = 0;                   // entries of all lights' graze lists: the shadow rays' candidates
    int n_regular_real = 0;               // regular triangles without the copies that fill up the axis ranges
    int pretest_any = 1;                  // ... and whether those are used
    int pretest = 1;                      // bounding-volume tests in front of the FP32 triangle test
    std::vector<DevLight> h_lights;
    int shared_consts = 1;                // RT_OPT_SHARED_CONSTANTS

    // optional LBVH (RT_OPT_USE_BVH)
    int use_bvh = 0;
    bool bvh_built = false;
    DevBuf<BvhNode> bvh_nodes;
    DevBuf<BvhNode4> bvh_wide;            // the same nodes four-wide (what RT_OPT_USE_BVH = 1 traverses)
    DevBuf<GroupBvh> bvh_groups;
    DevBuf<int> bvh_spill;                // deep traversal-stack entries, one strip per thread of the launch
    DevBuf<unsigned long long> bvh_tally; // [2 kinds of query][aux_grid][4]: see BvhArgs::tally
    std::vector<Group> h_groups;
    std::vector<float> h_centroid_box;   // per group: min[3], max[3] of the triangle centroids
    double scene_bound = 0;

    // options
    int force_exact = 0;
    int profile = 0;
    long long pass_pixels = 16 << 20;     // a 4K frame goes through in one pass (about 0.6 KB of workspace per pixel)

    // blocks per launch of the scan kernels, by [MODE][bounding test][margin form][counting]: as many as are
    // resident at once for that instance.  The launches use the figure of the plain scan (4 blocks per CU) for
    // every instance: the bounded instances hold 3 blocks per CU, the fourth waits for a slot, and with the
    // number of tasks kept equal the two geometries measured the same (3.45 / 3.42 ms on the reflection scan).
    int filter_grid[2][5][2][2] = {};
    int aux_grid = 0;
    hipStream_t last_stream = nullptr;

    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    // summed durations (RT_OPT_PROFILE): [0 .. RT_WORK_KINDS) the scan launches by kind (see WorkCount; LBVH
    // traversals are booked as kinds 0 and 5), [RT_WORK_KINDS] the reference-precision kernels
    double ms[RT_WORK_KINDS + 1] = {};
    unsigned launches[RT_WORK_KINDS + 1] = {};
};

