// One pass of the trace loop as a sequence of launches: nearest hit and shadow queries of a level
// (launch_trace), the LBVH build, the level loop (run_pass) and the split of a render call into passes
// (render_rows).
#pragma once

namespace {

struct Timed {
    rt_ctx* c; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr;
    Timed(rt_ctx* c_, hipStream_t st_, int kind_, bool enabled = true) : c(c_), st(st_), kind(kind_) {
        if (c->profile && enabled) {
            a = take_event(c); b = take_event(c);
            if (a && b) (void)hipEventRecord(a, st);
        }
    }
    void stop() {
        if (a && b) {
            (void)hipEventRecord(b, st);
            c->spans.push_back({a, b, kind});
        }
    }
};

// Nearest hit (MODE 0) or shadow (MODE 1) of one queue: filter, listed pairs, the rest.
// `events`: false for the entry point that allocates nothing (event timing creates events).
// Blocks for a kernel that streams `items` queue entries (four per thread), between one block per compute unit and
// the full grid.
// Is the in-plane-origin question asked now (rt_ctx::plane_mode)?  Until the calibration frame has been traced: yes.
bool plane_asked(const rt_ctx* c) {
    if (c->calibrating) return true;
    if (c->plane_mode != 2) return c->plane_mode == 1;
    return c->plane_calibrated ? c->plane_found > 0 : true;
}

int blocks_for(const rt_ctx* c, size_t items) {
    const size_t blocks = (items + 4 * RT_BLOCK - 1) / (4 * RT_BLOCK);
    return (int)std::min<size_t>((size_t)c->aux_grid, std::max<size_t>((size_t)c->n_cu, blocks));
}

// The instances of the shadow and reflection scans without the in-plane-origin machinery (k_filter's PLANE = false), for
// launches that do not ask that question.
template <int M, int R, int CN>
static void launch_lean_filter(bool robust, dim3 grid, dim3 block, hipStream_t st, const DevScene* dsc, const v16f* records, const TraceArgs& A) {
    if constexpr (CN == 2 || CN == 4) {
        if (robust) hipLaunchKernelGGL((k_filter<M, R, true, CN, false, false>), grid, block, 0, st, dsc, records, A);
        else hipLaunchKernelGGL((k_filter<M, R, false, CN, false, false>), grid, block, 0, st, dsc, records, A);
    }
}

// `est_rays`: how many rays the queue is expected to hold (a hint, see PassHint; max_rays when nothing is known).
template <int MODE>
int launch_trace(rt_ctx* c, const Workspace& W, TraceArgs A, hipStream_t st, size_t max_rays, bool camera_rays = false,
                 bool events = true, size_t est_rays = ~(size_t)0) {
    const DevScene* dsc = c->d_scene.p;
    est_rays = std::min(est_rays, max_rays);
    // (a queue longer than expected is still walked in full: by fewer threads)
    const int ray_grid = blocks_for(c, 4 * est_rays + est_rays / 4);            // kernels with one ray per thread
    const int pair_grid = blocks_for(c, 4 * est_rays + est_rays);               // the listed pairs: a few per ray, four per thread
    A.camera_rays = (MODE == 0 && camera_rays) ? 1 : 0;
    A.beside_others = c->cur ? 1 : 0;
    A.plane_guard = plane_asked(c) ? 1 : 0;
    A.sphere_cull_min = c->sphere_cull_min;
    A.shared_origin_a = (MODE == 0 && camera_rays && c->shared_consts) ? c->eye_a.p : nullptr;
    A.shared_dir_rinv = (MODE == 1 && c->shared_consts && c->dir_rinv.n > 0) ? c->dir_rinv.p : nullptr;
    A.light_dir_row = c->light_dir_row.p;
    A.cone_camera = (A.shared_origin_a && c->pretest && c->cone_camera.n > 0) ? c->cone_camera.p : nullptr;
    const bool any_bounds = MODE == 0 && !camera_rays && c->pretest && c->pretest_any && c->sphere_any.n > 0;
    A.sphere_any = any_bounds ? c->sphere_any.p : nullptr;
    A.plane_any = any_bounds ? c->plane_any.p : nullptr;
    const bool light_bounds = MODE == 1 && c->shared_consts && c->pretest && c->bound_lights.n > 0;
    A.bound_lights = light_bounds ? c->bound_lights.p : nullptr;
    A.light_frame = light_bounds ? c->light_frame.p : nullptr;
    A.cl_camera = c->cl_camera.p; A.cl_lights = c->cl_lights.p;
    A.graze_slots = c->graze_slots.p; A.graze_start = c->graze_start.p;
    if (MODE == 1 && light_bounds) A.plane_any = c->plane_any.p;
    A.cl_sphere = c->cl_sphere.p; A.cl_normal = c->cl_normal.p; A.n_clusters = c->n_clusters;
    // this launch's header and flags, the launch before's, the next one's (see RT_FLAGS_HEADER)
    const unsigned int x = W.launch_no[MODE]++;
    uint8_t* const set = W.ray_flags[MODE];
    uint8_t* const flags = set + 3 * (size_t)RT_FLAGS_HEADER;
    auto header = [&](unsigned int k) { return set + (size_t)(k % 3u) * RT_FLAGS_HEADER; };
    A.pair_count = (unsigned long long*)header(x);
    A.task_counter = (unsigned int*)(header(x) + RT_PAIR_LISTS * 128);
    A.any_flagged = A.task_counter + 1;
    A.prev_words = (const unsigned int*)(header(x + 2u) + RT_PAIR_LISTS * 128);
    A.next_header = (unsigned long long*)header(x + 1u);
    A.ray_flags = flags + (size_t)(x & 1u) * W.flags_stride[MODE];
    A.prev_ray_flags = flags + (size_t)((x + 1u) & 1u) * W.flags_stride[MODE];
    A.pairs = W.pairs[MODE];
    A.pair_cap = (unsigned int)W.pair_cap[MODE]; A.counters = c->counters.p; A.scan_all = c->force_exact;
    // as many sub-lists as leave each at least 65536 entries (a flush appends up to 4096)
    A.pair_lists = 1;
    while (A.pair_lists < RT_PAIR_LISTS && W.pair_cap[MODE] / (2 * A.pair_lists) >= 65536) A.pair_lists *= 2;
    A.pair_seg = (unsigned int)(W.pair_cap[MODE] / A.pair_lists);
    const bool work = c->work_counters && c->work.p;
    A.work = work ? c->work.p : nullptr;
    // Headers and flags are at zero when a launch starts: the launches keep them so among themselves; only a workspace
    // that is new, or whose last launch of this kind was not enqueued to its end, is cleared here.
    // (RAYTRACER_AMD_CLEAR_EVERY_LAUNCH=1, a diagnostic switch: the fill in front of every launch all the same)
    if (!W.flags_clean[MODE] || c->clear_every_launch) HIP_TRY(hipMemsetAsync(W.ray_flags[MODE], 0, W.ray_flags_n[MODE], st));
    W.flags_clean[MODE] = false;
    bool guard_scan = false;              // LBVH mode: the in-plane-origin scan ran and listed pairs
    // A triangle whose unit normal is not a number is in the way of every shadow ray (see brute_scan, which marks them):
    // traceShadowRay (raytracer.cpp:551-558) returns at the first object in the way, and this launch looks for no other.
    const bool answered = MODE == 1 && c->n_void_normal > 0 && !c->force_exact;
    if (!c->force_exact && c->hs.n_regular > 0 && !answered) {
        int scan_kind = MODE == 0 ? 0 : 5;
        if (!c->use_bvh) {
            if (MODE == 0 && A.cone_camera) scan_kind = 1;
            else if (MODE == 0 && A.sphere_any) scan_kind = 4;
            else if (MODE == 1 && A.bound_lights) scan_kind = (c->h_lights.empty() || c->h_lights[0].is_point) ? 2 : 3;
        }
        Timed t(c, st, scan_kind, events);
        if (c->use_bvh) {
            // In a scene of spheres the shadow rays ask those first (cheap, wave by wave): the traversal then starts only
            // the rays that no sphere is in the way of.
            if (MODE == 1 && c->sphere_cull_min > 0 && c->hs.n_spheres >= c->sphere_cull_min) {
                TraceArgs SA = A;
                SA.spheres = 1;
                hipLaunchKernelGGL(k_spheres_first<MODE>, dim3(ray_grid), dim3(RT_BLOCK), 0, st, dsc, (const TriExact*)c->tri_exact.p, SA);
                A.spheres = 2;
            }
            BvhArgs ba;
            ba.bvh = c->bvh_groups.p; ba.nodes = c->bvh_nodes.p; ba.wide = c->bvh_wide.p; ba.filt = (const float*)c->tri_filt.p;
            ba.exact = c->tri_exact.p; ba.groups = c->groups.p; ba.xforms = c->xforms.p;
            int* spill = c->cur ? c->cur->bvh_spill.p : c->bvh_spill.p;
            unsigned long long* tally = c->cur ? c->cur->bvh_tally.p : c->bvh_tally.p;
            ba.spill = spill + (size_t)MODE * ((size_t)c->aux_grid * RT_BVH_BLOCK * RT_BVH_SPILL);   // one strip set per kind of query
            ba.counters = (BvhCounters*)&c->counters.p->bvh_node_tests;
            ba.tally = tally + (size_t)MODE * (size_t)c->aux_grid * 4;
            if (c->use_bvh != 2) hipLaunchKernelGGL((k_bvh_trace<MODE, true>), dim3(ray_grid), dim3(RT_BVH_BLOCK), 0, st, dsc, ba, A);
            else hipLaunchKernelGGL((k_bvh_trace<MODE, false>), dim3(ray_grid), dim3(RT_BVH_BLOCK), 0, st, dsc, ba, A);
            // The one hit no box hierarchy finds: a ray that starts exactly in the plane of a triangle and grazes it
            // "hits" it at its own origin wherever the triangle is (raytracer.cpp:319-321).  The bounded scan's own
            // machinery for that question -- camera rays: the triangles the host found the eye's plane distance too
            // small for; shadow rays: the lights' graze lists; reflection rays: clusters whose normal cone and sphere
            // admit such a plane -- lists the candidates, k_exact_pairs (below) decides them.
            guard_scan = plane_asked(c) && ((MODE == 0 && camera_rays) ? c->n_eye_plane > 0
                                          : MODE == 0 ? c->sphere_any.n > 0 : (c->n_graze > 0 && c->bound_lights.n > 0));
            if (guard_scan) {
                TraceArgs GA = A;
                GA.in_plane_only = 1;
                GA.shared_origin_a = (MODE == 0 && camera_rays) ? c->eye_a.p : nullptr;
                GA.shared_dir_rinv = (MODE == 1 && c->dir_rinv.n > 0) ? c->dir_rinv.p : nullptr;
                GA.cone_camera = (MODE == 0 && camera_rays) ? c->cone_camera.p : nullptr;
                GA.sphere_any = (MODE == 0 && !camera_rays) ? c->sphere_any.p : nullptr;
                GA.plane_any = c->plane_any.p;
                GA.bound_lights = MODE == 1 ? c->bound_lights.p : nullptr;
                GA.light_frame = MODE == 1 ? c->light_frame.p : nullptr;
                const dim3 grid(std::max(1, c->filter_grid[MODE][0][1][0] / (c->cur ? c->beside_grid_div : 1))), block(RT_BLOCK);
                const v16f* records = (const v16f*)c->tri_filt.p;
#define RT_LAUNCH_GUARD(M, R, CN)                                                                                    \
                do {                                                                                                  \
                    if (c->robust) hipLaunchKernelGGL((k_filter<M, R, true, CN, false>), grid, block, 0, st, dsc, records, GA);   \
                    else hipLaunchKernelGGL((k_filter<M, R, false, CN, false>), grid, block, 0, st, dsc, records, GA);             \
                } while (0)
                if (MODE == 0 && camera_rays) RT_LAUNCH_GUARD(0, RT_RPL_SHARED, 1);
                else if (MODE == 0) RT_LAUNCH_GUARD(0, RT_RPL_ANY, 4);
                else RT_LAUNCH_GUARD(1, RT_RPL_SHARED, 2);
#undef RT_LAUNCH_GUARD
            }
        } else {
            const dim3 block(RT_BLOCK);
            const v16f* records = (const v16f*)c->tri_filt.p;
#define RT_LAUNCH_FILTER(M, R, CN)                                                                                  \
            do {                                                                                                    \
                const dim3 grid(std::max(1, c->filter_grid[M][0][1][0] / (c->cur ? c->beside_grid_div : 1)));          \
                if (work) {                                                                                         \
                    if (c->robust) hipLaunchKernelGGL((k_filter<M, R, true, CN, true>), grid, block, 0, st, dsc, records, A);  \
                    else hipLaunchKernelGGL((k_filter<M, R, false, CN, true>), grid, block, 0, st, dsc, records, A); \
                } else if ((CN == 2 || CN == 4) && !A.plane_guard && c->lean_scans) {                               \
                    launch_lean_filter<M, R, CN>(c->robust != 0, grid, block, st, dsc, records, A);                 \
                } else {                                                                                            \
                    if (c->robust) hipLaunchKernelGGL((k_filter<M, R, true, CN, false>), grid, block, 0, st, dsc, records, A); \
                    else hipLaunchKernelGGL((k_filter<M, R, false, CN, false>), grid, block, 0, st, dsc, records, A); \
                }                                                                                                   \
            } while (0)
            if (MODE == 0 && A.cone_camera) RT_LAUNCH_FILTER(0, RT_RPL_SHARED, 1);
            else if (MODE == 0 && A.sphere_any) RT_LAUNCH_FILTER(0, RT_RPL_ANY, 4);
            else if (MODE == 1 && A.bound_lights) RT_LAUNCH_FILTER(1, RT_RPL_SHARED, 2);
            else RT_LAUNCH_FILTER(MODE, RT_RPL, 0);
#undef RT_LAUNCH_FILTER
        }
        HIP_TRY(hipGetLastError());
        t.stop();
    }
    Timed t(c, st, RT_WORK_KINDS, events);
    if (!c->force_exact && c->hs.n_regular > 0 && !answered && (!c->use_bvh || guard_scan)) {      // (the LBVH traversal itself lists no pairs)
        hipLaunchKernelGGL(k_exact_pairs<MODE>, dim3(pair_grid), dim3(RT_BLOCK), 0, st, dsc,
                           (const TriExact*)c->tri_exact.p, A);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_brute<MODE>, dim3(ray_grid), dim3(RT_BLOCK), 0, st, dsc, (const TriExact*)c->tri_exact.p, A);
    HIP_TRY(hipGetLastError());
    W.flags_clean[MODE] = true;
    t.stop();
    return 0;
}

// Builds the per-group LBVHs on the device (once per context).
int build_bvh(rt_ctx* c) {
    if (c->bvh_built) return 0;
    HIP_TRY(hipSetDevice(c->device));
    const int NR = c->hs.n_regular;
    const size_t G = c->h_groups.size();
    HIP_TRY(c->bvh_nodes.alloc((size_t)std::max(NR, 1)));
    HIP_TRY(c->bvh_wide.alloc((size_t)std::max(NR, 1)));
    std::vector<GroupBvh> hb(G);
    int max_n = 0;
    for (size_t g = 0; g < G; g++) max_n = std::max(max_n, c->h_groups[g].fax[3] - c->h_groups[g].fax[0]);
    DevBuf<unsigned int> keys_in, keys_out, arrive;
    DevBuf<int> vals_in, vals_out, parent_node, parent_leaf, leaf_of_slot;
    DevBuf<char> temp;
    // passes of tree rotations over the finished binary tree (k_bvh_rotate); RAYTRACER_AMD_BVH_ROTATIONS overrides
    int rotation_passes = RT_BVH_ROTATION_PASSES;
    if (const char* env = std::getenv("RAYTRACER_AMD_BVH_ROTATIONS")) rotation_passes = std::max(0, std::min(16, std::atoi(env)));
    const size_t m = (size_t)std::max(max_n, 1);
    HIP_TRY(keys_in.alloc(m)); HIP_TRY(keys_out.alloc(m)); HIP_TRY(vals_in.alloc(m)); HIP_TRY(vals_out.alloc(m));
    HIP_TRY(parent_node.alloc(m)); HIP_TRY(parent_leaf.alloc(m)); HIP_TRY(arrive.alloc(m)); HIP_TRY(leaf_of_slot.alloc(m));
    size_t temp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in.p, keys_out.p, vals_in.p, vals_out.p, m, 0, 30));
    HIP_TRY(temp.alloc(std::max<size_t>(temp_bytes, 16)));
    const float pad_abs = (float)(64.0 * 1.1920928955078125e-7 * c->scene_bound);
    size_t node_off = 0;
    auto release_all = [&]() {
        keys_in.release(); keys_out.release(); vals_in.release(); vals_out.release();
        parent_node.release(); parent_leaf.release(); arrive.release(); temp.release(); leaf_of_slot.release();
    };
    for (size_t g = 0; g < G; g++) {
        const Group& Gr = c->h_groups[g];
        const int base = Gr.fax[0], n = Gr.fax[3] - Gr.fax[0];
        hb[g].node_offset = node_off;
        hb[g].n = n;
        hb[g].first_slot = base;
        if (n <= 1) continue;
        const float* bx = &c->h_centroid_box[6 * g];
        float3 cmin = make_float3(bx[0], bx[1], bx[2]);
        // One scale for the three axes (the largest extent): a thin group -- a height field, a wall --
        // is then split along the directions it extends in, not sliced into layers across the thin one.
        const float ext = std::max(bx[3] - bx[0], std::max(bx[4] - bx[1], bx[5] - bx[2]));
        const float sc1 = ext > 0.f ? 1.0f / ext : 0.f;
        float3 cscale = make_float3(sc1, sc1, sc1);
        const int blocks = (n + 255) / 256;
        hipLaunchKernelGGL(k_bvh_morton, dim3(blocks), dim3(256), 0, 0, (const TriExact*)c->tri_exact.p, base, n, cmin,
                           cscale, keys_in.p, vals_in.p);
        hipError_t e = rocprim::radix_sort_pairs(temp.p, temp_bytes, keys_in.p, keys_out.p, vals_in.p, vals_out.p,
                                                 (size_t)n, 0, 30);
        if (e != hipSuccess) { release_all(); return fail(RT_ERR_DEVICE, std::string("radix sort: ") + hipGetErrorString(e)); }
        hipLaunchKernelGGL(k_bvh_hierarchy, dim3(blocks), dim3(256), 0, 0, (const unsigned int*)keys_out.p,
                           (const int*)vals_out.p, n, base, c->bvh_nodes.p + node_off, parent_node.p, parent_leaf.p);
        e = hipMemsetAsync(arrive.p, 0, (size_t)n * sizeof(unsigned int), 0);
        hipLaunchKernelGGL(k_bvh_refit, dim3(blocks), dim3(256), 0, 0, (const TriExact*)c->tri_exact.p,
                           (const int*)vals_out.p, n, base, c->bvh_nodes.p + node_off, (const int*)parent_node.p,
                           (const int*)parent_leaf.p, arrive.p, pad_abs);
        if (rotation_passes > 0 && n > 2) {
            hipLaunchKernelGGL(k_bvh_leaf_of_slot, dim3(blocks), dim3(256), 0, 0, n, (const int*)vals_out.p, leaf_of_slot.p);
            for (int pass = 0; pass < rotation_passes && e == hipSuccess; pass++) {
                e = hipMemsetAsync(arrive.p, 0, (size_t)n * sizeof(unsigned int), 0);
                hipLaunchKernelGGL(k_bvh_rotate, dim3(blocks), dim3(256), 0, 0, n, c->bvh_nodes.p + node_off,
                                   (const int*)parent_node.p, (const int*)parent_leaf.p, arrive.p, (unsigned int*)nullptr);
                hipLaunchKernelGGL(k_bvh_links, dim3(blocks), dim3(256), 0, 0, n, base, (const BvhNode*)(c->bvh_nodes.p + node_off),
                                   (const int*)leaf_of_slot.p, parent_node.p, parent_leaf.p);
            }
        }
        hipLaunchKernelGGL(k_bvh_widen, dim3(blocks), dim3(256), 0, 0, (const BvhNode*)(c->bvh_nodes.p + node_off), n - 1,
                           c->bvh_wide.p + node_off);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(0);
        if (e != hipSuccess) { release_all(); return fail(RT_ERR_DEVICE, std::string("bvh build: ") + hipGetErrorString(e)); }
        node_off += (size_t)(n - 1);
    }
    release_all();
    HIP_TRY(c->bvh_groups.upload(hb));
    HIP_TRY(c->bvh_spill.alloc(2 * (size_t)c->aux_grid * RT_BVH_BLOCK * RT_BVH_SPILL));
    HIP_TRY(c->bvh_tally.alloc(2 * (size_t)c->aux_grid * 4));
    HIP_TRY(hipMemset(c->bvh_tally.p, 0, 2 * (size_t)c->aux_grid * 4 * sizeof(unsigned long long)));
    c->bvh_built = true;
    return 0;
}

// FP32 scan copies of the rays (RayQueue::scan, TraceArgs::shadow_scan): kept when the workspace has room for
// them and the scans that read them will run.
bool keeps_scan_copies(const rt_ctx* c, const Workspace& ws) {
    return c->scan_copies && ws.scan[0] && !c->use_bvh && !c->force_exact && c->pretest && c->hs.n_regular > 0;
}
bool keeps_shadow_copies(const rt_ctx* c, const Workspace& ws) {
    return ws.shadow_scan[0] && c->shared_consts && c->bound_lights.n > 0 && c->light_frame.n > 0;
}

// One pass: rows q0 .. q0+nrows of the pass's row sequence, nrows * width <= cap.
// `ws`: the workspace to use (nrows * width <= ws.cap).  `side_work`: false keeps every launch on
// the caller's stream and records no event (the entry point that allocates nothing).
int run_pass(rt_ctx* c, const Workspace& ws, PassDesc pd, int nrows, hipStream_t st, bool side_work = true) {
    const int W = c->hs.width;
    const int n_pix = nrows * W;
    const int levels = ws.levels;
    const DevScene* dsc = c->d_scene.p;
    // FP32 scan copies of the rays for the bounded scans (RayQueue::scan): written by k_primary / k_hit when the
    // workspace has room for them and the scans that read them will run
    const bool scan = keeps_scan_copies(c, ws), scan_shadow = scan && keeps_shadow_copies(c, ws);
    RayQueue q[2] = {queue_of(ws, 0, scan), queue_of(ws, 1, scan)};
    hipLaunchKernelGGL(k_primary, dim3(blocks_for(c, (size_t)n_pix)), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q[0], ws.keys[0],
                       ws.pixel_of_ray, c->frame, ws.counts, ws.levels + 2);
    HIP_TRY(hipGetLastError());
    // Queue lengths of the last pass of this size in this workspace (see PassHint), a quarter on top; none: the pass's pixels
    const bool hinted = c->grid_hints && ws.hint && ws.hint->n_pix == n_pix && ws.hint->counts[0] == (uint32_t)n_pix && levels + 2 <= 72;
    auto expected = [&](int k) -> size_t {
        if (!hinted || ws.hint->counts[k] > (uint32_t)n_pix) return (size_t)n_pix;
        return std::min<size_t>((size_t)n_pix, (size_t)ws.hint->counts[k] + ws.hint->counts[k] / 4 + 1024);
    };
    // Per level: nearest hit -> k_hit on the caller's stream; shadow rays -> k_phong on the side
    // stream, beside the next level's nearest-hit launches (both only read what k_hit wrote).
    // The short launches of the deep levels then fill each other's idle SIMDs.  With event
    // timing on (RT_OPT_PROFILE) everything stays on one stream so that a kernel's duration is
    // its own.  Hit auxiliaries and occlusion bits alternate between two buffers by level
    // parity; k_hit of level k+2 waits for k_phong of level k before it overwrites them.
    hipStream_t side = c->cur ? c->cur->side : c->side_stream;
    const bool overlap = side_work && c->overlap && !c->profile && levels > 1 && side != st;      // (a lane may have one stream only)
    hipEvent_t* ev_hit = c->cur ? c->cur->ev_hit.data() : c->ev_hit.data();
    hipEvent_t* ev_phong = c->cur ? c->cur->ev_phong.data() : c->ev_phong.data();
    hipStream_t sb = overlap ? side : st;
    // (the fold: one launch along the chains of reflected hits, or, for deep scenes, level by level)
    const bool chain = levels <= RT_FOLD_MAX_LEVELS && !c->fold_by_level;
    for (int k = 0; k < levels; k++) {
        const int cur = k & 1, nxt = cur ^ 1;
        LevelStore lvl = level_of(ws, k);
        double* ab = ws.aux[cur];
        HitAux aux;
        aux.nx = ab; aux.ny = ab + ws.cap; aux.nz = ab + 2 * ws.cap;
        aux.vx = ab + 3 * ws.cap; aux.vy = ab + 4 * ws.cap; aux.vz = ab + 5 * ws.cap;
        uint32_t* occl = ws.occluded[cur];
        TraceArgs ta{};
        ta.rays = q[cur]; ta.n_items = ws.counts + k; ta.keys = ws.keys[cur];
        ta.occl_words = ws.occl_words;
        ta.scan_nearest = (scan && c->all_identity) ? 1 : 0;
        int rc = launch_trace<0>(c, ws, ta, st, (size_t)n_pix, k == 0, side_work, expected(k));
        if (rc) return rc;
        const int level_grid = blocks_for(c, expected(k)), hits_grid = blocks_for(c, expected(k + 1));

        if (overlap && k >= 2) HIP_TRY(hipStreamWaitEvent(st, ev_phong[k - 2], 0));
        HitArgs ha{};
        ha.sc = dsc; ha.cur = q[cur]; ha.next = q[nxt]; ha.keys = ws.keys[cur]; ha.next_keys = ws.keys[nxt];
        ha.n_rays = ws.counts + k; ha.n_hits = ws.counts + k + 1; ha.lvl = lvl; ha.aux = aux;
        ha.hit_counts = ws.hit_counts;
        ha.chain = chain ? 1 : 0;
        ha.camera_rays = k == 0 ? 1 : 0;
        ha.parent_next = (chain && k > 0) ? level_of(ws, k - 1).next : nullptr;
        ha.occluded = occl; ha.occl_words = ws.occl_words;
        ha.shadow_scan = scan_shadow ? ws.shadow_scan[cur] : nullptr; ha.shadow_stride = (uint32_t)ws.cap;
        ha.light_frame = c->light_frame.p;
        hipLaunchKernelGGL(k_hit_counts, dim3(level_grid), dim3(RT_BLOCK), 0, st, (const unsigned long long*)ws.keys[cur],
                           (const uint32_t*)(ws.counts + k), ws.hit_counts);
        hipLaunchKernelGGL(k_hit, dim3(level_grid), dim3(RT_BLOCK), 0, st, ha);      // (the same grid: a block's range of the queue is the one that was counted)
        HIP_TRY(hipGetLastError());
        if (overlap) {
            HIP_TRY(hipEventRecord(ev_hit[k], st));
            HIP_TRY(hipStreamWaitEvent(sb, ev_hit[k], 0));
        }

        if (c->hs.n_lights > 0) {
            TraceArgs sa{};
            sa.rays = q[nxt]; sa.n_items = ws.counts + k + 1; sa.source_obj = lvl.obj;
            sa.occluded = occl; sa.occl_words = ws.occl_words;
            sa.shadow_scan = scan_shadow ? ws.shadow_scan[cur] : nullptr; sa.shadow_stride = (uint32_t)ws.cap;
            rc = launch_trace<1>(c, ws, sa, sb, (size_t)n_pix * (size_t)std::max(1, c->hs.n_lights), false, side_work,
                                 expected(k + 1) * (size_t)std::max(1, c->hs.n_lights));
            if (rc) return rc;
        }

        PhongArgs pa{};
        pa.sc = dsc; pa.next = q[nxt]; pa.aux = aux; pa.lvl = lvl; pa.occluded = occl;
        pa.occl_words = ws.occl_words; pa.cap = (uint32_t)ws.cap; pa.n_hits = ws.counts + k + 1;
        pa.init_child = chain ? 0 : 1;
        hipLaunchKernelGGL(k_phong, dim3(hits_grid), dim3(RT_BLOCK), 0, sb, pa);
        HIP_TRY(hipGetLastError());
        if (overlap) HIP_TRY(hipEventRecord(ev_phong[k], sb));
    }
    // the fold (and whatever the caller enqueues next) comes after everything on the side stream
    if (overlap) HIP_TRY(hipStreamWaitEvent(st, ev_phong[levels - 1], 0));
    // this pass's queue lengths for the next pass of its size: written to pinned memory by the kernel that counts
    uint32_t* hint_counts = nullptr;
    if (c->grid_hints && ws.hint && levels + 2 <= 72) {
        ws.hint->n_pix = n_pix;
        hint_counts = ws.hint->counts;
    }
    if (chain) {
        FoldAllArgs fa{};
        fa.hint = hint_counts;
        fa.sc = dsc; fa.parent = ws.lvl_parent; fa.obj = ws.lvl_obj; fa.phong = ws.lvl_phong; fa.next = (const int32_t*)ws.lvl_child;
        fa.cap = (uint32_t)ws.cap; fa.levels = levels; fa.counts = ws.counts;
        fa.pixel_of_ray = ws.pixel_of_ray; fa.frame = c->frame; fa.counters = c->counters.p;
        hipLaunchKernelGGL(k_fold_all, dim3(blocks_for(c, expected(1))), dim3(RT_BLOCK), 0, st, fa);
        HIP_TRY(hipGetLastError());
    } else {
        for (int k = levels - 1; k >= 0; k--) {
            FoldArgs fa{};
            fa.sc = dsc; fa.lvl = level_of(ws, k); fa.parent_lvl = level_of(ws, k > 0 ? k - 1 : 0);
            fa.cap = (uint32_t)ws.cap; fa.n_hits = ws.counts + k + 1; fa.level = k;
            fa.pixel_of_ray = ws.pixel_of_ray; fa.frame = c->frame;
            hipLaunchKernelGGL(k_fold, dim3(blocks_for(c, expected(k + 1))), dim3(RT_BLOCK), 0, st, fa);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, st, dsc, (const uint32_t*)ws.counts, levels, c->counters.p, hint_counts);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// Rows y_start + j*stride_rows + [0, tile_rows) for j = 0 .. n_tiles-1 (clipped to the frame).
int render_rows(rt_ctx* c, int y_start, int tile_rows, int stride_rows, int n_tiles, hipStream_t st) {
    HIP_TRY(hipSetDevice(c->device));
    // (the first render call of a context traces the calibration frame of RT_OPT_IN_PLANE_ORIGIN first)
    if (int rc = calibrate_plane(c)) return rc;
    const int H = c->hs.height, W = c->hs.width;
    // valid rows form a prefix of the pass's row sequence: only the last tile can be clipped
    long long total_rows = 0;
    for (int j = 0; j < n_tiles; j++) {
        long long y0 = (long long)y_start + (long long)j * stride_rows;
        if (y0 >= H) break;
        total_rows += std::min<long long>(tile_rows, H - y0);
    }
    // Event timing and the counting instances book per launch into host-side sums and one device table:
    // such calls run one at a time on the caller's stream.
    const bool in_lanes = c->in_flight > 1 && !c->profile && !c->work_counters;
    if (!in_lanes) {
        int rc = ensure_workspace(c);
        if (rc) return rc;
        // (calls still in flight in lanes come first: pixels land in call order)
        for (Lane* L : c->lanes)
            if (L->pending) HIP_TRY(hipStreamWaitEvent(st, L->done, 0));
        c->prev_done = nullptr;
        const int rows_per_pass = (int)(c->ws.cap / W);
        c->last_stream = st;
        for (long long q0 = 0; q0 < total_rows; q0 += rows_per_pass) {
            int nrows = (int)std::min<long long>(rows_per_pass, total_rows - q0);
            PassDesc pd{y_start, tile_rows, stride_rows, (int)q0};
            rc = run_pass(c, c->ws, pd, nrows, st);
            if (rc) return rc;
        }
        return 0;
    }
    int rc = ensure_lanes(c, st);
    if (rc) return rc;
    Lane* L = c->lanes[c->seq % (unsigned long long)c->in_flight];
    // Behind what the caller's stream holds now (and, by stream order, behind the lane's previous call) -- when it
    // holds anything.  The caller's stream shares its hardware queue with one of the lanes, and an event recorded on
    // it queues up behind everything that lane has been given: with an idle caller's stream asked to mark every call,
    // calls waited for marks stuck behind a whole frame of another lane (20 frames took 100 ms instead of 93
    // whenever the first of them did not go to that lane).
    const hipError_t busy = hipStreamQuery(st);
    (void)hipGetLastError();                   // (hipErrorNotReady is an answer, not a failure to report later)
    if (busy != hipSuccess) {
        HIP_TRY(hipEventRecord(L->in, st));
        HIP_TRY(hipStreamWaitEvent(L->main, L->in, 0));
    }
    // A call that writes a buffer another call in flight writes runs wholly behind that call: it clears its pixels
    // before it traces them, and a reader that has joined the earlier call must not see them go black.  (A caller
    // that pipelines frames binds one buffer per call in flight and meets none of these waits.)
    for (Lane* other : c->lanes)
        if (other != L && other->pending && other->frame == c->frame) HIP_TRY(hipStreamWaitEvent(L->main, other->done, 0));
    L->frame = c->frame;
    const int rows_per_pass = (int)(L->ws.cap / W);
    c->last_stream = st;
    c->cur = L;
    for (long long q0 = 0; q0 < total_rows && rc == 0; q0 += rows_per_pass) {
        int nrows = (int)std::min<long long>(rows_per_pass, total_rows - q0);
        PassDesc pd{y_start, tile_rows, stride_rows, (int)q0};
        rc = run_pass(c, L->ws, pd, nrows, L->main);
    }
    c->cur = nullptr;
    // (also after a failed launch: what was enqueued still runs, and the next call's folds wait for it)
    hipError_t e = hipEventRecord(L->done, L->main);
    L->pending = true;
    L->seq = c->seq++;
    c->prev_done = L->done;
    c->prev_frame = c->frame;
    if (rc) return rc;
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("frames in flight: ") + hipGetErrorString(e));
    return 0;
}

}  // namespace
