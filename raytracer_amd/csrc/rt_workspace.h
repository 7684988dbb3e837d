// The per-pass workspace: how one block of device memory is cut into queues, keys, per-level hit records and
// pair lists; sizing for the current options; event timing bookkeeping.
#pragma once

namespace {

// Queue i of a workspace; `scan`: with its FP32 scan copies (when the workspace keeps them).
RayQueue queue_of(const Workspace& w, int i, bool scan) {
    RayQueue q;
    double* b = w.rayq[i];
    const size_t cap = w.cap;
    q.ox = b; q.oy = b + cap; q.oz = b + 2 * cap;
    q.dx = b + 3 * cap; q.dy = b + 4 * cap; q.dz = b + 5 * cap;
    q.scan = scan ? w.scan[i] : nullptr;
    q.scan_stride = (uint32_t)cap;
    return q;
}

LevelStore level_of(const Workspace& w, int k) {
    LevelStore l;
    l.parent = w.lvl_parent + (size_t)k * w.cap;
    l.obj = w.lvl_obj + (size_t)k * w.cap;
    l.phong = w.lvl_phong + (size_t)k * 3 * w.cap;
    l.child = w.lvl_child + (size_t)k * 3 * w.cap;
    l.next = (int32_t*)l.child;
    return l;
}

// Cuts a workspace for passes of `cap` pixels out of the block at `base` (256-byte aligned
// slices) and returns the bytes it takes; base == nullptr only measures.  The pair lists hold
// `pairs_per_ray` entries per ray of the largest launch of their kind (a few are typical; a
// list that runs full sends its rays through the full reference-precision scan instead), at
// least `min_pairs`, at most 2^28.
// `scan_copies`: also room for the FP32 scan copies of the rays (shadow rays: up to RT_SCAN_MAX_LIGHTS lights).
#define RT_SCAN_MAX_LIGHTS 8
size_t carve_workspace(Workspace* w, unsigned char* base, size_t cap, int levels, int n_lights, int aux_grid,
                       size_t pairs_per_ray, size_t min_pairs, bool scan_copies) {
    size_t off = 0;
    auto take = [&](size_t bytes) -> unsigned char* {
        unsigned char* at = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    Workspace t;
    t.cap = cap; t.levels = levels;
    t.occl_words = std::max(1, (n_lights + 31) / 32);
    for (int i = 0; i < 2; i++) {
        t.rayq[i] = (double*)take(6 * cap * sizeof(double));
        t.keys[i] = (unsigned long long*)take(cap * sizeof(unsigned long long));
        t.aux[i] = (double*)take(6 * cap * sizeof(double));
        t.occluded[i] = (uint32_t*)take(cap * (size_t)t.occl_words * sizeof(uint32_t));
    }
    t.pixel_of_ray = (int32_t*)take(cap * sizeof(int32_t));
    t.lvl_parent = (int32_t*)take((size_t)levels * cap * sizeof(int32_t));
    t.lvl_obj = (int32_t*)take((size_t)levels * cap * sizeof(int32_t));
    t.lvl_phong = (double*)take((size_t)levels * 3 * cap * sizeof(double));
    t.lvl_child = (double*)take((size_t)levels * 3 * cap * sizeof(double));
    t.counts = (uint32_t*)take((size_t)(levels + 2) * sizeof(uint32_t));
    t.hit_counts = (uint32_t*)take((size_t)aux_grid * sizeof(uint32_t));
    const size_t max_rays[2] = {cap, cap * (size_t)std::max(1, n_lights)};
    for (int m = 0; m < 2; m++) {
        t.flags_stride[m] = (max_rays[m] + 255) & ~(size_t)255;
        t.ray_flags_n[m] = 3 * (size_t)RT_FLAGS_HEADER + 2 * t.flags_stride[m];
        t.ray_flags[m] = (uint8_t*)take(t.ray_flags_n[m]);
        t.pair_cap[m] = std::min<size_t>(std::max<size_t>(max_rays[m] * pairs_per_ray, min_pairs), (size_t)1 << 28);
        // (test hook: a pair list of at most so many entries -- 0: every ray with a candidate takes the full scan)
        if (const char* env = std::getenv("RAYTRACER_AMD_PAIR_CAP")) t.pair_cap[m] = std::min<size_t>(t.pair_cap[m], (size_t)std::strtoull(env, nullptr, 10));
        t.pairs[m] = (unsigned long long*)take(t.pair_cap[m] * sizeof(unsigned long long));
    }
    if (scan_copies) {
        for (int i = 0; i < 2; i++) {
            t.scan[i] = (float*)take(9 * cap * sizeof(float));
            if (n_lights >= 1 && n_lights <= RT_SCAN_MAX_LIGHTS) t.shadow_scan[i] = (float4*)take((size_t)n_lights * cap * sizeof(float4));
        }
    }
    if (w) *w = t;
    return off;
}

// Sizes the context's own workspace for the current options.  Nothing in the context changes
// unless every step succeeds: a failure leaves the previous workspace (or none) in place and
// the caller returns the error without launching anything.
// Pixels per pass for the current options (whole rows), and the error if even one row is too many.
int pass_size(const rt_ctx* c, size_t* want_out) {
    size_t frame_px = (size_t)c->hs.width * c->hs.height;
    // shadow rays of a pass are numbered with 31 bits
    const long long by_lights = ((1ll << 31) - 1) / std::max(1, c->hs.n_lights);
    const long long budget = std::max<long long>(std::min<long long>(c->pass_pixels, by_lights), c->hs.width);
    size_t want = (size_t)std::min<long long>((long long)frame_px, budget);
    // whole rows per pass
    want = std::max<size_t>(c->hs.width, want / c->hs.width * c->hs.width);
    const size_t shadow_rays = want * (size_t)std::max(1, c->hs.n_lights);
    if (shadow_rays >= (1ull << 31))
        return fail(RT_ERR_RANGE, "pass too large (one row of shadow rays exceeds 2^31): fewer lights or a narrower frame");
    *want_out = want;
    return 0;
}

int ensure_workspace(rt_ctx* c) {
    size_t want = 0;
    int rc = pass_size(c, &want);
    if (rc) return rc;
    const int levels = c->hs.max_depth + 1;
    if (c->ws.cap == want && c->ws.levels == levels && c->ws_block.p) return 0;
    if (!c->side_stream) HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    while ((int)c->ev_hit.size() < levels || (int)c->ev_phong.size() < levels) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (c->ev_hit.size() <= c->ev_phong.size()) c->ev_hit.push_back(e); else c->ev_phong.push_back(e);
    }
    // room for 16 listed pairs per ray of the largest launch (a few are typical), 8 B each
    const size_t bytes = carve_workspace(nullptr, nullptr, want, levels, c->hs.n_lights, c->aux_grid, 16, (size_t)1 << 20, true);
    HIP_TRY(hipDeviceSynchronize());          // no launch may still use the block released here
    c->ws = Workspace{};
    c->ws_block.release();
    hipError_t e = c->ws_block.alloc(bytes);
    if (e != hipSuccess) {
        c->ws_block.release();
        return fail(RT_ERR_DEVICE, std::string("workspace of ") + std::to_string(bytes >> 20) + " MiB: " + hipGetErrorString(e) +
                                       " (lower RT_OPT_PASS_PIXELS)");
    }
    Workspace w;
    carve_workspace(&w, c->ws_block.p, want, levels, c->hs.n_lights, c->aux_grid, 16, (size_t)1 << 20, true);
    if (!c->hint && c->grid_hints && hipHostMalloc((void**)&c->hint, sizeof(PassHint), hipHostMallocDefault) != hipSuccess) { c->hint = nullptr; (void)hipGetLastError(); }
    if (c->hint) std::memset(c->hint, 0xff, sizeof(PassHint));
    w.hint = c->hint;
    c->ws = w;
    return 0;
}

hipEvent_t take_event(rt_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Resolve recorded spans into the running sums (requires their work to be finished).
void drain_spans(rt_ctx* c) {
    for (auto& s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->ms[s.kind] += ms;
            c->launches[s.kind]++;
        }
        c->event_pool.push_back(s.a);
        c->event_pool.push_back(s.b);
    }
    c->spans.clear();
}

}  // namespace
