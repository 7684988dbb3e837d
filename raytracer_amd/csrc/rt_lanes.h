// Render calls in flight (RT_OPT_FRAMES_IN_FLIGHT > 1): the hardware-queue probe, the deal of streams to lanes,
// the lanes' workspaces and events.
#pragma once

namespace {

// ---- RT_OPT_FRAMES_IN_FLIGHT > 1 ---------------------------------------------------------------------------

// Spins for `ticks` of the constant 100 MHz clock and notes when it began and ended (the probe below).
__global__ void k_spin(long long ticks, long long* span) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (span && threadIdx.x == 0) { span[0] = t0; span[1] = wall_clock64(); }
}

// The streams the lanes run on.  A process has few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); the runtime
// deals its streams onto them in an order of its own, and two streams on one queue run one after the other.  Lanes
// whose streams share a queue with another lane's do not overlap at all (measured: two lanes on an unlucky deal gain
// nothing, on four different queues 13 %), so the streams are sorted by queue here: twelve are created, and each is
// tried against one stream of every queue found so far -- a 200 us spin kernel on both, each noting the device clock
// when it began and when it ended: on different queues the two intervals overlap, on one queue the second begins
// when the first has ended.  The answer is read off the device's own clock, so a busy host, a profiler's
// interception or a slow launch path cannot turn it.  Once per context, ~10 ms.
struct QueueProbe {
    long long* spans = nullptr;          // device: [2 streams][begin, end]
    ~QueueProbe() { if (spans) (void)hipFree(spans); }
    int init() {
        HIP_TRY(hipMalloc((void**)&spans, 4 * sizeof(long long)));
        return 0;
    }
    // 1: a and b ran side by side; 0: one after the other; < 0: error.  `ticks`: length of the spin.
    int side_by_side(hipStream_t a, hipStream_t b, long long ticks) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, spans);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, spans + 2);
        HIP_TRY(hipStreamSynchronize(a));
        HIP_TRY(hipStreamSynchronize(b));
        long long h[4];
        HIP_TRY(hipMemcpy(h, spans, sizeof h, hipMemcpyDeviceToHost));
        const long long shared = std::min(h[1], h[3]) - std::max(h[0], h[2]);      // ticks during which both were running
        return shared > ticks / 4 ? 1 : 0;
    }
};

int probe_lane_streams_once(rt_ctx* c);
int probe_lane_streams(rt_ctx* c) {
    if (!c->lane_streams.empty()) return 0;
    const int rc = probe_lane_streams_once(c);
    if (rc) {               // (nothing half-made is kept: the next call starts over)
        for (auto st : c->lane_streams) (void)hipStreamDestroy(st);
        c->lane_streams.clear(); c->lane_stream_queue.clear(); c->lane_queues = 0;
    }
    return rc;
}
int probe_lane_streams_once(rt_ctx* c) {
    const int P = 12;
    for (int i = 0; i < P; i++) {
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->lane_streams.push_back(s);
    }
    c->lane_stream_queue.assign(P, -1);
    c->lane_queues = 0;
    QueueProbe probe;
    if (int rc = probe.init()) return rc;
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < P; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c->lane_streams[i], 100, (long long*)nullptr);     // (a stream's first launch sets it up)
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    const long long ticks = 20000;                      // 200 us
    std::vector<int> first_of;                          // a stream of every queue found so far
    for (int i = 0; i < P; i++) {
        for (size_t q = 0; q < first_of.size() && c->lane_stream_queue[i] < 0; q++) {
            const int side = probe.side_by_side(c->lane_streams[(size_t)first_of[q]], c->lane_streams[(size_t)i], ticks);
            if (side < 0) return side;
            if (side == 0) c->lane_stream_queue[i] = (int)q;
        }
        if (c->lane_stream_queue[i] < 0) {
            c->lane_stream_queue[i] = (int)first_of.size();
            first_of.push_back(i);
        }
    }
    HIP_TRY(hipGetLastError());
    c->lane_queues = (int)first_of.size();
    return 0;
}

// Which of the queues found does `st` (the caller's stream) sit on?  The same test, against one stream of each.
// A stream that is capturing a graph must not be given probe kernels: its queue stays unknown (-1).
int probe_caller_queue(rt_ctx* c, hipStream_t st) {
    c->caller_queue = -1;
    c->caller_probed = false;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (capturing != hipStreamCaptureStatusNone) return 0;
    QueueProbe probe;
    if (int rc = probe.init()) return rc;
    const long long ticks = 20000;
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100, (long long*)nullptr);
    HIP_TRY(hipStreamSynchronize(st));
    for (int q = 0; q < c->lane_queues && c->caller_queue < 0; q++) {
        int rep_stream = -1;
        for (size_t s = 0; s < c->lane_streams.size() && rep_stream < 0; s++) if (c->lane_stream_queue[s] == q) rep_stream = (int)s;
        if (rep_stream < 0) continue;
        const int side = probe.side_by_side(c->lane_streams[(size_t)rep_stream], st, ticks);
        if (side < 0) return side;
        if (side == 0) c->caller_queue = q;
    }
    c->caller_probed = true;
    return 0;
}

// Lane `i` of `k`: its main stream on queue i (mod the number of queues found).  While there are two queues per lane
// (two lanes on four queues) its side stream gets a queue of its own; otherwise the lane runs on its main stream
// alone.  Lanes must not meet on a queue: with the side streams of three lanes on the one queue their main streams
// leave over, three frames in flight take as long as one at a time (6.04 ms against 6.03), with every lane on a
// queue of its own 5.16 ms.
void deal_lane_streams(rt_ctx* c, int i, int k, hipStream_t* main_out, hipStream_t* side_out, int* main_queue_out, int* side_queue_out) {
    const int Q = std::max(1, c->lane_queues);
    // the queues in the order they are given out: the caller's stream's own queue last (what the caller's stream is
    // given -- rt_join's waits, the events that order a call behind the caller's work, an exchange -- queues up
    // behind everything a lane on that queue holds)
    std::vector<int> order;
    for (int q = 0; q < Q; q++) if (q != c->caller_queue) order.push_back(q);
    if (c->caller_queue >= 0 && c->caller_queue < Q) order.push_back(c->caller_queue);
    auto stream_on = [&](int queue, int nth) {         // the nth stream found on that queue (the last one if there are fewer)
        hipStream_t got = nullptr;
        for (size_t s = 0; s < c->lane_streams.size(); s++)
            if (c->lane_stream_queue[s] == queue) { got = c->lane_streams[s]; if (nth-- == 0) break; }
        return got;
    };
    const int main_queue = order[(size_t)(i % Q)];
    *main_out = stream_on(main_queue, i / Q);
    if (!*main_out) *main_out = c->lane_streams[(size_t)i % c->lane_streams.size()];      // (cannot happen: every queue found holds a stream)
    const int side_queue = (2 * k <= Q) ? order[(size_t)(k + i)] : -1;
    *side_out = side_queue >= 0 ? stream_on(side_queue, 0) : *main_out;
    if (!*side_out) *side_out = *main_out;
    *main_queue_out = main_queue;
    *side_queue_out = (*side_out == *main_out) ? -1 : side_queue;
    // RT_LANE_REPORT in the environment: what the probe found and how the lanes were dealt (stderr)
    if (std::getenv("RT_LANE_REPORT")) {
        if (i == 0) {
            fprintf(stderr, "lanes: %d hardware queue(s) found, the caller's stream on queue %d; streams by queue:", Q, c->caller_queue);
            for (size_t s = 0; s < c->lane_streams.size(); s++) fprintf(stderr, " %d", c->lane_stream_queue[s]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "lanes: lane %d of %d: main stream on queue %d, %s\n", i, k, main_queue,
                *side_out == *main_out ? "no side stream" : ("side stream on queue " + std::to_string(side_queue)).c_str());
    }
}

void release_lane(Lane* L) {
    if (!L) return;
    for (auto e : L->ev_hit) (void)hipEventDestroy(e);
    for (auto e : L->ev_phong) (void)hipEventDestroy(e);
    if (L->in) (void)hipEventDestroy(L->in);
    if (L->done) (void)hipEventDestroy(L->done);
    if (L->hint) (void)hipHostFree(L->hint);
    delete L;             // (the buffers release themselves; the streams belong to the context)
}

// Host-side wait for everything the context has enqueued: the last caller's stream and every lane.
int sync_all(rt_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    for (Lane* L : c->lanes) {
        if (!L->pending) continue;
        HIP_TRY(hipStreamSynchronize(L->main));      // (the side stream was joined into it before the folds)
        L->pending = false;
    }
    c->prev_done = nullptr;
    return 0;
}

void drop_lanes(rt_ctx* c) {
    // (what the lanes' traversals have counted since the last reset stays in the statistics)
    for (Lane* L : c->lanes) {
        if (!L->bvh_tally.p) continue;
        std::vector<unsigned long long> tally(L->bvh_tally.n);
        if (hipMemcpy(tally.data(), L->bvh_tally.p, tally.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) continue;
        for (size_t b = 0; b + 3 < tally.size(); b += 4)
            for (int k = 0; k < 3; k++) c->tally_carry[k] += tally[b + k];
    }
    for (Lane* L : c->lanes) release_lane(L);
    c->lanes.clear();
    c->prev_done = nullptr;
    c->cur = nullptr;
}

// `in_flight` lanes, each with a workspace for the current options, streams, events and (when the LBVH is built)
// traversal strips.  A lane is only added to the context when all of it exists.
int ensure_lanes(rt_ctx* c, hipStream_t caller) {
    size_t want = 0;
    int rc = pass_size(c, &want);
    if (rc) return rc;
    const int levels = c->hs.max_depth + 1;
    bool stale = (int)c->lanes.size() > c->in_flight;
    for (Lane* L : c->lanes) stale = stale || L->ws.cap != want || L->ws.levels != levels;
    if (stale) {
        HIP_TRY(hipDeviceSynchronize());
        drop_lanes(c);
    }
    if ((int)c->lanes.size() < c->in_flight) {
        rc = probe_lane_streams(c);
        if (rc) return rc;
        if (c->lanes.empty()) {             // (one deal for all lanes of a set)
            rc = probe_caller_queue(c, caller);
            if (rc) return rc;
        }
    }
    while ((int)c->lanes.size() < c->in_flight) {
        Lane* L = new Lane;
        auto give_up = [&](int code, const std::string& msg) { release_lane(L); return fail(code, msg); };
        hipError_t e = hipSuccess;
        deal_lane_streams(c, (int)c->lanes.size(), c->in_flight, &L->main, &L->side, &L->main_queue, &L->side_queue);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L->in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L->done, hipEventDisableTiming);
        for (int k = 0; k < 2 * levels && e == hipSuccess; k++) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) ((k & 1) ? L->ev_phong : L->ev_hit).push_back(ev);
        }
        if (e != hipSuccess) return give_up(RT_ERR_DEVICE, std::string("frames in flight: ") + hipGetErrorString(e));
        const size_t bytes = carve_workspace(nullptr, nullptr, want, levels, c->hs.n_lights, c->aux_grid, 16, (size_t)1 << 20, true);
        e = L->block.alloc(bytes);
        if (e != hipSuccess)
            return give_up(RT_ERR_DEVICE, std::string("workspace of ") + std::to_string(bytes >> 20) + " MiB for a frame in flight: " +
                                              hipGetErrorString(e) + " (lower RT_OPT_PASS_PIXELS or RT_OPT_FRAMES_IN_FLIGHT)");
        carve_workspace(&L->ws, L->block.p, want, levels, c->hs.n_lights, c->aux_grid, 16, (size_t)1 << 20, true);
        if (c->grid_hints && hipHostMalloc((void**)&L->hint, sizeof(PassHint), hipHostMallocDefault) != hipSuccess) { L->hint = nullptr; (void)hipGetLastError(); }
        if (L->hint) std::memset(L->hint, 0xff, sizeof(PassHint));
        L->ws.hint = L->hint;
        c->lanes.push_back(L);
    }
    for (Lane* L : c->lanes) {
        if (!c->bvh_built || L->bvh_spill.p) continue;
        HIP_TRY(L->bvh_spill.alloc(2 * (size_t)c->aux_grid * RT_BVH_BLOCK * RT_BVH_SPILL));
        HIP_TRY(L->bvh_tally.alloc(2 * (size_t)c->aux_grid * 4));
        HIP_TRY(hipMemset(L->bvh_tally.p, 0, 2 * (size_t)c->aux_grid * 4 * sizeof(unsigned long long)));
    }
    return 0;
}

}  // namespace
