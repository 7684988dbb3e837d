// C ABI of the MI355X trace loop (include/rt_abi.h): scene upload with the
// per-primitive precomputation, the pass scheduler that strings the kernels of
// rt_kernels.h together, download and statistics.
#include "rt_abi.h"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <string.h>

#include "rt_kernels.h"
#include "rt_bvh.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_host_util.h"

#define RT_MAX_FRAMES_IN_FLIGHT 8       // lanes a context may hold (RT_OPT_FRAMES_IN_FLIGHT)


// Workspace of rt_render_tile_noalloc: device memory the library holds from the moment it is
// loaded until the process ends, on every GPU it runs on, so that the call itself allocates
// nothing.  Host code never uses the array's name as an address: rt_upload_scene resolves the
// current device's copy once (hipGetSymbolAddress) and everything goes through that pointer.
#define RT_SCRATCH_BYTES ((size_t)64 << 20)
#define RT_SCRATCH_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];

namespace {
std::mutex g_scratch_mutex;
void* g_scratch_ptr[RT_SCRATCH_MAX_DEVICES];       // per device: address of its copy of `scratch`

// With `device` current: the address of that device's copy of the scratch array.
int resolve_scratch(int device) {
    if (device < 0 || device >= RT_SCRATCH_MAX_DEVICES) return 0;      // (such a device has no static workspace)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return 0;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RT_ERR_DEVICE, "hipGetSymbolAddress(scratch) returned a null address");
    g_scratch_ptr[device] = ptr;
    return 0;
}
}  // namespace

#include "rt_context.h"
#include "rt_workspace.h"
#include "rt_lanes.h"
static int calibrate_plane(rt_ctx* c);          // (render_rows starts the calibration frame of RT_OPT_IN_PLANE_ORIGIN when it is due)
#include "rt_pass.h"

static int upload_scene(rt_ctx** out, const rt_scene_desc* d);
extern "C" void rt_free(rt_ctx* c);
#include "rt_upload.h"

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

// No exception may cross the C boundary (host allocations, host threads of the precomputation):
// the entry points that can meet one turn it into an error code.
#define RT_GUARD_BEGIN try {
#define RT_GUARD_END(name)                                                                              \
    } catch (const std::bad_alloc&) { return fail(RT_ERR_DEVICE, name ": out of host memory"); }       \
      catch (const std::exception& e) { return fail(RT_ERR_DEVICE, std::string(name ": ") + e.what()); } \
      catch (...) { return fail(RT_ERR_DEVICE, name ": unknown failure"); }

int rt_upload_scene(rt_ctx** out, const rt_scene_desc* d) {
    RT_GUARD_BEGIN
    return upload_scene(out, d);
    RT_GUARD_END("rt_upload_scene")
}

}  // extern "C"

extern "C" {

// Installs the form of the filter the options now ask for, if it is not the one installed (and rebuilds the LBVH over
// the new slot order when the context uses one).
static int reinstall_form(rt_ctx* c) {
    const int want = chosen_form(c);
    if (want == c->robust) return 0;
    const int had_bvh = c->use_bvh;
    int rc = install_filter(c, want);
    if (rc) return rc;
    if (had_bvh) return build_bvh(c);
    return 0;
}

// RT_OPT_IN_PLANE_ORIGIN = 2: traces one whole frame, in the mode the context is in, with the in-plane-origin question
// asked everywhere, and keeps the question iff some pair of that frame took the branch (rt_ctx::plane_mode).  The
// frame goes to the context's own frame buffer, its counts stay out of the statistics.
static int calibrate_plane(rt_ctx* c) {
    if (c->plane_mode != 2 || c->plane_calibrated || c->calibrating) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(c->device));
    int rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    DevCounters before;
    HIP_TRY(hipMemcpy(&before, c->counters.p, sizeof before, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> tally;
    if (c->bvh_tally.p) {
        tally.resize(2 * (size_t)c->aux_grid * 4);
        HIP_TRY(hipMemcpy(tally.data(), c->bvh_tally.p, tally.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    uint32_t* const frame = c->frame;
    const int in_flight = c->in_flight, profile = c->profile, work = c->work_counters, exact = c->force_exact;
    c->frame = c->own_frame.p; c->in_flight = 1; c->profile = 0; c->work_counters = 0; c->force_exact = 0;
    c->calibrating = true;
    rc = render_rows(c, 0, c->hs.height, c->hs.height, 1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    c->calibrating = false;
    c->frame = frame; c->in_flight = in_flight; c->profile = profile; c->work_counters = work; c->force_exact = exact;
    if (rc) return rc;
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("calibration frame: ") + hipGetErrorString(e));
    DevCounters after;
    HIP_TRY(hipMemcpy(&after, c->counters.p, sizeof after, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c->counters.p, &before, sizeof before, hipMemcpyHostToDevice));
    if (!tally.empty()) HIP_TRY(hipMemcpy(c->bvh_tally.p, tally.data(), tally.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(c->own_frame.p, 0, (size_t)c->hs.width * c->hs.height * 4));
    c->plane_found = after.in_plane_origin_hits - before.in_plane_origin_hits;
    c->plane_calibrated = true;
    c->plane_calibration_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("RT_GUARD_REPORT"))
        fprintf(stderr, "in-plane-origin question: the calibration frame found %llu pairs that take the branch, in %.3f s: %s\n",
                c->plane_found, c->plane_calibration_s, c->plane_found ? "asked" : "not asked");
    return 0;
}

int rt_set_option(rt_ctx* c, int option, int64_t value) {
    if (!c) return fail(RT_ERR_ARG, "rt_set_option: null ctx");
    RT_GUARD_BEGIN
    // (tables that an option rewrites must not be in use: calls in flight in lanes are waited for)
    for (Lane* L : c->lanes)
        if (L->pending) { HIP_TRY(hipSetDevice(c->device)); int rc = sync_all(c); if (rc) return rc; break; }
    switch (option) {
        case RT_OPT_EXACT_ONLY: c->force_exact = value ? 1 : 0; return 0;
        case RT_OPT_PROFILE: c->profile = value ? 1 : 0; return 0;
        case RT_OPT_PASS_PIXELS:
            if (value < 1) return fail(RT_ERR_ARG, "rt_set_option: pass pixels must be positive");
            c->pass_pixels = value;
            return 0;
        case RT_OPT_FILTER_FORM: {
            if (value < 0 || value > 2) return fail(RT_ERR_ARG, "rt_set_option: filter form is 0 (auto), 1 (static) or 2 (per-test)");
            c->form_option = (int)value;
            return reinstall_form(c);
        }
        case RT_OPT_SHARED_CONSTANTS: c->shared_consts = value ? 1 : 0; return 0;
        case RT_OPT_OVERLAP: c->overlap = value ? 1 : 0; return 0;
        case RT_OPT_WORK_COUNTERS: c->work_counters = value ? 1 : 0; return 0;
        case RT_OPT_SCAN_COPIES: c->scan_copies = value ? 1 : 0; return 0;
        case RT_OPT_FRAMES_IN_FLIGHT: {
            if (value < 1 || value > RT_MAX_FRAMES_IN_FLIGHT) return fail(RT_ERR_ARG, "rt_set_option: 1 to 8 frames in flight");
            HIP_TRY(hipSetDevice(c->device));
            int rc = sync_all(c);
            if (rc) return rc;
            // (another number of lanes is another deal of the streams: the lanes are made anew at the next call;
            // at 1 their workspaces go back to the device)
            if ((int)value != c->in_flight) drop_lanes(c);
            c->in_flight = (int)value;
            // (this call blocks anyway: the lanes' streams are sorted by hardware queue here, so that the first
            // render call only has to find the queue of the caller's stream)
            if (value > 1) return probe_lane_streams(c);
            return 0;
        }
        case RT_OPT_IN_PLANE_ORIGIN:
            if (value < 0 || value > 2) return fail(RT_ERR_ARG, "rt_set_option: the in-plane-origin question is 0 (never asked), 1 (always) or 2 (decided by a calibration frame)");
            c->plane_mode = (int)value;
            return 0;
        case RT_OPT_BOUNDING_TESTS:
            c->pretest = value ? 1 : 0; c->pretest_any = value == 1 ? 1 : 0;
            return reinstall_form(c);          // (the form chosen for a scan without bounding tests may be the other one)
        case RT_OPT_USE_BVH:
            if (value) {
                HIP_TRY(hipSetDevice(c->device));
                int rc = build_bvh(c);
                if (rc) return rc;
            }
            c->use_bvh = value == 2 ? 2 : (value ? 1 : 0);      // 1: four-wide nodes, 2: the binary tree they are made from
            // (the calibration frame of RT_OPT_IN_PLANE_ORIGIN, if still due, here rather than inside the first render call)
            if (value) return calibrate_plane(c);
            return 0;
        default: return fail(RT_ERR_ARG, "rt_set_option: unknown option");
    }
    RT_GUARD_END("rt_set_option")
}

int rt_render_tile(rt_ctx* c, int y0, int rows, void* stream) {
    if (!c) return fail(RT_ERR_ARG, "rt_render_tile: null ctx");
    if (rows < 0 || y0 < 0 || (long long)y0 + rows > c->hs.height)
        return fail(RT_ERR_RANGE, "rt_render_tile: rows outside the frame");
    if (rows == 0) return 0;
    return render_rows(c, y0, rows, rows, 1, (hipStream_t)stream);
}

int rt_render_tiles(rt_ctx* c, int first_tile, int tile_rows, int tile_step, void* stream) {
    if (!c) return fail(RT_ERR_ARG, "rt_render_tiles: null ctx");
    if (first_tile < 0 || tile_rows <= 0 || tile_step <= 0) return fail(RT_ERR_ARG, "rt_render_tiles: bad tiling");
    const int n_total = (c->hs.height + tile_rows - 1) / tile_rows;
    if (first_tile >= n_total) return 0;
    const int n_mine = (n_total - first_tile + tile_step - 1) / tile_step;
    return render_rows(c, first_tile * tile_rows, tile_rows, tile_step * tile_rows, n_mine, (hipStream_t)stream);
}

int rt_join(rt_ctx* c, void* stream, int keep_latest) {
    if (!c) return fail(RT_ERR_ARG, "rt_join: null ctx");
    if (keep_latest < 0) return fail(RT_ERR_ARG, "rt_join: keep_latest must not be negative");
    RT_GUARD_BEGIN
    HIP_TRY(hipSetDevice(c->device));
    for (Lane* L : c->lanes)
        if (L->pending && c->seq - L->seq > (unsigned long long)keep_latest)
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, L->done, 0));
    return 0;
    RT_GUARD_END("rt_join")
}

int rt_render_tile_noalloc(rt_ctx* c, int y0, int rows, void* stream) {
    if (!c) return fail(RT_ERR_ARG, "rt_render_tile_noalloc: null ctx");
    RT_GUARD_BEGIN
    if (rows < 0 || y0 < 0 || (long long)y0 + rows > c->hs.height)
        return fail(RT_ERR_RANGE, "rt_render_tile_noalloc: rows outside the frame");
    if (rows == 0) return 0;
    if (c->device < 0 || c->device >= RT_SCRATCH_MAX_DEVICES || !g_scratch_ptr[c->device])
        return fail(RT_ERR_STATE, "rt_render_tile_noalloc: no static workspace on this device");
    unsigned char* base = (unsigned char*)g_scratch_ptr[c->device];      // this device's copy, never the array's name
    const int W = c->hs.width, levels = c->hs.max_depth + 1, lights = c->hs.n_lights;
    // the whole tile is one pass; 8 listed pairs per ray (a list that runs full costs time, not pixels)
    const size_t pixels = (size_t)rows * (size_t)W;
    const size_t need = carve_workspace(nullptr, nullptr, pixels, levels, lights, c->aux_grid, 8, 0, false);
    if (need > RT_SCRATCH_BYTES || pixels * (size_t)std::max(1, lights) >= ((size_t)1 << 31)) {
        // largest row count of this scene that fits (the size grows monotonically with the rows)
        int lo = 0, hi = rows;
        while (lo + 1 < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (carve_workspace(nullptr, nullptr, (size_t)mid * W, levels, lights, c->aux_grid, 8, 0, false) <= RT_SCRATCH_BYTES) lo = mid; else hi = mid;
        }
        char msg[256];
        snprintf(msg, sizeof msg, "rt_render_tile_noalloc: %d rows of %d pixels need %zu bytes of workspace, the static workspace "
                 "holds %zu (at most %d rows per call for this scene's depth and lights)", rows, W, need, (size_t)RT_SCRATCH_BYTES, lo);
        return fail(RT_ERR_RANGE, msg);
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    Workspace ws;
    carve_workspace(&ws, base, pixels, levels, lights, c->aux_grid, 8, 0, false);
    // The array keeps whatever the previous call left in it, at whatever offsets that call's sizes
    // put it: queue lengths, counters and flags are cleared by the pass (a workspace struct that is new
    // has the launch headers and flags filled before their first use).
    // (calls still in flight in lanes come first: pixels land in call order; a wait allocates nothing)
    for (Lane* L : c->lanes)
        if (L->pending) HIP_TRY(hipStreamWaitEvent(st, L->done, 0));
    c->last_stream = st;
    PassDesc pd{y0, rows, rows, 0};
    return run_pass(c, ws, pd, rows, st, /*side_work=*/false);
    RT_GUARD_END("rt_render_tile_noalloc")
}

int rt_primary_hits(rt_ctx* c, int y0, int rows, uint64_t* keys, size_t count) {
    if (!c || !keys) return fail(RT_ERR_ARG, "rt_primary_hits: null argument");
    if (rows <= 0 || y0 < 0 || (long long)y0 + rows > c->hs.height)
        return fail(RT_ERR_RANGE, "rt_primary_hits: rows outside the frame");
    const int W = c->hs.width;
    if (count < (size_t)rows * W) return fail(RT_ERR_RANGE, "rt_primary_hits: buffer too small");
    RT_GUARD_BEGIN
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());          // renders still in flight on other streams use the same workspace
    int rc = ensure_workspace(c);
    if (rc) return rc;
    const Workspace& ws = c->ws;
    const int rows_per_pass = (int)(ws.cap / W);
    hipStream_t st = nullptr;
    const DevScene* dsc = c->d_scene.p;
    std::vector<uint64_t> queued;
    for (int q0 = 0; q0 < rows; q0 += rows_per_pass) {
        const int nrows = std::min(rows_per_pass, rows - q0);
        const int n_pix = nrows * W;
        PassDesc pd{y0, rows, rows, q0};
        const bool scan = keeps_scan_copies(c, ws);
        RayQueue q = queue_of(ws, 0, scan);
        hipLaunchKernelGGL(k_primary, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q, ws.keys[0],
                           ws.pixel_of_ray, c->frame, ws.counts, ws.levels + 2);
        HIP_TRY(hipGetLastError());
        TraceArgs ta{};
        ta.rays = q; ta.n_items = ws.counts; ta.keys = ws.keys[0]; ta.occl_words = ws.occl_words;
        ta.scan_nearest = (scan && c->all_identity) ? 1 : 0;
        rc = launch_trace<0>(c, ws, ta, st, (size_t)n_pix, true);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        // the queue is in tile order (queue_pixel); the caller's array is in scanline order
        queued.resize((size_t)n_pix);
        HIP_TRY(hipMemcpy(queued.data(), ws.keys[0], (size_t)n_pix * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int p = 0; p < n_pix; p++) {
            int row, x;
            queue_pixel(p, W, nrows, &row, &x);
            keys[(size_t)(q0 + row) * W + x] = queued[(size_t)p];
        }
    }
    return 0;
    RT_GUARD_END("rt_primary_hits")
}

int rt_bounce_probe(rt_ctx* c, int y0, int rows, double* point, double* refl, uint32_t* occluded, uint64_t* keys1, size_t count) {
    if (!c) return fail(RT_ERR_ARG, "rt_bounce_probe: null ctx");
    if (rows <= 0 || y0 < 0 || (long long)y0 + rows > c->hs.height)
        return fail(RT_ERR_RANGE, "rt_bounce_probe: rows outside the frame");
    const int W = c->hs.width;
    if (count < (size_t)rows * W) return fail(RT_ERR_RANGE, "rt_bounce_probe: buffers too small");
    RT_GUARD_BEGIN
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());          // renders still in flight on other streams use the same workspace
    int rc = ensure_workspace(c);
    if (rc) return rc;
    const Workspace& ws = c->ws;
    if (ws.levels < 2 && keys1) return fail(RT_ERR_STATE, "rt_bounce_probe: the scene's depth is 0, there is no reflected ray");
    const int rows_per_pass = (int)(ws.cap / W);
    const int words = ws.occl_words;
    hipStream_t st = nullptr;
    const DevScene* dsc = c->d_scene.p;
    const size_t n_px = (size_t)rows * W;
    if (point) std::fill(point, point + 3 * n_px, 0.0);
    if (refl) std::fill(refl, refl + 3 * n_px, 0.0);
    if (occluded) std::fill(occluded, occluded + (size_t)words * n_px, 0u);
    if (keys1) std::fill(keys1, keys1 + n_px, (uint64_t)RT_MISS_KEY);
    std::vector<int32_t> parent;
    std::vector<double> rays;
    std::vector<uint32_t> occl;
    std::vector<uint64_t> k1;
    for (int q0 = 0; q0 < rows; q0 += rows_per_pass) {
        const int nrows = std::min(rows_per_pass, rows - q0);
        const int n_pix = nrows * W;
        PassDesc pd{y0, rows, rows, q0};
        const bool scan = keeps_scan_copies(c, ws), scan_shadow = scan && keeps_shadow_copies(c, ws);
        RayQueue q[2] = {queue_of(ws, 0, scan), queue_of(ws, 1, scan)};
        hipLaunchKernelGGL(k_primary, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q[0], ws.keys[0],
                           ws.pixel_of_ray, c->frame, ws.counts, ws.levels + 2);
        HIP_TRY(hipGetLastError());
        TraceArgs ta{};
        ta.rays = q[0]; ta.n_items = ws.counts; ta.keys = ws.keys[0]; ta.occl_words = words;
        ta.scan_nearest = (scan && c->all_identity) ? 1 : 0;
        rc = launch_trace<0>(c, ws, ta, st, (size_t)n_pix, true);
        if (rc) return rc;
        LevelStore lvl = level_of(ws, 0);
        double* ab = ws.aux[0];
        HitAux aux;
        aux.nx = ab; aux.ny = ab + ws.cap; aux.nz = ab + 2 * ws.cap;
        aux.vx = ab + 3 * ws.cap; aux.vy = ab + 4 * ws.cap; aux.vz = ab + 5 * ws.cap;
        HitArgs ha{};
        ha.sc = dsc; ha.cur = q[0]; ha.next = q[1]; ha.keys = ws.keys[0]; ha.next_keys = ws.keys[1];
        ha.n_rays = ws.counts; ha.n_hits = ws.counts + 1; ha.lvl = lvl; ha.aux = aux;
        ha.hit_counts = ws.hit_counts;
        ha.camera_rays = 1;
        ha.occluded = ws.occluded[0]; ha.occl_words = words;
        ha.shadow_scan = scan_shadow ? ws.shadow_scan[0] : nullptr; ha.shadow_stride = (uint32_t)ws.cap;
        ha.light_frame = c->light_frame.p;
        hipLaunchKernelGGL(k_hit_counts, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, (const unsigned long long*)ws.keys[0],
                           (const uint32_t*)ws.counts, ws.hit_counts);
        hipLaunchKernelGGL(k_hit, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, ha);
        HIP_TRY(hipGetLastError());
        if (c->hs.n_lights > 0) {
            TraceArgs sa{};
            sa.rays = q[1]; sa.n_items = ws.counts + 1; sa.source_obj = lvl.obj;
            sa.occluded = ws.occluded[0]; sa.occl_words = words;
            sa.shadow_scan = scan_shadow ? ws.shadow_scan[0] : nullptr; sa.shadow_stride = (uint32_t)ws.cap;
            rc = launch_trace<1>(c, ws, sa, st, (size_t)n_pix * (size_t)std::max(1, c->hs.n_lights));
            if (rc) return rc;
        }
        if (keys1) {
            TraceArgs tb{};
            tb.rays = q[1]; tb.n_items = ws.counts + 1; tb.keys = ws.keys[1]; tb.occl_words = words;
            tb.scan_nearest = (scan && c->all_identity) ? 1 : 0;
            rc = launch_trace<0>(c, ws, tb, st, (size_t)n_pix, false);
            if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(st));
        uint32_t n_hits = 0;
        HIP_TRY(hipMemcpy(&n_hits, ws.counts + 1, sizeof n_hits, hipMemcpyDeviceToHost));
        if (n_hits > (uint32_t)n_pix) return fail(RT_ERR_DEVICE, "rt_bounce_probe: hit count out of range");
        parent.resize(n_hits); rays.resize(6 * (size_t)n_hits); occl.resize((size_t)words * n_hits); k1.resize(n_hits);
        if (n_hits) {
            HIP_TRY(hipMemcpy(parent.data(), lvl.parent, n_hits * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int a = 0; a < 6; a++)
                HIP_TRY(hipMemcpy(rays.data() + (size_t)a * n_hits, ws.rayq[1] + (size_t)a * ws.cap, n_hits * sizeof(double), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(occl.data(), ws.occluded[0], (size_t)words * n_hits * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (keys1) HIP_TRY(hipMemcpy(k1.data(), ws.keys[1], n_hits * sizeof(uint64_t), hipMemcpyDeviceToHost));
        }
        for (uint32_t h = 0; h < n_hits; h++) {
            int row, x;
            queue_pixel(parent[h], W, nrows, &row, &x);       // the queue is in tile order; the caller's arrays in scanline order
            const size_t px = (size_t)(q0 + row) * W + x;
            for (int a = 0; a < 3; a++) {
                if (point) point[3 * px + a] = rays[(size_t)a * n_hits + h];
                if (refl) refl[3 * px + a] = rays[(size_t)(3 + a) * n_hits + h];
            }
            if (occluded) for (int w = 0; w < words; w++) occluded[px * words + w] = occl[(size_t)h * words + w];
            if (keys1) keys1[px] = k1[h];
        }
    }
    return 0;
    RT_GUARD_END("rt_bounce_probe")
}

int rt_queue_pixel(int p, int width, int rows, int* row, int* x) {
    if (!row || !x) return fail(RT_ERR_ARG, "rt_queue_pixel: null argument");
    if (width <= 0 || rows <= 0 || p < 0 || (long long)p >= (long long)width * rows)
        return fail(RT_ERR_ARG, "rt_queue_pixel: index outside the pass");
    queue_pixel(p, width, rows, row, x);
    return 0;
}

int rt_bind_frame(rt_ctx* c, void* device_rgba, size_t bytes) {
    if (!c) return fail(RT_ERR_ARG, "rt_bind_frame: null ctx");
    if (!device_rgba) { c->frame = c->own_frame.p; return 0; }
    if (bytes < (size_t)c->hs.width * c->hs.height * 4) return fail(RT_ERR_RANGE, "rt_bind_frame: buffer too small");
    if ((uintptr_t)device_rgba % 4) return fail(RT_ERR_ARG, "rt_bind_frame: buffer must be 4-byte aligned");
    c->frame = (uint32_t*)device_rgba;
    return 0;
}

int rt_download(rt_ctx* c, uint8_t* rgba, size_t bytes) {
    if (!c || !rgba) return fail(RT_ERR_ARG, "rt_download: null argument");
    const size_t need = (size_t)c->hs.width * c->hs.height * 4;
    if (bytes < need) return fail(RT_ERR_RANGE, "rt_download: buffer too small");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    HIP_TRY(hipMemcpy(rgba, c->frame, need, hipMemcpyDeviceToHost));
    return 0;
}

int rt_get_stats(rt_ctx* c, rt_stats* out) {
    if (!c || !out) return fail(RT_ERR_ARG, "rt_get_stats: null argument");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    drain_spans(c);
    DevCounters dc;
    HIP_TRY(hipMemcpy(&dc, c->counters.p, sizeof dc, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->rays_primary = dc.rays_primary; out->rays_reflect = dc.rays_reflect; out->rays_shadow = dc.rays_shadow;
    out->hits = dc.hits; out->closest_tests = dc.closest_tests; out->anyhit_tests = dc.anyhit_tests;
    out->exact_tests = dc.exact_tests;
    out->bvh_node_tests = dc.bvh_node_tests; out->bvh_leaf_tests = dc.bvh_leaf_tests;
    if (c->bvh_tally.p) {       // the LBVH traversals count per block (BvhArgs::tally)
        std::vector<unsigned long long> tally(2 * (size_t)c->aux_grid * 4);
        HIP_TRY(hipMemcpy(tally.data(), c->bvh_tally.p, tally.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < tally.size(); b += 4) {
            out->bvh_node_tests += tally[b]; out->bvh_leaf_tests += tally[b + 1]; out->exact_tests += tally[b + 2];
        }
        for (Lane* L : c->lanes) {
            if (!L->bvh_tally.p) continue;
            HIP_TRY(hipMemcpy(tally.data(), L->bvh_tally.p, tally.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (size_t b = 0; b < tally.size(); b += 4) {
                out->bvh_node_tests += tally[b]; out->bvh_leaf_tests += tally[b + 1]; out->exact_tests += tally[b + 2];
            }
        }
    }
    out->bvh_node_tests += c->tally_carry[0]; out->bvh_leaf_tests += c->tally_carry[1]; out->exact_tests += c->tally_carry[2];
    out->in_plane_origin_hits = dc.in_plane_origin_hits; out->flagged_rays = dc.flagged_rays;
    out->closest_ms = c->ms[0] + c->ms[1] + c->ms[4]; out->anyhit_ms = c->ms[2] + c->ms[3] + c->ms[5];
    out->closest_launches = c->launches[0] + c->launches[1] + c->launches[4];
    out->anyhit_launches = c->launches[2] + c->launches[3] + c->launches[5];
    out->camera_ms = c->ms[1]; out->camera_launches = c->launches[1]; out->bvh_plane_guard = plane_asked(c) ? 1u : 0u;
    out->exact_ms = c->ms[RT_WORK_KINDS]; out->exact_launches = c->launches[RT_WORK_KINDS];
    static_assert(RT_WORK_KINDS == RT_SCAN_KINDS && sizeof(rt_scan_work) == RT_WORK_FIELDS * sizeof(uint64_t), "scan work layout");
    HIP_TRY(hipMemcpy(out->scan_work, c->work.p, sizeof out->scan_work, hipMemcpyDeviceToHost));
    for (int k = 0; k < RT_SCAN_KINDS; k++) { out->scan_ms[k] = c->ms[k]; out->scan_launches[k] = c->launches[k]; }
    // how the calls in flight were dealt onto hardware queues (RT_OPT_FRAMES_IN_FLIGHT > 1, after the first render call)
    out->lanes = (uint32_t)c->lanes.size();
    out->lane_hw_queues_found = (uint32_t)c->lane_queues;
    {
        std::vector<int> seen;
        int shared = 0;
        for (Lane* L : c->lanes) {
            for (int q : {L->main_queue, L->side_queue}) {
                if (q < 0) continue;
                if (std::find(seen.begin(), seen.end(), q) != seen.end()) shared = 1;
                seen.push_back(q);
            }
        }
        out->lanes_share_a_queue = c->lanes.empty() ? 0 : shared;
        out->caller_stream_shares_a_queue = (c->lanes.empty() || !c->caller_probed) ? -1
            : (c->caller_queue >= 0 && std::find(seen.begin(), seen.end(), c->caller_queue) != seen.end() ? 1 : 0);
    }
    out->n_triangles = c->n_regular_real + c->hs.n_irregular; out->n_spheres = c->hs.n_spheres;
    out->n_groups = c->hs.n_groups; out->n_irregular = c->hs.n_irregular;
    out->filter_form = c->robust ? 2 : 1;
    return 0;
}

int rt_reset_stats(rt_ctx* c) {
    if (!c) return fail(RT_ERR_ARG, "rt_reset_stats: null ctx");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_all(c); if (rc) return rc; }
    drain_spans(c);
    for (int k = 0; k <= RT_WORK_KINDS; k++) { c->ms[k] = 0; c->launches[k] = 0; }
    for (int k = 0; k < 3; k++) c->tally_carry[k] = 0;
    HIP_TRY(hipMemset(c->counters.p, 0, sizeof(DevCounters)));
    HIP_TRY(hipMemset(c->work.p, 0, RT_WORK_KINDS * RT_WORK_FIELDS * sizeof(unsigned long long)));
    if (c->bvh_tally.p) HIP_TRY(hipMemset(c->bvh_tally.p, 0, 2 * (size_t)c->aux_grid * 4 * sizeof(unsigned long long)));
    for (Lane* L : c->lanes)
        if (L->bvh_tally.p) HIP_TRY(hipMemset(L->bvh_tally.p, 0, 2 * (size_t)c->aux_grid * 4 * sizeof(unsigned long long)));
    return 0;
}

int rt_device_info(rt_ctx* c, int32_t* compute_units, int32_t* clock_khz) {
    if (!c) return fail(RT_ERR_ARG, "rt_device_info: null ctx");
    if (compute_units) *compute_units = c->n_cu;
    if (clock_khz) *clock_khz = c->clock_khz;
    return 0;
}

void rt_free(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    c->d_scene.release(); c->tri_filt.release(); c->tri_exact.release(); c->spheres.release(); c->sphere_filt.release(); c->groups.release();
    c->xforms.release(); c->xf_identity.release(); c->obj_kind.release(); c->obj_slot.release();
    c->obj_material.release(); c->obj_transform.release(); c->obj_interp.release(); c->obj_geom.release();
    c->obj_normals.release(); c->materials.release(); c->lights.release(); c->own_frame.release();
    c->counters.release(); c->work.release();
    c->ws_block.release();
    if (c->hint) (void)hipHostFree(c->hint);
    drop_lanes(c);
    for (auto st : c->lane_streams) (void)hipStreamDestroy(st);
    for (auto e : c->ev_hit) (void)hipEventDestroy(e);
    for (auto e : c->ev_phong) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    c->bvh_nodes.release(); c->bvh_wide.release(); c->bvh_groups.release(); c->bvh_spill.release(); c->bvh_tally.release();
    c->eye_a.release(); c->dir_rinv.release(); c->light_dir_row.release(); c->cone_camera.release();
    c->bound_lights.release(); c->light_frame.release(); c->sphere_any.release(); c->plane_any.release();
    c->cl_camera.release(); c->cl_lights.release(); c->cl_sphere.release(); c->cl_normal.release();
    c->graze_slots.release(); c->graze_start.release();
    delete c;
}

}  // extern "C"
