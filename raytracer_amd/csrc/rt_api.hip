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

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct Vec { double x, y, z; };
Vec vsub(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
double vdot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec vcross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double vlen(Vec a) { return std::sqrt(vdot(a, a)); }

// M * (p, 1) without the perspective divide: only used for bounds.
Vec affine_point(const double* m, Vec p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
Vec linear_dir(const double* m, Vec v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

struct Box {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    void add(Vec p) {
        double c[3] = {p.x, p.y, p.z};
        for (int i = 0; i < 3; i++) {
            if (c[i] < lo[i]) lo[i] = c[i];
            if (c[i] > hi[i]) hi[i] = c[i];
        }
    }
    Vec corner(int k) const { return {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]}; }
};

double max_abs(Vec p) { return std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))); }

// body(part, begin, end) over [0, n) cut into equal parts, one host thread per part; a single
// inline call when n is small.  Returns the number of parts (indices 0 .. parts-1 were used).
inline size_t host_parts(size_t n) {
    if (n < 65536) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min<size_t>(16, std::max(1u, hw));
}
template <class F> void parallel_ranges(size_t n, size_t parts, F&& body) {
    if (parts <= 1) { body((size_t)0, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    const size_t per = (n + parts - 1) / parts;
    for (size_t k = 0; k < parts; k++)
        pool.emplace_back([&body, k, per, n]() { body(k, std::min(n, k * per), std::min(n, (k + 1) * per)); });
    for (auto& th : pool) th.join();
}

template <class T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }          // (a buffer that goes out of scope on an error path is not leaked)
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    hipError_t upload(const T* src, size_t count) {
        hipError_t e = alloc(src ? count : 0);
        if (e != hipSuccess || !src || count == 0) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t upload(const std::vector<T>& v) { return upload(v.data(), v.size()); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
};

}  // namespace

// Workspace of rt_render_tile_noalloc: device memory the library holds from the moment it is
// loaded until the process ends, on every GPU it runs on, so that the call itself allocates
// nothing.  Host code never uses the array's name as an address: rt_upload_scene resolves the
// current device's copy once (hipGetSymbolAddress) and everything goes through that pointer.
#define RT_MAX_FRAMES_IN_FLIGHT 8
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
    uint32_t* hit_offsets = nullptr;                   // [aux_grid + 1]: k_hit_offsets -> k_hit
    // [RT_FLAGS_HEADER + rays]: the pair list's fill counters and the task counter in front of the per-ray
    // flags, so that one fill clears them all before a launch; one set per kind of query
    // ([0] nearest hit, [1] shadow) so that a level's shadow launches can run beside the next
    // level's nearest-hit launches
    uint8_t* ray_flags[2] = {nullptr, nullptr};
    size_t ray_flags_n[2] = {0, 0};
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
    int auto_robust = 0;                  // the form chosen from the scene's own numbers
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
    // RT_OPT_BVH_PLANE_GUARD: behind every LBVH traversal, the scan that asks the in-plane-origin question alone.
    // Mode 2 (default) decides once per context, when the LBVH is first switched on: the frame's rays are the same
    // in every render call (camera, size and depth belong to the scene), so one whole frame traced with the scan on
    // tells whether any pair of it takes that branch at all; only then is the scan kept.
    int bvh_guard_mode = 2;
    int bvh_guard = 1;                    // the scan runs (mode 1, or mode 2 after a calibration frame that found such pairs)
    bool bvh_guard_calibrated = false;
    unsigned long long bvh_guard_found = 0;   // in-plane-origin pairs of the calibration frame
    double bvh_guard_calibration_s = 0;
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
    return l;
}

// Cuts a workspace for passes of `cap` pixels out of the block at `base` (256-byte aligned
// slices) and returns the bytes it takes; base == nullptr only measures.  The pair lists hold
// `pairs_per_ray` entries per ray of the largest launch of their kind (a few are typical; a
// list that runs full sends its rays through the full reference-precision scan instead), at
// least `min_pairs`, at most 2^28.
#define RT_FLAGS_HEADER (RT_PAIR_LISTS * 128 + 16)     // bytes in front of a workspace's ray flags (see launch_trace)
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
    t.hit_offsets = (uint32_t*)take(((size_t)aux_grid + 1) * sizeof(uint32_t));
    const size_t max_rays[2] = {cap, cap * (size_t)std::max(1, n_lights)};
    for (int m = 0; m < 2; m++) {
        t.ray_flags_n[m] = max_rays[m] + RT_FLAGS_HEADER;
        t.ray_flags[m] = (uint8_t*)take(t.ray_flags_n[m]);
        t.pair_cap[m] = std::min<size_t>(std::max<size_t>(max_rays[m] * pairs_per_ray, min_pairs), (size_t)1 << 28);
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
    // (the last entry of hit_offsets counts finished blocks and must start at zero)
    e = hipMemset(w.hit_offsets, 0, ((size_t)c->aux_grid + 1) * sizeof(uint32_t));
    if (e != hipSuccess) {
        c->ws_block.release();
        return fail(RT_ERR_DEVICE, std::string("workspace: ") + hipGetErrorString(e));
    }
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
        e = hipMemset(L->ws.hit_offsets, 0, ((size_t)c->aux_grid + 1) * sizeof(uint32_t));
        if (e != hipSuccess) return give_up(RT_ERR_DEVICE, std::string("frames in flight: ") + hipGetErrorString(e));
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
int blocks_for(const rt_ctx* c, size_t items) {
    const size_t blocks = (items + 4 * RT_BLOCK - 1) / (4 * RT_BLOCK);
    return (int)std::min<size_t>((size_t)c->aux_grid, std::max<size_t>((size_t)c->n_cu, blocks));
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
    // (the header in front of the flags: RT_PAIR_LISTS fill counters on a 128-byte line each, then 16 bytes with
    // the task counter and "some ray is flagged")
    A.pair_count = (unsigned long long*)W.ray_flags[MODE];
    A.task_counter = (unsigned int*)(W.ray_flags[MODE] + RT_FLAGS_HEADER - 8);
    A.any_flagged = A.task_counter + 1;
    A.ray_flags = W.ray_flags[MODE] + RT_FLAGS_HEADER; A.pairs = W.pairs[MODE];
    A.pair_cap = (unsigned int)W.pair_cap[MODE]; A.counters = c->counters.p; A.scan_all = c->force_exact;
    // as many sub-lists as leave each at least 65536 entries (a flush appends up to 4096)
    A.pair_lists = 1;
    while (A.pair_lists < RT_PAIR_LISTS && W.pair_cap[MODE] / (2 * A.pair_lists) >= 65536) A.pair_lists *= 2;
    A.pair_seg = (unsigned int)(W.pair_cap[MODE] / A.pair_lists);
    const bool work = c->work_counters && c->work.p;
    A.work = work ? c->work.p : nullptr;
    // (no queue of the pass is longer than max_rays; beyond the expected length a flag may be left over from an earlier
    // launch: such a ray takes the full reference-precision scan, which costs time, not pixels)
    HIP_TRY(hipMemsetAsync(W.ray_flags[MODE], 0, std::min(W.ray_flags_n[MODE], std::min(max_rays, est_rays + est_rays / 4 + 4096) + RT_FLAGS_HEADER), st));
    bool guard_scan = false;              // LBVH mode: the in-plane-origin scan ran and listed pairs
    if (!c->force_exact && c->hs.n_regular > 0) {
        int scan_kind = MODE == 0 ? 0 : 5;
        if (!c->use_bvh) {
            if (MODE == 0 && A.cone_camera) scan_kind = 1;
            else if (MODE == 0 && A.sphere_any) scan_kind = 4;
            else if (MODE == 1 && A.bound_lights) scan_kind = (c->h_lights.empty() || c->h_lights[0].is_point) ? 2 : 3;
        }
        Timed t(c, st, scan_kind, events);
        if (c->use_bvh) {
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
            guard_scan = c->bvh_guard && ((MODE == 0 && camera_rays) ? c->n_eye_plane > 0
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
    if (!c->force_exact && c->hs.n_regular > 0 && (!c->use_bvh || guard_scan)) {      // (the LBVH traversal itself lists no pairs)
        hipLaunchKernelGGL(k_exact_pairs<MODE>, dim3(pair_grid), dim3(RT_BLOCK), 0, st, dsc,
                           (const TriExact*)c->tri_exact.p, A);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_brute<MODE>, dim3(ray_grid), dim3(RT_BLOCK), 0, st, dsc, (const TriExact*)c->tri_exact.p, A);
    HIP_TRY(hipGetLastError());
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
    HIP_TRY(hipMemsetAsync(ws.counts, 0, (levels + 2) * sizeof(uint32_t), st));
    // FP32 scan copies of the rays for the bounded scans (RayQueue::scan): written by k_primary / k_hit when the
    // workspace has room for them and the scans that read them will run
    const bool scan = keeps_scan_copies(c, ws), scan_shadow = scan && keeps_shadow_copies(c, ws);
    RayQueue q[2] = {queue_of(ws, 0, scan), queue_of(ws, 1, scan)};
    hipLaunchKernelGGL(k_primary, dim3(blocks_for(c, (size_t)n_pix)), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q[0], ws.keys[0],
                       ws.pixel_of_ray, c->frame, ws.counts);
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
        ha.n_rays = ws.counts + k; ha.hit_offsets = ws.hit_offsets; ha.lvl = lvl; ha.aux = aux;
        ha.occluded = occl; ha.occl_words = ws.occl_words;
        ha.shadow_scan = scan_shadow ? ws.shadow_scan[cur] : nullptr; ha.shadow_stride = (uint32_t)ws.cap;
        ha.light_frame = c->light_frame.p;
        hipLaunchKernelGGL(k_hit_offsets, dim3(level_grid), dim3(RT_BLOCK), 0, st, (const unsigned long long*)ws.keys[cur],
                           (const uint32_t*)(ws.counts + k), ws.hit_offsets, ws.hit_offsets + c->aux_grid, ws.counts + k + 1);
        hipLaunchKernelGGL(k_hit, dim3(level_grid), dim3(RT_BLOCK), 0, st, ha);      // (the same grid: a block's range of the queue is the one it counted)
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
        hipLaunchKernelGGL(k_phong, dim3(hits_grid), dim3(RT_BLOCK), 0, sb, pa);
        HIP_TRY(hipGetLastError());
        if (overlap) HIP_TRY(hipEventRecord(ev_phong[k], sb));
    }
    // the fold (and whatever the caller enqueues next) comes after everything on the side stream
    if (overlap) HIP_TRY(hipStreamWaitEvent(st, ev_phong[levels - 1], 0));
    for (int k = levels - 1; k >= 0; k--) {
        FoldArgs fa{};
        fa.sc = dsc; fa.lvl = level_of(ws, k); fa.parent_lvl = level_of(ws, k > 0 ? k - 1 : 0);
        fa.cap = (uint32_t)ws.cap; fa.n_hits = ws.counts + k + 1; fa.level = k;
        fa.pixel_of_ray = ws.pixel_of_ray; fa.frame = c->frame;
        hipLaunchKernelGGL(k_fold, dim3(blocks_for(c, expected(k + 1))), dim3(RT_BLOCK), 0, st, fa);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, st, dsc, (const uint32_t*)ws.counts, levels, c->counters.p);
    HIP_TRY(hipGetLastError());
    // this pass's queue lengths for the next pass of its size (pinned memory: the copy is a device-side write)
    if (c->grid_hints && ws.hint && levels + 2 <= 72) {
        ws.hint->n_pix = n_pix;
        HIP_TRY(hipMemcpyAsync(ws.hint->counts, ws.counts, (size_t)(levels + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    return 0;
}

// Rows y_start + j*stride_rows + [0, tile_rows) for j = 0 .. n_tiles-1 (clipped to the frame).
int render_rows(rt_ctx* c, int y_start, int tile_rows, int stride_rows, int n_tiles, hipStream_t st) {
    HIP_TRY(hipSetDevice(c->device));
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

// Orders the triangles for the chosen form of the filter (regular ones by transform group and
// dominant axis, then the irregular ones), builds the groups, and uploads the triangle records,
// the groups and the scene header.  Called by rt_upload_scene and by RT_OPT_FILTER_FORM.
// Smallest sphere around a triangle (centre in its plane): the circumscribed circle, or the
// circle over the longest edge when the angle opposite to it is not acute.
void bounding_sphere(const TriExact& ex, Vec* centre, double* radius) {
    const Vec a{ex.v1[0], ex.v1[1], ex.v1[2]};
    const Vec ab{ex.e2[0], ex.e2[1], ex.e2[2]}, ac{ex.e1[0], ex.e1[1], ex.e1[2]};   // v2 - v1, v3 - v1
    const Vec bc = vsub(ac, ab);
    const double lab = vdot(ab, ab), lac = vdot(ac, ac), lbc = vdot(bc, bc);
    Vec rel;        // centre - a
    if (lab >= lac + lbc) rel = {ab.x * .5, ab.y * .5, ab.z * .5};
    else if (lac >= lab + lbc) rel = {ac.x * .5, ac.y * .5, ac.z * .5};
    else if (lbc >= lab + lac) rel = {(ab.x + ac.x) * .5, (ab.y + ac.y) * .5, (ab.z + ac.z) * .5};
    else {
        const Vec n = vcross(ab, ac);
        const double nn = vdot(n, n);
        const Vec t1 = vcross(n, ab), t2 = vcross(ac, n);
        rel = {(lac * t1.x + lab * t2.x) / (2 * nn), (lac * t1.y + lab * t2.y) / (2 * nn), (lac * t1.z + lab * t2.z) / (2 * nn)};
    }
    if (!(std::isfinite(rel.x) && std::isfinite(rel.y) && std::isfinite(rel.z))) rel = {(ab.x + ac.x) / 3, (ab.y + ac.y) / 3, (ab.z + ac.z) / 3};
    *centre = {a.x + rel.x, a.y + rel.y, a.z + rel.z};
    // the radius is measured, not derived: whatever the centre, the sphere holds the three vertices
    const Vec da = rel, db = vsub(rel, ab), dc = vsub(rel, ac);
    *radius = std::sqrt(std::max(vdot(da, da), std::max(vdot(db, db), vdot(dc, dc))));
}

// Cone from `apex` around the sphere (centre, radius): unit axis and cosine of the half-angle,
// lowered by the rounding of the FP32 test; cos = -2 (every direction passes) when the apex is
// inside the sphere or nearly so.
void bounding_cone(Vec apex, Vec centre, double radius, float out[4]) {
    const Vec d = vsub(centre, apex);
    const double dist = vlen(d);
    out[0] = 0.f; out[1] = 0.f; out[2] = 1.f; out[3] = -2.f;
    if (!(dist > 0) || !std::isfinite(dist) || !(radius < 0.999 * dist)) return;
    const double sn = radius / dist;
    out[0] = (float)(d.x / dist); out[1] = (float)(d.y / dist); out[2] = (float)(d.z / dist);
    out[3] = (float)(std::sqrt(1.0 - sn * sn) - 4e-6);
}

int install_filter(rt_ctx* c, int robust) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());          // no launch may still read the records replaced here
    const std::vector<HostTri>& tris = c->h_tris;
    auto regular = [&](const HostTri& t) { return robust ? t.ok_robust : t.ok_static; };
    // regular triangles by (transform, dominant axis), irregular ones by transform, file order
    // within a key: a counting sort, the keys are few
    std::vector<int> order_reg, order_irr;
    {
        int n_xf = 1;
        for (const HostTri& t : tris) n_xf = std::max(n_xf, t.xf + 1);
        std::vector<size_t> n_reg((size_t)n_xf * 3 + 1, 0), n_irr((size_t)n_xf + 1, 0);
        for (const HostTri& t : tris) {
            if (regular(t)) n_reg[(size_t)t.xf * 3 + t.axis + 1]++; else n_irr[(size_t)t.xf + 1]++;
        }
        for (size_t k = 1; k < n_reg.size(); k++) n_reg[k] += n_reg[k - 1];
        for (size_t k = 1; k < n_irr.size(); k++) n_irr[k] += n_irr[k - 1];
        order_reg.resize(n_reg.back()); order_irr.resize(n_irr.back());
        for (int k = 0; k < (int)tris.size(); k++) {
            const HostTri& t = tris[k];
            if (regular(t)) order_reg[n_reg[(size_t)t.xf * 3 + t.axis]++] = k; else order_irr[n_irr[(size_t)t.xf]++] = k;
        }
    }
    // Within a (transform, axis) run the scans ask their bounding questions of clusters of 64 and batches of 4
    // consecutive slots, so consecutive slots should be neighbours in space.  A mesh written patch by patch (every
    // mesh of the reference is) has that in file order already, and file order is kept: on the teapot and on
    // campanile + teddy it measured 3-5 % faster than the order along a space-filling curve through the centroids
    // (63-bit Morton codes in the run's own centroid box).  The curve takes over only where it is clearly better -- the
    // clusters' extents (largest distance of a member's centroid from the cluster's mean centroid) add up to less than
    // 0.7 of file order's -- i.e. for meshes whose faces come in no spatial order.  Nothing else depends on the order
    // of the slots (ties between hits go by object index, which travels with the record).
    // RAYTRACER_AMD_SLOT_ORDER=file / curve forces one or the other (A/B runs).
    {
        const char* env = std::getenv("RAYTRACER_AMD_SLOT_ORDER");
        const bool file_order = env && std::string(env) == "file", force_curve = env && std::string(env) == "curve";
        auto extent_sum = [&](const int* idx, size_t n) {
            double sum = 0;
            for (size_t b = 0; b < n; b += 64) {
                const size_t m = std::min<size_t>(64, n - b);
                double mean[3] = {0, 0, 0};
                for (size_t k = 0; k < m; k++) for (int q = 0; q < 3; q++) mean[q] += tris[(size_t)idx[b + k]].ex.v1[q] + (tris[(size_t)idx[b + k]].ex.e1[q] + tris[(size_t)idx[b + k]].ex.e2[q]) * (1.0 / 3.0);
                for (int q = 0; q < 3; q++) mean[q] /= (double)m;
                double far2 = 0;
                for (size_t k = 0; k < m; k++) {
                    double d2 = 0;
                    for (int q = 0; q < 3; q++) {
                        const double d = tris[(size_t)idx[b + k]].ex.v1[q] + (tris[(size_t)idx[b + k]].ex.e1[q] + tris[(size_t)idx[b + k]].ex.e2[q]) * (1.0 / 3.0) - mean[q];
                        d2 += d * d;
                    }
                    far2 = std::max(far2, d2);
                }
                sum += std::sqrt(far2);
            }
            return sum;
        };
        auto centroid = [&](int k, int a) { return tris[(size_t)k].ex.v1[a] + (tris[(size_t)k].ex.e1[a] + tris[(size_t)k].ex.e2[a]) * (1.0 / 3.0); };
        auto spread21 = [](unsigned long long v) {          // 21 bits -> every third bit
            v &= 0x1fffffull;
            v = (v | v << 32) & 0x1f00000000ffffull;
            v = (v | v << 16) & 0x1f0000ff0000ffull;
            v = (v | v << 8) & 0x100f00f00f00f00full;
            v = (v | v << 4) & 0x10c30c30c30c30c3ull;
            v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        size_t r = 0;
        std::vector<std::pair<unsigned long long, int>> keyed;
        while (!file_order && r < order_reg.size()) {
            const int x = tris[(size_t)order_reg[r]].xf, a = tris[(size_t)order_reg[r]].axis;
            size_t e = r;
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            while (e < order_reg.size() && tris[(size_t)order_reg[e]].xf == x && tris[(size_t)order_reg[e]].axis == a) {
                for (int q = 0; q < 3; q++) { const double cq = centroid(order_reg[e], q); lo[q] = std::min(lo[q], cq); hi[q] = std::max(hi[q], cq); }
                e++;
            }
            double ext = 0;
            for (int q = 0; q < 3; q++) ext = std::max(ext, hi[q] - lo[q]);
            if (e - r > 64 && ext > 0 && std::isfinite(ext)) {
                keyed.resize(e - r);
                const double scale = 2097151.0 / ext;               // one scale for the three axes: cells are cubes
                parallel_ranges(e - r, host_parts(e - r), [&](size_t, size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; k++) {
                        unsigned long long code = 0;
                        for (int q = 0; q < 3; q++) {
                            const double v = (centroid(order_reg[r + k], q) - lo[q]) * scale;
                            const unsigned long long cell = (unsigned long long)std::min(2097151.0, std::max(0.0, v));
                            code |= spread21(cell) << q;
                        }
                        keyed[k] = {code, order_reg[r + k]};
                    }
                });
                std::sort(keyed.begin(), keyed.end());              // (ties: by index, i.e. file order)
                std::vector<int> curve(keyed.size());
                for (size_t k = 0; k < keyed.size(); k++) curve[k] = keyed[k].second;
                if (force_curve || extent_sum(curve.data(), curve.size()) < 0.7 * extent_sum(&order_reg[r], e - r))
                    for (size_t k = 0; k < curve.size(); k++) order_reg[r + k] = curve[k];
            }
            r = e;
        }
    }
    // Every (transform, axis) range starts at a multiple of 64 slots, so that clusters of 64 consecutive
    // slots never straddle two ranges: the slots between a range's last triangle and the next multiple of
    // 64 are holes (order_reg = -1).  No scan visits a hole (the ranges end at their last triangle); the
    // kernels that walk whole groups (full reference-precision scan, LBVH) skip them by TriExact::hole.
    c->n_regular_real = (int)order_reg.size();
    std::vector<int> slot_xf, slot_axis;             // per slot, holes included
    {
        std::vector<int> padded;
        padded.reserve(order_reg.size() + 192);
        size_t r = 0;
        while (r < order_reg.size()) {
            const int x = tris[order_reg[r]].xf, a = tris[order_reg[r]].axis;
            while (r < order_reg.size() && tris[order_reg[r]].xf == x && tris[order_reg[r]].axis == a) {
                padded.push_back(order_reg[r++]); slot_xf.push_back(x); slot_axis.push_back(a);
            }
            while (padded.size() % 64) { padded.push_back(-1); slot_xf.push_back(x); slot_axis.push_back(a); }
        }
        order_reg.swap(padded);
    }
    const int NR = (int)order_reg.size(), NI = (int)order_irr.size();
    const size_t parts = host_parts(tris.size());
    std::vector<v16f> filt((size_t)NR + 1);          // + one readable padding record
    std::memset(&filt[NR], 0, sizeof(v16f));
    std::vector<TriExact> exact((size_t)NR + NI);
    std::vector<int32_t>& slot = c->h_slot;
    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            if (order_reg[k] < 0) {
                for (int j = 0; j < 16; j++) filt[k][j] = 0.f;
                std::memset(&exact[k], 0, sizeof(TriExact));
                exact[k].hole = 1; exact[k].obj = -1;
                continue;
            }
            const HostTri& t = tris[order_reg[k]];
            const float* f = robust ? t.f_robust : t.f_static;
            for (int j = 0; j < 16; j++) filt[k][j] = f[j];
            exact[k] = t.ex;
            exact[k].hole = 0;
            slot[t.ex.obj] = (int32_t)k;
        }
    });
    for (int k = 0; k < NI; k++) {
        const HostTri& t = tris[order_irr[k]];
        exact[(size_t)NR + k] = t.ex;
        exact[(size_t)NR + k].hole = 0;
        slot[t.ex.obj] = NR + k;
    }
    std::vector<Group> groups;
    {
        int r = 0, q = 0;
        while (r < NR || q < NI) {
            int xr = r < NR ? slot_xf[r] : INT32_MAX;
            int xq = q < NI ? tris[order_irr[q]].xf : INT32_MAX;
            int x = std::min(xr, xq);
            Group G{};
            G.xf = x; G.identity = c->h_xf_ident[x];
            G.q[0] = c->h_q0[x]; G.q[1] = c->h_q1[x];
            for (int a = 0; a < 3; a++) {
                G.fax[a] = r;
                G.fcnt[a] = 0;
                while (r < NR && slot_xf[r] == x && slot_axis[r] == a) { if (order_reg[r] >= 0) G.fcnt[a]++; r++; }
            }
            G.fax[3] = r;
            G.ibeg = NR + q;
            while (q < NI && tris[order_irr[q]].xf == x) q++;
            G.iend = NR + q;
            for (int k2 = G.fax[0]; k2 < G.fax[3]; k2++) exact[k2].grp = (int32_t)groups.size();
            for (int k2 = G.ibeg; k2 < G.iend; k2++) exact[k2].grp = (int32_t)groups.size();
            groups.push_back(G);
        }
    }

    c->h_groups = groups;
    c->all_identity = true;
    for (const Group& g : groups) if (!g.identity) c->all_identity = false;
    c->h_centroid_box.assign(6 * groups.size(), 0.f);
    for (size_t gi = 0; gi < groups.size(); gi++) {
        float* bx = &c->h_centroid_box[6 * gi];
        for (int a = 0; a < 3; a++) { bx[a] = FLT_MAX; bx[3 + a] = -FLT_MAX; }
        for (int k2 = groups[gi].fax[0]; k2 < groups[gi].fax[3]; k2++) {
            const TriExact& ex = exact[k2];
            if (ex.hole) continue;
            for (int a = 0; a < 3; a++) {
                float cc = (float)(ex.v1[a] + (ex.e1[a] + ex.e2[a]) * (1.0 / 3.0));
                bx[a] = std::min(bx[a], cc); bx[3 + a] = std::max(bx[3 + a], cc);
            }
        }
    }
    // a hierarchy built over the previous order is void
    c->bvh_built = false;
    c->bvh_nodes.release(); c->bvh_groups.release();

    // Constants shared by all rays of a launch, in double precision like the reference's own
    // dot products (raytracer.cpp:316-317): a = -(n . (eye - v1)) for camera rays (the eye in the
    // group's object space), 1 / (n . dir) for the rays towards each directional light
    // (world-space direction against the untransformed normal, raytracer.cpp:553).
    {
        std::vector<float> eye_a((size_t)NR + 1, 0.f);
        std::vector<float> cone_cam(4 * ((size_t)NR + 8), 0.f);      // readable a batch past the end
        for (size_t k = (size_t)NR; k < (size_t)NR + 8; k++) cone_cam[4 * k + 3] = -2.f;
        // bounding sphere of every regular triangle: what the reference can accept lies within
        // box_pad of the triangle; the rest covers the rounding of centre and radius
        std::vector<double> sphere(4 * (size_t)NR);
        const double sphere_pad = 64.0 * 1.1920928955078125e-7 * c->scene_bound;
        parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                Vec centre; double radius;
                bounding_sphere(exact[k], &centre, &radius);
                sphere[4 * k] = centre.x; sphere[4 * k + 1] = centre.y; sphere[4 * k + 2] = centre.z;
                sphere[4 * k + 3] = (radius + exact[k].box_pad) * (1 + 1e-6) + sphere_pad;
            }
        });
        // rays with nothing in common: sphere as (centre, rho^2 - |centre|^2 + rounding of the expanded
        // distance test, 64 eps R^2 with R the group's bound on |origin| and |vertex|), plane as (n, d0)
        std::vector<float> sphere_any(4 * ((size_t)NR + 8), 0.f), plane_any(4 * ((size_t)NR + 8), 0.f);
        for (size_t k = (size_t)NR; k < (size_t)NR + 8; k++) { sphere_any[4 * k + 3] = -INFINITY; plane_any[4 * k + 3] = INFINITY; }
        const Vec eye{c->hs.cam_eye[0], c->hs.cam_eye[1], c->hs.cam_eye[2]};
        for (const Group& G : groups) {
            const Vec o = G.identity ? eye : affine_point(&c->h_xf[48 * (size_t)G.xf + 16], eye);
            const size_t first = (size_t)G.fax[0], count = (size_t)(G.fax[3] - G.fax[0]);
            const double Rg = (double)G.q[0].e_a / (16.0 * 1.1920928955078125e-7) * (1 + 1e-3);
            const double slack_any = 64.0 * 1.1920928955078125e-7 * Rg * Rg;
            parallel_ranges(count, host_parts(count), [&](size_t, size_t k0, size_t k1) {
                for (size_t k = first + k0; k < first + k1; k++) {
                    const TriExact& ex = exact[k];
                    {
                        const double cx = sphere[4 * k], cy = sphere[4 * k + 1], cz = sphere[4 * k + 2], rho = sphere[4 * k + 3];
                        const double kk = rho * rho - (cx * cx + cy * cy + cz * cz) + slack_any;
                        sphere_any[4 * k] = (float)cx; sphere_any[4 * k + 1] = (float)cy; sphere_any[4 * k + 2] = (float)cz;
                        float kf = std::isfinite(kk) ? (float)kk : INFINITY;
                        if (std::isfinite(kf)) kf += std::fabs(kf) * 1e-6f;
                        if (!std::isfinite(sphere_any[4 * k]) || !std::isfinite(sphere_any[4 * k + 1]) || !std::isfinite(sphere_any[4 * k + 2])) {
                            sphere_any[4 * k] = sphere_any[4 * k + 1] = sphere_any[4 * k + 2] = 0.f; kf = INFINITY;
                        }
                        sphere_any[4 * k + 3] = kf;
                        for (int q = 0; q < 4; q++) plane_any[4 * k + q] = filt[k][q];
                    }
                    const Vec w0{o.x - ex.v1[0], o.y - ex.v1[1], o.z - ex.v1[2]};
                    eye_a[k] = (float)(-(ex.n[0] * w0.x + ex.n[1] * w0.y + ex.n[2] * w0.z));
                    bounding_cone(o, {sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]}, sphere[4 * k + 3], &cone_cam[4 * k]);
                    // the in-plane-origin case (raytracer.cpp:319-325) ignores where the triangle is
                    if (!(std::fabs(eye_a[k]) > G.q[0].a_zero)) cone_cam[4 * k + 3] = -2.f;
                }
            });
        }
        c->n_eye_plane = 0;
        for (const Group& G : groups)
            for (int k = G.fax[0]; k < G.fax[3]; k++)
                if (!exact[(size_t)k].hole && !(std::fabs(eye_a[(size_t)k]) > G.q[0].a_zero)) c->n_eye_plane++;
        // Per light (shadow rays meet the untransformed triangles in world space, raytracer.cpp:553):
        // point light: cone from the light, and the light's distance from the triangle's plane;
        // directional light: disc in the plane perpendicular to the light's direction, coordinates
        // about the centre of the scene.
        // (16 bytes per triangle and light: beyond 2 GiB the table is left out and shadow rays take the
        // triangle test for every pair)
        if (c->h_lights.size() * ((size_t)NR + 8) * 16 > ((size_t)2 << 30)) {
            c->bound_lights.release(); c->light_frame.release();
        } else {
            const size_t NL = c->h_lights.size(), stride = (size_t)NR + 8;
            std::vector<float> bound(NL * stride * 4, 0.f);
            std::vector<double> frame(std::max<size_t>(NL, 1) * 8, 0.0);
            const double half = c->scene_bound;                       // bound of |origin| and |vertex| for every ray and triangle
            for (size_t l = 0; l < NL; l++) {
                const DevLight& lt = c->h_lights[l];
                float* B = &bound[l * stride * 4];
                for (size_t k = (size_t)NR; k < stride; k++) { B[4 * k + 2] = lt.is_point ? 1.f : -INFINITY; B[4 * k + 3] = lt.is_point ? 2.f : 0.f; }
                if (lt.is_point) {
                    const Vec Lp{lt.pos[0], lt.pos[1], lt.pos[2]};
                    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                        for (size_t k = k0; k < k1; k++) {
                            bounding_cone(Lp, {sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]}, sphere[4 * k + 3], &B[4 * k]);
                        }
                    });
                } else {
                    // Light::lightRay: direction = (-pos).normalize() (raytracer.cpp:113-117)
                    const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
                    const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
                    const Vec dir{x * inv, y * inv, z * inv};
                    // an orthonormal pair perpendicular to dir
                    Vec t = std::fabs(dir.x) <= std::fabs(dir.y) && std::fabs(dir.x) <= std::fabs(dir.z) ? Vec{1, 0, 0}
                          : (std::fabs(dir.y) <= std::fabs(dir.z) ? Vec{0, 1, 0} : Vec{0, 0, 1});
                    Vec e1 = vcross(dir, t);
                    const double l1 = vlen(e1);
                    e1 = {e1.x / l1, e1.y / l1, e1.z / l1};
                    const Vec e2 = vcross(dir, e1);
                    const bool usable = std::isfinite(e1.x) && std::isfinite(e1.y) && std::isfinite(e1.z) && std::isfinite(e2.x) &&
                                        std::isfinite(e2.y) && std::isfinite(e2.z);
                    double* F = &frame[8 * l];
                    F[0] = e1.x; F[1] = e1.y; F[2] = e1.z; F[3] = e2.x; F[4] = e2.y; F[5] = e2.z; F[6] = 0.0; F[7] = 0.0;
                    // |coordinates| <= half; rounding of the expanded form, see filter_range
                    const double rmax = half * (1 + 1e-3);
                    const double slack = 16.0 * 1.1920928955078125e-7 * rmax * rmax;
                    parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                        for (size_t k = k0; k < k1; k++) {
                            const Vec ctr{sphere[4 * k], sphere[4 * k + 1], sphere[4 * k + 2]};
                            const double cu = vdot(ctr, e1), cv = vdot(ctr, e2), rho = sphere[4 * k + 3];
                            double kk = rho * rho + 3.0 * 1.1920928955078125e-7 * rho * rmax - cu * cu - cv * cv + slack;
                            if (!usable || !std::isfinite(kk)) kk = INFINITY;
                            B[4 * k] = (float)(-2.0 * cu); B[4 * k + 1] = (float)(-2.0 * cv);
                            B[4 * k + 2] = (kk == INFINITY) ? INFINITY : (float)kk + std::fabs((float)kk) * 1e-6f;
                            B[4 * k + 3] = 0.f;
                            if (!std::isfinite(B[4 * k]) || !std::isfinite(B[4 * k + 1])) { B[4 * k] = B[4 * k + 1] = 0.f; B[4 * k + 2] = INFINITY; }
                        }
                    });
                }
            }
            HIP_TRY(c->bound_lights.upload(bound));
            HIP_TRY(c->light_frame.upload(frame));
        }
        std::vector<int32_t> row(std::max<size_t>(c->h_lights.size(), 1), -1);
        std::vector<float> rinv;
        int n_rows = 0;
        for (size_t l = 0; l < c->h_lights.size(); l++) {
            const DevLight& lt = c->h_lights[l];
            if (lt.is_point) continue;
            // Light::lightRay: direction = (-pos).normalize() (raytracer.cpp:113-117, algebra3.h:638-662)
            const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
            const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
            const Vec dir{x * inv, y * inv, z * inv};
            row[l] = n_rows++;
            rinv.resize((size_t)n_rows * ((size_t)NR + 1), 0.f);
            float* out = &rinv[(size_t)row[l] * ((size_t)NR + 1)];
            parallel_ranges((size_t)NR, parts, [&](size_t, size_t k0, size_t k1) {
                for (size_t k = k0; k < k1; k++) {
                    const TriExact& ex = exact[k];
                    out[k] = (float)(1.0 / (ex.n[0] * dir.x + ex.n[1] * dir.y + ex.n[2] * dir.z));
                }
            });
        }
        // ---- one level up: clusters of 64 consecutive slots ----
        {
            const size_t NC = (size_t)NR / 64, NL = c->h_lights.size();
            c->n_clusters = (int)NC;
            std::vector<float> cl_sphere(4 * (NC + 1), 0.f), cl_normal(4 * (NC + 1), 0.f), cl_cam(4 * (NC + 1), 0.f);
            std::vector<float> cl_lights(std::max<size_t>(NL, 1) * (NC + 1) * 4, 0.f);
            std::vector<int32_t> graze_slots, graze_start(std::max<size_t>(NL, 1) * (NC + 1), 0);
            std::vector<double> csph(4 * (NC + 1), 0.0);
            // which group a cluster belongs to (for the eye in that group's space)
            std::vector<int> cl_group(NC + 1, 0);
            for (size_t gi = 0; gi < groups.size(); gi++)
                for (int k = groups[gi].fax[0] / 64; k < groups[gi].fax[3] / 64; k++) cl_group[(size_t)k] = (int)gi;
            parallel_ranges(NC, host_parts(NC * 64), [&](size_t, size_t c0, size_t c1) {
                for (size_t cl = c0; cl < c1; cl++) {
                    // sphere: centre = mean of the members' centres, radius = what holds every member's sphere
                    double mx = 0, my = 0, mz = 0, nx = 0, ny = 0, nz = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        mx += sphere[4 * k]; my += sphere[4 * k + 1]; mz += sphere[4 * k + 2];
                        nx += filt[k][0]; ny += filt[k][1]; nz += filt[k][2];
                    }
                    int members = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) members += exact[k].hole ? 0 : 1;
                    mx /= members; my /= members; mz /= members;             // (a cluster holds at least one triangle)
                    double rad = 0;
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        const double dx = sphere[4 * k] - mx, dy = sphere[4 * k + 1] - my, dz = sphere[4 * k + 2] - mz;
                        // (the radius of the triangle-level test carries the rounding slack of its expanded form)
                        const double rk = std::sqrt((double)sphere_any[4 * k + 3] + (double)sphere_any[4 * k] * sphere_any[4 * k] +
                                                    (double)sphere_any[4 * k + 1] * sphere_any[4 * k + 1] +
                                                    (double)sphere_any[4 * k + 2] * sphere_any[4 * k + 2]);
                        rad = std::max(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::max(sphere[4 * k + 3], rk));
                    }
                    rad = rad * (1 + 1e-5) + sphere_pad;
                    const bool fin = std::isfinite(mx) && std::isfinite(my) && std::isfinite(mz) && std::isfinite(rad);
                    csph[4 * cl] = fin ? mx : 0; csph[4 * cl + 1] = fin ? my : 0; csph[4 * cl + 2] = fin ? mz : 0;
                    csph[4 * cl + 3] = fin ? rad : INFINITY;
                    for (int q = 0; q < 4; q++) cl_sphere[4 * cl + q] = (float)csph[4 * cl + q];
                    if (std::isfinite(cl_sphere[4 * cl + 3])) cl_sphere[4 * cl + 3] *= 1.0f + 1e-6f;
                    // normal cone: mean normal and the sine of the largest angle a member's normal makes with it
                    const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
                    double sd = 1.0;
                    if (nl > 1e-6 && std::isfinite(nl)) {
                        nx /= nl; ny /= nl; nz /= nl;
                        double cmin = 1.0;
                        for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                            const double len = std::sqrt((double)filt[k][0] * filt[k][0] + (double)filt[k][1] * filt[k][1] + (double)filt[k][2] * filt[k][2]);
                            cmin = std::min(cmin, (filt[k][0] * nx + filt[k][1] * ny + filt[k][2] * nz) / len);
                        }
                        sd = cmin > 0 ? std::sqrt(std::max(0.0, 1.0 - cmin * cmin)) + 1e-5 : 1.0;
                        if (!std::isfinite(sd)) sd = 1.0;
                    } else { nx = 0; ny = 0; nz = 1; }
                    cl_normal[4 * cl] = (float)nx; cl_normal[4 * cl + 1] = (float)ny; cl_normal[4 * cl + 2] = (float)nz;
                    cl_normal[4 * cl + 3] = (float)std::min(1.0, sd);
                    // camera rays: the cone from the eye (in the group's space); "every direction" when the eye
                    // lies in a member's plane (the in-plane-origin case ignores where the triangle is)
                    const Group& G = groups[(size_t)cl_group[cl]];
                    const Vec o = G.identity ? eye : affine_point(&c->h_xf[48 * (size_t)G.xf + 16], eye);
                    bounding_cone(o, {csph[4 * cl], csph[4 * cl + 1], csph[4 * cl + 2]}, csph[4 * cl + 3], &cl_cam[4 * cl]);
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++)
                        if (!exact[k].hole && !(std::fabs(eye_a[k]) > G.q[0].a_zero)) cl_cam[4 * cl + 3] = -2.f;
                }
            });
            cl_cam[4 * NC + 3] = -2.f;
            const double half = c->scene_bound;
            for (size_t l = 0; l < NL; l++) {
                const DevLight& lt = c->h_lights[l];
                float* B = &cl_lights[l * (NC + 1) * 4];
                int32_t* GS = &graze_start[l * (NC + 1)];
                const Vec Lp{lt.pos[0], lt.pos[1], lt.pos[2]};
                const double x = -lt.pos[0], y = -lt.pos[1], z = -lt.pos[2];
                const double len = std::sqrt(x * x + y * y + z * z), inv = 1. / len;
                const Vec dir{x * inv, y * inv, z * inv};
                // the light's plane coordinates as the per-triangle table has them
                Vec e1{0, 0, 0}, e2{0, 0, 0};
                bool usable = false;
                if (!lt.is_point) {
                    Vec t = std::fabs(dir.x) <= std::fabs(dir.y) && std::fabs(dir.x) <= std::fabs(dir.z) ? Vec{1, 0, 0}
                          : (std::fabs(dir.y) <= std::fabs(dir.z) ? Vec{0, 1, 0} : Vec{0, 0, 1});
                    e1 = vcross(dir, t);
                    const double l1 = vlen(e1);
                    e1 = {e1.x / l1, e1.y / l1, e1.z / l1};
                    e2 = vcross(dir, e1);
                    usable = std::isfinite(e1.x) && std::isfinite(e1.y) && std::isfinite(e1.z) && std::isfinite(e2.x) &&
                             std::isfinite(e2.y) && std::isfinite(e2.z);
                }
                const double rmax = half * (1 + 1e-3);
                const double slack = 16.0 * 1.1920928955078125e-7 * rmax * rmax;
                const double l1norm = std::fabs(Lp.x) + std::fabs(Lp.y) + std::fabs(Lp.z);
                for (size_t cl = 0; cl < NC; cl++) {
                    const Vec ctr{csph[4 * cl], csph[4 * cl + 1], csph[4 * cl + 2]};
                    const double rho = csph[4 * cl + 3];
                    // Graze list: what the plane guard of any task could name.  Its limit is b_hi * (largest distance
                    // from the task's origins to the light) + 2 a_zero + rounding for a point light -- origins lie
                    // within the scene's bound of the origin of coordinates --, b_hi for a directional light; the
                    // kernel evaluates the same quantities in FP32, whose rounding is allowed for here.
                    GS[cl] = (int32_t)graze_slots.size();
                    for (size_t k = 64 * cl; k < 64 * cl + 64; k++) {
                        if (exact[k].hole) continue;
                        const double nxk = filt[k][0], nyk = filt[k][1], nzk = filt[k][2], d0 = filt[k][3];
                        const double o = lt.is_point ? std::fabs(d0 - (nxk * Lp.x + nyk * Lp.y + nzk * Lp.z))
                                                     : std::fabs(nxk * dir.x + nyk * dir.y + nzk * dir.z);
                        const GroupFilter& q1 = c->h_q1[0];       // (shadow thresholds are the same for every transform)
                        const double far = std::sqrt(3.0) * half + vlen(Lp);
                        const double limit = lt.is_point ? (double)q1.b_hi * far * (1 + 1e-3) + 2.0 * q1.a_zero + 1e-5 * (l1norm + half) + 1e-6
                                                         : (double)q1.b_hi * (1 + 2e-3) + 1e-5;
                        if (!(o > limit)) graze_slots.push_back((int32_t)k);
                    }
                    if (lt.is_point) {
                        bounding_cone(Lp, ctr, rho, &B[4 * cl]);
                    } else {
                        const double cu = vdot(ctr, e1), cv = vdot(ctr, e2);
                        double kk = rho * rho + 3.0 * 1.1920928955078125e-7 * rho * rmax - cu * cu - cv * cv + slack;
                        if (!usable || !std::isfinite(kk)) kk = INFINITY;
                        B[4 * cl] = (float)(-2.0 * cu); B[4 * cl + 1] = (float)(-2.0 * cv);
                        B[4 * cl + 2] = (kk == INFINITY) ? INFINITY : (float)kk + std::fabs((float)kk) * 1e-6f;
                        B[4 * cl + 3] = 0.f;
                        if (!std::isfinite(B[4 * cl]) || !std::isfinite(B[4 * cl + 1])) { B[4 * cl] = B[4 * cl + 1] = 0.f; B[4 * cl + 2] = INFINITY; }
                    }
                }
            }
            HIP_TRY(c->cl_sphere.upload(cl_sphere)); HIP_TRY(c->cl_normal.upload(cl_normal)); HIP_TRY(c->cl_camera.upload(cl_cam));
            for (size_t l = 0; l < NL; l++) graze_start[l * (NC + 1) + NC] = (int32_t)((l + 1 < NL) ? graze_start[(l + 1) * (NC + 1)] : (int32_t)graze_slots.size());
            c->n_graze = graze_slots.size();
            graze_slots.push_back(0);                        // (never empty)
            HIP_TRY(c->cl_lights.upload(cl_lights)); HIP_TRY(c->graze_slots.upload(graze_slots)); HIP_TRY(c->graze_start.upload(graze_start));
        }
        HIP_TRY(c->eye_a.upload(eye_a));
        HIP_TRY(c->cone_camera.upload(cone_cam));
        HIP_TRY(c->sphere_any.upload(sphere_any)); HIP_TRY(c->plane_any.upload(plane_any));
        HIP_TRY(c->light_dir_row.upload(row));
        if (n_rows > 0) HIP_TRY(c->dir_rinv.upload(rinv)); else c->dir_rinv.release();
    }

    HIP_TRY(c->tri_filt.upload(filt)); HIP_TRY(c->tri_exact.upload(exact));
    HIP_TRY(c->groups.upload(groups)); HIP_TRY(c->obj_slot.upload(slot));
    DevScene& s = c->hs;
    s.tri_filt = c->tri_filt.p; s.tri_exact = c->tri_exact.p; s.groups = c->groups.p; s.obj_slot = c->obj_slot.p;
    s.n_regular = NR; s.n_irregular = NI; s.n_groups = (int)groups.size();
    s.n_prims_nominal = c->n_regular_real + NI + s.n_spheres;
    s.robust_filter = robust ? 1 : 0;
    std::vector<DevScene> one(1, s);
    HIP_TRY(c->d_scene.upload(one));
    c->robust = robust ? 1 : 0;
    return 0;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

// No exception may cross the C boundary (host allocations, host threads of the precomputation):
// the entry points that can meet one turn it into an error code.
#define RT_GUARD_BEGIN try {
#define RT_GUARD_END(name)                                                                              \
    } catch (const std::bad_alloc&) { return fail(RT_ERR_DEVICE, name ": out of host memory"); }       \
      catch (const std::exception& e) { return fail(RT_ERR_DEVICE, std::string(name ": ") + e.what()); } \
      catch (...) { return fail(RT_ERR_DEVICE, name ": unknown failure"); }

static int upload_scene(rt_ctx** out, const rt_scene_desc* d);

int rt_upload_scene(rt_ctx** out, const rt_scene_desc* d) {
    RT_GUARD_BEGIN
    return upload_scene(out, d);
    RT_GUARD_END("rt_upload_scene")
}

}  // extern "C"

static int upload_scene(rt_ctx** out, const rt_scene_desc* d) {
    if (!out || !d) return fail(RT_ERR_ARG, "rt_upload_scene: null argument");
    if (d->struct_size != sizeof(rt_scene_desc)) return fail(RT_ERR_ARG, "rt_upload_scene: struct_size mismatch");
    if (d->width <= 0 || d->height <= 0 || d->max_depth < 0 || d->max_depth > 64)
        return fail(RT_ERR_ARG, "rt_upload_scene: bad width/height/max_depth");
    if ((long long)d->width * d->height > (1ll << 30)) return fail(RT_ERR_ARG, "rt_upload_scene: frame too large");
    if (d->n_objects < 0 || d->n_lights < 0 || d->n_materials < 0 || d->n_transforms < 0)
        return fail(RT_ERR_ARG, "rt_upload_scene: negative count");
    const int N = d->n_objects;
    if (N > 0 && (!d->obj_kind || !d->obj_material || !d->obj_transform || !d->obj_interp || !d->obj_geom ||
                  !d->materials || !d->transforms))
        return fail(RT_ERR_ARG, "rt_upload_scene: null object array");
    if (d->n_lights > 0 && !d->lights) return fail(RT_ERR_ARG, "rt_upload_scene: null lights");
    for (int i = 0; i < N; i++) {
        if (d->obj_kind[i] != 0 && d->obj_kind[i] != 1) return fail(RT_ERR_ARG, "rt_upload_scene: bad obj_kind");
        if (d->obj_material[i] < 0 || d->obj_material[i] >= d->n_materials)
            return fail(RT_ERR_ARG, "rt_upload_scene: material index out of range");
        if (d->obj_transform[i] < 0 || d->obj_transform[i] >= d->n_transforms)
            return fail(RT_ERR_ARG, "rt_upload_scene: transform index out of range");
        if (d->obj_interp[i] && !d->obj_normals) return fail(RT_ERR_ARG, "rt_upload_scene: obj_normals missing");
    }

    // Every transform must be affine: the bounds and margins behind the FP32 filter are derived
    // without a perspective divide (what parseSceneFile can build -- translations, rotations,
    // scalings and their products, raytracer.cpp:806-824 -- always is).
    for (int t = 0; t < d->n_transforms; t++) {
        const double* rows[2] = {d->transforms[t].m + 12, d->transforms[t].inv + 12};
        for (const double* r : rows)
            if (!(r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0 && r[3] == 1.0))
                return fail(RT_ERR_ARG, "rt_upload_scene: transform " + std::to_string(t) +
                                            " is not affine (the last row of m and of inv must be 0 0 0 1)");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RT_ERR_DEVICE, "rt_upload_scene: no HIP device (the trace loop has no CPU fallback)");
    rt_ctx* c = new rt_ctx();
    auto bail = [&](int code) { rt_free(c); return code; };
    hipDeviceProp_t prop;
    if (hipGetDevice(&c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess)
        return bail(fail(RT_ERR_DEVICE, "rt_upload_scene: cannot query the current device"));
    c->n_cu = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;

    const int NX = d->n_transforms;
    // RT_UPLOAD_TIMING in the environment: print how long each phase of the upload took (stderr)
    const bool timing = std::getenv("RT_UPLOAD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rt_upload_scene: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // ---- transforms ----
    std::vector<double> xf(48 * (size_t)NX);
    std::vector<int32_t> xf_ident(NX);
    for (int t = 0; t < NX; t++) {
        std::memcpy(&xf[48 * (size_t)t], d->transforms[t].m, 16 * 8);
        std::memcpy(&xf[48 * (size_t)t + 16], d->transforms[t].inv, 16 * 8);
        std::memcpy(&xf[48 * (size_t)t + 32], d->transforms[t].inv_t, 16 * 8);
        int ident = 1;
        for (int k = 0; k < 48; k++) {
            double want = ((k % 16) % 5 == 0) ? 1.0 : 0.0;
            if (std::memcmp(&xf[48 * (size_t)t + k], &want, 8) != 0) ident = 0;
        }
        xf_ident[t] = ident;
    }

    // ---- world bounds: every point a ray can start from ----
    Box world;
    world.add({d->cam_eye[0], d->cam_eye[1], d->cam_eye[2]});
    std::vector<double> xf_extent(std::max(NX, 1), 0.0);   // largest |coordinate| of a transform's object-space vertices
    {
        const size_t parts = host_parts((size_t)N);
        std::vector<Box> part_box(parts);
        std::vector<std::vector<double>> part_extent(parts, std::vector<double>(std::max(NX, 1), 0.0));
        parallel_ranges((size_t)N, parts, [&](size_t part, size_t i0, size_t i1) {
            Box& box = part_box[part];
            std::vector<double>& extent = part_extent[part];
            for (size_t i = i0; i < i1; i++) {
                const double* g = d->obj_geom + 9 * i;
                const double* M = &xf[48 * (size_t)d->obj_transform[i]];
                if (d->obj_kind[i] == 0) {
                    for (int k = 0; k < 3; k++) {
                        Vec v{g[3 * k], g[3 * k + 1], g[3 * k + 2]};
                        box.add(affine_point(M, v));
                        double& ext = extent[d->obj_transform[i]];
                        ext = std::max(ext, max_abs(v));
                    }
                } else {
                    double r = std::fabs(g[3]);
                    for (int k = 0; k < 8; k++) {
                        Vec v{g[0] + ((k & 1) ? r : -r), g[1] + ((k & 2) ? r : -r), g[2] + ((k & 4) ? r : -r)};
                        box.add(affine_point(M, v));
                    }
                }
            }
        });
        for (size_t part = 0; part < parts; part++) {
            for (int a = 0; a < 3; a++) {
                world.lo[a] = std::min(world.lo[a], part_box[part].lo[a]);
                world.hi[a] = std::max(world.hi[a], part_box[part].hi[a]);
            }
            for (int t = 0; t < NX; t++) xf_extent[t] = std::max(xf_extent[t], part_extent[part][t]);
        }
    }

    lap("bounds");
    // ---- per transform and kind of query: the magnitudes the filter's operands can have ----
    // Nearest-hit rays reach a group in its object space (raytracer.cpp:251-259): origins are the
    // world box under M^-1, camera directions are unnormalised, later directions have unit
    // length in world space.  Shadow rays stay in world space and meet the untransformed
    // vertices (raytracer.cpp:553); their directions are normalised (raytracer.cpp:108-119).
    auto clamp_big = [](double x, double cap) { return x < cap ? x : cap; };   // NaN -> cap
    auto widen = [](double x) { return x * (1.0 + 1e-3) + 1e-3; };
    Vec cam_dir[4];
    double cam_len = 1.0;
    {
        Vec eye{d->cam_eye[0], d->cam_eye[1], d->cam_eye[2]};
        const double* corners[4] = {d->cam_ll, d->cam_lr, d->cam_ul, d->cam_ur};
        for (int k = 0; k < 4; k++) {
            cam_dir[k] = vsub({corners[k][0], corners[k][1], corners[k][2]}, eye);
            cam_len = std::max(cam_len, vlen(cam_dir[k]));
        }
    }
    double world_ob = 0.0;
    for (int k = 0; k < 8; k++) world_ob = std::max(world_ob, max_abs(world.corner(k)));
    world_ob = widen(clamp_big(world_ob, 1e30));
    struct Mag { double ob, vm, R; };
    std::vector<Mag> mag0(std::max(NX, 1)), mag1(std::max(NX, 1));
    double R_max = std::sqrt(3.0) * world_ob;
    for (int t = 0; t < NX; t++) {
        double ob = world_ob, vm = cam_len;
        if (!xf_ident[t]) {
            const double* I = &xf[48 * (size_t)t + 16];
            ob = 0.0;
            for (int k = 0; k < 8; k++) ob = std::max(ob, max_abs(affine_point(I, world.corner(k))));
            ob = widen(clamp_big(ob, 1e30));
            double fro = 0;
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) fro += I[4 * r + cc] * I[4 * r + cc];
            vm = std::max(1.0, std::sqrt(fro));
            for (int k = 0; k < 4; k++) vm = std::max(vm, vlen(linear_dir(I, cam_dir[k])));
        }
        vm = clamp_big(vm, 1e15) * (1.0 + 1e-3);
        const double ext = widen(clamp_big(xf_extent[t], 1e30));
        mag0[t] = {ob, vm, std::sqrt(3.0) * std::max(ob, ext)};
        mag1[t] = {world_ob, 1.0 + 1e-3, std::sqrt(3.0) * std::max(world_ob, ext)};
        R_max = std::max(R_max, std::max(mag0[t].R, mag1[t].R));
    }

    // ---- filter thresholds (every constant is rounded up) ----
    const double EPSF = 1.1920928955078125e-7;   // 2^-23
    // amp_out: position-error amplification of the most grazing ray the reference still accepts
    auto thresholds = [&](float eps, const Mag& m, double* amp_out) {
        GroupFilter f{};
        const double Eb = 8 * EPSF * m.vm, Ea = 16 * EPSF * m.R;
        double b_lo = (double)eps * (1 - 1e-3) - Eb;
        double r_lo = (double)eps * (1 - 1e-3) - 4 * (Ea + (double)eps * Eb) / std::max(b_lo, 1e-30);
        if (!(b_lo > 0)) { b_lo = 0; r_lo = -FLT_MAX; }      // nothing is certain about grazing rays
        f.r_lo = (float)std::max(r_lo, -(double)FLT_MAX);
        f.a_zero = (float)(2 * Ea * (1 + 1e-6));
        f.b_hi = (float)(((double)std::fabs(eps) + Eb) * (1 + 1e-3));
        f.e_a = (float)(Ea * (1 + 1e-6)); f.e_b = (float)(Eb * (1 + 1e-6));
        f.r_hi = (float)((double)std::fabs(eps) * (1 + 1e-3));
        f.origin_bound = (float)(m.ob * (1 - 1e-6));
        f.dir_len2_bound = (float)(m.vm * m.vm * (1 - 1e-6));
        *amp_out = 18.0 * m.vm / std::max(b_lo, 1e-30) + 20.0;
        return f;
    };
    c->h_q0.resize(std::max(NX, 1)); c->h_q1.resize(std::max(NX, 1));
    std::vector<double> amp0(std::max(NX, 1)), amp1(std::max(NX, 1));
    for (int t = 0; t < NX; t++) {
        c->h_q0[t] = thresholds(d->eps, mag0[t], &amp0[t]);
        c->h_q1[t] = thresholds(d->shadow_eps, mag1[t], &amp1[t]);
    }

    // ---- triangles: reference-precision records and both forms of the filter record ----
    std::vector<HostTri>& tris = c->h_tris;
    std::vector<SphereRec> sph;
    std::vector<SphereFilt> sph_filt;
    c->h_slot.assign(N, -1);
    std::vector<int> tri_objects;            // Scene::objects indices of the triangles, in order
    for (int i = 0; i < N; i++) {
        const double* g = d->obj_geom + 9 * (size_t)i;
        if (d->obj_kind[i] == 1) {
            SphereRec s{};
            s.c[0] = g[0]; s.c[1] = g[1]; s.c[2] = g[2];
            float radius = (float)g[3];
            s.r2 = (double)radius * (double)radius;
            s.obj = i; s.xf = d->obj_transform[i]; s.identity = xf_ident[s.xf];
            c->h_slot[i] = (int32_t)sph.size();
            sph.push_back(s);
            SphereFilt f{};
            const double* I = &xf[48 * (size_t)s.xf + 16];
            for (int r = 0; r < 3; r++) {
                double row = 0;
                for (int q = 0; q < 4; q++) { f.m[4 * r + q] = (float)I[4 * r + q]; if (q < 3) row += std::fabs(I[4 * r + q]); }
                f.norm = std::max(f.norm, (float)(row * (1 + 1e-6)));
                f.tabs += (float)(std::fabs(I[4 * r + 3]) * (1 + 1e-6));
            }
            for (int q = 0; q < 3; q++) { f.c[q] = (float)g[q]; f.tabs += (float)(std::fabs(g[q]) * (1 + 1e-6)); }
            f.r2 = (float)s.r2;
            f.identity = s.identity;
            // (a transform or sphere whose floats are not finite: the pre-test keeps every pair, NaNs compare false)
            sph_filt.push_back(f);
        } else {
            tri_objects.push_back(i);
        }
    }
    tris.assign(tri_objects.size(), HostTri{});
    // one record per triangle, independent of the others: large meshes are spread over host threads
    auto build_records = [&](size_t k_begin, size_t k_end) {
      for (size_t k_tri = k_begin; k_tri < k_end; k_tri++) {
        const int i = tri_objects[k_tri];
        const double* g = d->obj_geom + 9 * (size_t)i;
        HostTri t{};
        t.xf = d->obj_transform[i];
        Vec v1{g[0], g[1], g[2]}, v2{g[3], g[4], g[5]}, v3{g[6], g[7], g[8]};
        Vec e1 = vsub(v3, v1), e2 = vsub(v2, v1);                  // raytracer.cpp:288-289
        Vec cr = vcross(e2, e1);
        double len = std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
        double inv = 1. / len;                                      // algebra3.h:638-662
        Vec n{cr.x * inv, cr.y * inv, cr.z * inv};
        float uu = (float)vdot(e1, e1), uv = (float)vdot(e1, e2), vv = (float)vdot(e2, e2);
        float D = uv * uv - uu * vv;                                // raytracer.cpp:340 (float arithmetic)
        TriExact& ex = t.ex;
        ex.v1[0] = v1.x; ex.v1[1] = v1.y; ex.v1[2] = v1.z;
        ex.e1[0] = e1.x; ex.e1[1] = e1.y; ex.e1[2] = e1.z;
        ex.e2[0] = e2.x; ex.e2[1] = e2.y; ex.e2[2] = e2.z;
        ex.n[0] = n.x; ex.n[1] = n.y; ex.n[2] = n.z;
        ex.uu = uu; ex.uv = uv; ex.vv = vv; ex.D = D; ex.obj = i;
        bool ok = std::isfinite(n.x) && std::isfinite(n.y) && std::isfinite(n.z) && std::isfinite(uu) &&
                  std::isfinite(uv) && std::isfinite(vv) && std::isfinite(D) && D != 0.0f;
        if (ok) {
            // s = S.(x - v1), t = T.(x - v1) are the reference's barycentrics (raytracer.cpp:344, 348)
            // as linear functions, built from ITS rounded uu, uv, vv, D.
            double fuu = uu, fuv = uv, fvv = vv, fD = D;
            Vec S{(fuv * e2.x - fvv * e1.x) / fD, (fuv * e2.y - fvv * e1.y) / fD, (fuv * e2.z - fvv * e1.z) / fD};
            Vec T{(fuv * e1.x - fuu * e2.x) / fD, (fuv * e1.y - fuu * e2.y) / fD, (fuv * e1.z - fuu * e2.z) / fD};
            double le1 = std::sqrt(fuu), le2 = std::sqrt(fvv);
            double wmax = 1.5 * (le1 + le2);
            // rounding of the reference's own float formula
            double dBs = 2 * EPSF * wmax * (std::fabs(fuv) * le2 + fvv * le1) / std::fabs(fD) + 3 * EPSF;
            double dBt = 2 * EPSF * wmax * (std::fabs(fuv) * le1 + fuu * le2) / std::fabs(fD) + 3 * EPSF;
            ok = dBs < 0.125 && dBt < 0.125 && std::isfinite(dBs) && std::isfinite(dBt);
            // a point the reference accepts lies within (dBs, dBt) barycentric units of the triangle
            ex.box_pad = (float)(4.0 * (dBs + dBt) * (le1 + le2));
            if (ok) {
                // Each coordinate as a function of the two hit-point coordinates (u, v) other than
                // the normal's dominant axis k: on the plane x_k = (d0 - n_u x_u - n_v x_v) / n_k.
                double d0 = vdot(n, v1);
                double nn[3] = {n.x, n.y, n.z};
                int k = 0;
                if (std::fabs(nn[1]) > std::fabs(nn[k])) k = 1;
                if (std::fabs(nn[2]) > std::fabs(nn[k])) k = 2;
                const int iu = (k + 1) % 3, iv = (k + 2) % 3;
                t.axis = k;
                Vec U{-(S.x + T.x), -(S.y + T.y), -(S.z + T.z)};
                const double off3[3] = {-vdot(S, v1), -vdot(T, v1), 1.0 - vdot(U, v1)};
                const Vec* planes[3] = {&S, &T, &U};
                double coef[3][3];
                double norm2d[3];
                for (int q = 0; q < 3; q++) {
                    double pc[3] = {planes[q]->x, planes[q]->y, planes[q]->z};
                    coef[q][0] = pc[iu] - pc[k] * nn[iu] / nn[k];
                    coef[q][1] = pc[iv] - pc[k] * nn[iv] / nn[k];
                    coef[q][2] = off3[q] + pc[k] * d0 / nn[k];
                    norm2d[q] = std::sqrt(coef[q][0] * coef[q][0] + coef[q][1] * coef[q][1]);
                }
                const Mag& m0 = mag0[t.xf];
                const Mag& m1 = mag1[t.xf];
                for (int form = 0; form < 2; form++) {
                    float* f = form ? t.f_robust : t.f_static;
                    f[0] = (float)n.x; f[1] = (float)n.y; f[2] = (float)n.z; f[3] = (float)d0;
                    double ds, dt;
                    if (form == 0) {
                        // rounding of the FP32 filter itself (hit position and plane evaluation), worst
                        // case over both kinds of query and every ray
                        const double RA = std::max(m0.R * amp0[t.xf], m1.R * amp1[t.xf]);
                        ds = 4 * (EPSF * norm2d[0] * RA + dBs) + 1e-6;
                        dt = 4 * (EPSF * norm2d[1] * RA + dBt) + 1e-6;
                        f[13] = f[14] = 0.f;
                    } else {
                        // the same bound with the ray's own 1/|n.dir| in place of the worst case:
                        // 18 vm/|b| + 20 <= 40 vm/|b| because |b| <= vm; the factor is applied per test
                        ds = 4 * dBs + 1e-6;
                        dt = 4 * dBt + 1e-6;
                        const double nsum = norm2d[0] + norm2d[1];
                        f[13] = (float)(4 * EPSF * nsum * m0.R * 42.0 * m0.vm);
                        f[14] = (float)(4 * EPSF * nsum * m1.R * 42.0 * m1.vm);
                    }
                    const double du = ds + dt;
                    bool fok = std::isfinite(ds) && std::isfinite(dt) && du < 1e6;
                    f[4] = (float)coef[0][0]; f[5] = (float)coef[0][1]; f[6] = (float)(coef[0][2] + ds);
                    f[7] = (float)coef[1][0]; f[8] = (float)coef[1][1]; f[9] = (float)(coef[1][2] + dt);
                    f[10] = (float)coef[2][0]; f[11] = (float)coef[2][1]; f[12] = (float)(coef[2][2] + du);
                    f[15] = (float)(du * (1 + 1e-6));      // the largest margin folded into the offsets (LBVH pruning)
                    for (int q = 0; q < 16; q++) fok = fok && std::isfinite(f[q]);
                    (form ? t.ok_robust : t.ok_static) = fok;
                    if (form == 0) t.margin_static = (float)std::min(du, 1e30);
                }
            }
        }
        tris[k_tri] = t;
      }
    };
    parallel_ranges(tri_objects.size(), host_parts(tri_objects.size()),
                    [&](size_t, size_t k0, size_t k1) { build_records(k0, k1); });
    c->h_xf_ident = xf_ident;
    c->h_xf = xf;
    c->scene_bound = R_max;

    lap("triangle records");
    // ---- which form of the filter: the cheap static margins unless they are useless here ----
    {
        double sum = 0; size_t n_ok = 0, n_only_robust = 0;
        // plane -> number of triangles in it: open addressing over twice as many slots as triangles
        size_t n_slots = 16;
        while (n_slots < 2 * tris.size()) n_slots *= 2;
        std::vector<unsigned long long> slot_key(n_slots, 0);
        std::vector<int> slot_count(n_slots, 0);
        int largest_plane = 0;
        for (const HostTri& t : tris) {
            if (t.ok_robust && !t.ok_static) n_only_robust++;
            if (!t.ok_static) continue;
            sum += std::min((double)t.margin_static, 100.0); n_ok++;
            // triangles lying in one plane: a ray leaving that plane lists them all in the static form
            const float* f = t.f_static;
            unsigned long long key = (unsigned long long)(unsigned int)t.xf * 0x9E3779B97F4A7C15ull;
            for (int q = 0; q < 4; q++) {
                const double scale = q < 3 ? 2048.0 : 2048.0 / std::max(mag0[t.xf].R, 1e-30);
                key = (key ^ (unsigned long long)(long long)std::llround(f[q] * scale)) * 0x100000001B3ull;
            }
            key |= 1ull;                                    // 0 marks an empty slot
            size_t at = (size_t)(key >> 17) & (n_slots - 1);
            while (slot_key[at] != 0 && slot_key[at] != key) at = (at + 1) & (n_slots - 1);
            slot_key[at] = key;
            largest_plane = std::max(largest_plane, ++slot_count[at]);
        }
        const double mean_margin = n_ok ? sum / (double)n_ok : 0.0;
        c->auto_robust = (n_only_robust > 0 || mean_margin > 1.0 || largest_plane > 16) ? 1 : 0;
    }

    std::vector<DevMaterial> mats(d->n_materials);
    for (int i = 0; i < d->n_materials; i++) std::memcpy(&mats[i], &d->materials[i], sizeof(DevMaterial));
    std::vector<DevLight> lights(d->n_lights);
    for (int i = 0; i < d->n_lights; i++) std::memcpy(&lights[i], &d->lights[i], sizeof(DevLight));
    c->h_lights = lights;
    static_assert(sizeof(DevMaterial) == sizeof(rt_material), "material layout");
    static_assert(sizeof(DevLight) == sizeof(rt_light), "light layout");
    static_assert(sizeof(TriExact) == 128, "TriExact layout");

#define UP(buf, ...)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = c->buf.upload(__VA_ARGS__);                                                    \
        if (e_ != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("upload " #buf ": ") + hipGetErrorString(e_))); \
    } while (0)
    // the per-object arrays go to the device straight from the caller's buffers
    const size_t n_obj = (size_t)N;
    UP(spheres, sph); UP(sphere_filt, sph_filt); UP(xforms, xf); UP(xf_identity, xf_ident);
    UP(obj_kind, d->obj_kind, n_obj); UP(obj_material, d->obj_material, n_obj);
    UP(obj_transform, d->obj_transform, n_obj); UP(obj_interp, d->obj_interp, n_obj);
    UP(obj_geom, d->obj_geom, 9 * n_obj); UP(obj_normals, d->obj_normals, 9 * n_obj);
    UP(materials, mats); UP(lights, lights);
#undef UP

    DevScene& s = c->hs;
    s.spheres = c->spheres.p; s.sphere_filt = c->sphere_filt.p;
    s.xforms = c->xforms.p; s.xf_identity = c->xf_identity.p; s.obj_kind = c->obj_kind.p;
    s.obj_material = c->obj_material.p; s.obj_transform = c->obj_transform.p; s.obj_interp = c->obj_interp.p;
    s.obj_geom = c->obj_geom.p; s.obj_normals = c->obj_normals.p; s.materials = c->materials.p; s.lights = c->lights.p;
    s.n_spheres = (int)sph.size();
    s.n_objects = N; s.n_lights = d->n_lights; s.n_transforms = NX;
    s.last_xf = N > 0 ? d->obj_transform[N - 1] : 0;
    s.width = d->width; s.height = d->height; s.max_depth = d->max_depth;
    s.eps = d->eps; s.shadow_eps = d->shadow_eps;
    std::memcpy(s.cam_eye, d->cam_eye, 24); std::memcpy(s.cam_ll, d->cam_ll, 24); std::memcpy(s.cam_lr, d->cam_lr, 24);
    std::memcpy(s.cam_ul, d->cam_ul, 24); std::memcpy(s.cam_ur, d->cam_ur, 24); std::memcpy(s.ambient, d->ambient, 24);
    // the shading ray of an empty transform table cannot occur: N > 0 implies NX > 0

    lap("form choice + small uploads");
    // triangle records, groups and the scene header, in the form chosen above
    if (int rc = install_filter(c, c->auto_robust)) return bail(rc);
    lap("install_filter");
    hipError_t e = c->own_frame.alloc((size_t)d->width * d->height);
    if (e != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("frame: ") + hipGetErrorString(e)));
    c->frame = c->own_frame.p;
    e = hipMemset(c->frame, 0, (size_t)d->width * d->height * 4);
    if (e == hipSuccess) e = c->counters.alloc(1);
    if (e == hipSuccess) e = hipMemset(c->counters.p, 0, sizeof(DevCounters));
    if (e == hipSuccess) e = c->work.alloc(RT_WORK_KINDS * RT_WORK_FIELDS);
    if (e == hipSuccess) e = hipMemset(c->work.p, 0, RT_WORK_KINDS * RT_WORK_FIELDS * sizeof(unsigned long long));
    if (e != hipSuccess) return bail(fail(RT_ERR_DEVICE, std::string("init: ") + hipGetErrorString(e)));

    // launch geometry: as many resident blocks as the filter kernels admit
#define RT_GRID_OF(M, R, CN, RB, WK)                                                                               \
    do {                                                                                                            \
        int nb = 0;                                                                                                 \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_filter<M, R, RB, CN, WK>, RT_BLOCK, 0) != hipSuccess || nb < 1) nb = 2; \
        c->filter_grid[M][CN][RB ? 1 : 0][WK ? 1 : 0] = c->n_cu * std::min(nb, 8);                                  \
    } while (0)
#define RT_GRID_OF_ALL(M, R, CN) do { RT_GRID_OF(M, R, CN, false, false); RT_GRID_OF(M, R, CN, true, false); RT_GRID_OF(M, R, CN, false, true); RT_GRID_OF(M, R, CN, true, true); } while (0)
    RT_GRID_OF_ALL(0, RT_RPL, 0);
    RT_GRID_OF_ALL(1, RT_RPL, 0);
    RT_GRID_OF_ALL(0, RT_RPL_SHARED, 1);
    RT_GRID_OF_ALL(0, RT_RPL_ANY, 4);
    RT_GRID_OF_ALL(1, RT_RPL_SHARED, 2);
#undef RT_GRID_OF_ALL
#undef RT_GRID_OF
    c->aux_grid = c->n_cu * 8;
    if (const char* env = std::getenv("RAYTRACER_AMD_GRID_HINTS")) c->grid_hints = std::atoi(env) ? 1 : 0;
    if (const char* env = std::getenv("RAYTRACER_AMD_BESIDE_DIV")) c->beside_grid_div = std::max(1, std::min(16, std::atoi(env)));
    // this device's copy of the static workspace (for rt_render_tile_noalloc)
    if (int rc = resolve_scratch(c->device)) return bail(rc);
    *out = c;
    return 0;
}

extern "C" {

// RT_OPT_BVH_PLANE_GUARD = 2: one whole frame through the LBVH with the in-plane-origin scan behind every traversal,
// into the context's own frame buffer, its counts kept out of the statistics.  The scan stays on iff some pair of that
// frame took the in-plane-origin branch of Triangle::intersection (raytracer.cpp:319-321).
static int calibrate_bvh_guard(rt_ctx* c) {
    if (c->bvh_guard_mode != 2) { c->bvh_guard = c->bvh_guard_mode; return 0; }
    if (c->bvh_guard_calibrated) { c->bvh_guard = c->bvh_guard_found ? 1 : 0; return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipDeviceSynchronize());
    DevCounters before;
    HIP_TRY(hipMemcpy(&before, c->counters.p, sizeof before, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> tally(2 * (size_t)c->aux_grid * 4);
    HIP_TRY(hipMemcpy(tally.data(), c->bvh_tally.p, tally.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint32_t* const frame = c->frame;
    const int in_flight = c->in_flight, profile = c->profile, work = c->work_counters, use = c->use_bvh, exact = c->force_exact;
    c->frame = c->own_frame.p; c->in_flight = 1; c->profile = 0; c->work_counters = 0; c->force_exact = 0;
    c->use_bvh = 1; c->bvh_guard = 1;
    int rc = render_rows(c, 0, c->hs.height, c->hs.height, 1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    c->frame = frame; c->in_flight = in_flight; c->profile = profile; c->work_counters = work; c->use_bvh = use; c->force_exact = exact;
    if (rc) return rc;
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, std::string("LBVH calibration frame: ") + hipGetErrorString(e));
    DevCounters after;
    HIP_TRY(hipMemcpy(&after, c->counters.p, sizeof after, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c->counters.p, &before, sizeof before, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->bvh_tally.p, tally.data(), tally.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(c->own_frame.p, 0, (size_t)c->hs.width * c->hs.height * 4));
    c->bvh_guard_found = after.in_plane_origin_hits - before.in_plane_origin_hits;
    c->bvh_guard_calibrated = true;
    c->bvh_guard = c->bvh_guard_found ? 1 : 0;
    c->bvh_guard_calibration_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("RT_GUARD_REPORT"))
        fprintf(stderr, "LBVH in-plane-origin scan: calibration frame found %llu such pairs in %.3f s: scan %s\n",
                c->bvh_guard_found, c->bvh_guard_calibration_s, c->bvh_guard ? "on" : "off");
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
            const int want = value == 0 ? c->auto_robust : (value == 2 ? 1 : 0);
            if (want == c->robust) return 0;
            const int had_bvh = c->use_bvh;
            int rc = install_filter(c, want);
            if (rc) return rc;
            if (had_bvh) return build_bvh(c);
            return 0;
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
        case RT_OPT_BVH_PLANE_GUARD:
            if (value < 0 || value > 2) return fail(RT_ERR_ARG, "rt_set_option: the LBVH's in-plane-origin scan is 0 (off), 1 (on) or 2 (decided by a calibration frame)");
            c->bvh_guard_mode = (int)value;
            c->bvh_guard = value == 2 ? 1 : (int)value;
            if (c->use_bvh) return calibrate_bvh_guard(c);
            return 0;
        case RT_OPT_BOUNDING_TESTS: c->pretest = value ? 1 : 0; c->pretest_any = value == 1 ? 1 : 0; return 0;
        case RT_OPT_USE_BVH:
            if (value) {
                HIP_TRY(hipSetDevice(c->device));
                int rc = build_bvh(c);
                if (rc) return rc;
                rc = calibrate_bvh_guard(c);
                if (rc) return rc;
            }
            c->use_bvh = value == 2 ? 2 : (value ? 1 : 0);      // 1: four-wide nodes, 2: the binary tree they are made from
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
    // put it: the one slice that must start at zero (the count of finished blocks behind the
    // hit offsets) is cleared here; queue lengths, counters and flags are cleared by the pass.
    // (calls still in flight in lanes come first: pixels land in call order; a wait allocates nothing)
    for (Lane* L : c->lanes)
        if (L->pending) HIP_TRY(hipStreamWaitEvent(st, L->done, 0));
    HIP_TRY(hipMemsetAsync(ws.hit_offsets, 0, ((size_t)c->aux_grid + 1) * sizeof(uint32_t), st));
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
        HIP_TRY(hipMemsetAsync(ws.counts, 0, (ws.levels + 2) * sizeof(uint32_t), st));
        const bool scan = keeps_scan_copies(c, ws);
        RayQueue q = queue_of(ws, 0, scan);
        hipLaunchKernelGGL(k_primary, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q, ws.keys[0],
                           ws.pixel_of_ray, c->frame, ws.counts);
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
        HIP_TRY(hipMemsetAsync(ws.counts, 0, (ws.levels + 2) * sizeof(uint32_t), st));
        const bool scan = keeps_scan_copies(c, ws), scan_shadow = scan && keeps_shadow_copies(c, ws);
        RayQueue q[2] = {queue_of(ws, 0, scan), queue_of(ws, 1, scan)};
        hipLaunchKernelGGL(k_primary, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, dsc, pd, n_pix, q[0], ws.keys[0],
                           ws.pixel_of_ray, c->frame, ws.counts);
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
        ha.n_rays = ws.counts; ha.hit_offsets = ws.hit_offsets; ha.lvl = lvl; ha.aux = aux;
        ha.occluded = ws.occluded[0]; ha.occl_words = words;
        ha.shadow_scan = scan_shadow ? ws.shadow_scan[0] : nullptr; ha.shadow_stride = (uint32_t)ws.cap;
        ha.light_frame = c->light_frame.p;
        hipLaunchKernelGGL(k_hit_offsets, dim3(c->aux_grid), dim3(RT_BLOCK), 0, st, (const unsigned long long*)ws.keys[0],
                           (const uint32_t*)ws.counts, ws.hit_offsets, ws.hit_offsets + c->aux_grid, ws.counts + 1);
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
    out->camera_ms = c->ms[1]; out->camera_launches = c->launches[1]; out->bvh_plane_guard = (c->use_bvh && c->bvh_guard) ? 1u : 0u;
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
