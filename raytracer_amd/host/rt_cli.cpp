// Host driver: stands where the reference's main() does (raytracer.cpp:843-868).
//
//   rt_render <scene> [out.png] [-w W] [-h H] [-depth D] [-eps E] [-tile ROWS] [-gpus N] [-frames F] [-inflight K]
//             [-packed] [-bvh] [-exact] [-stats]
//
// With no options it does what `./raytracer <scene> [out.png]` does: 1000 x 1000, eps 0.05,
// depth 6, output.png (raytracer.cpp:847-862).  Parsing goes through the host loader
// (include/rt_host.h), the trace loop through the C ABI (include/rt_abi.h), the PNG through
// rth_write_png.  There is no CPU rendering path here: without a GPU the program fails.
//
// -gpus N: the frame is cut into interleaved scanline tiles of ROWS rows (default 16); GPU r of N
// traces tiles r, r + N, r + 2N, ... (rt_render_tiles) into a full-size frame of its own, and one
// exchange at the end of the frame moves every tile to GPU 0: grouped ncclSend / ncclRecv (RCCL
// over xGMI) of each tile's rows -- which are contiguous in the frame (rth_tile_span) -- straight
// into the same rows of GPU 0's frame, no staging copy on either side.  One host thread per GPU,
// one communicator per GPU from ncclCommInitAll.  The reference has no counterpart: it is one
// thread on one CPU.
//
// -packed (with -gpus N): one message per peer instead of one per tile.  The tiles a GPU owns lie at a constant
// stride in its frame (tile k, k + N, ...: rth_tile_span), so one strided copy (hipMemcpy2DAsync) packs them into a
// contiguous buffer, one ncclSend moves it, GPU 0 receives every peer's buffer with one ncclRecv each and one strided
// copy per peer puts the tiles in place.  Same bytes in GPU 0's frame as the per-tile exchange.
//
// A GPU that fails (upload, allocation, a render call) must not leave the others waiting in an exchange it will
// never join: the ranks agree after set-up (a host barrier and a shared flag) and none enters the frame loop unless
// all are ready; a failure inside the loop aborts every communicator (ncclCommAbort), which ends transfers already
// enqueued.  The process then reports the failing GPU's message and exits with status 1.
//
// -frames F -inflight K (with -gpus; -gpus 1 if that is not given): the frame is rendered F times with up to K
// render calls in flight per GPU (RT_OPT_FRAMES_IN_FLIGHT), each into a frame buffer of its own; the exchange
// of frame i is enqueued behind rt_join(stream, K - 1), beside the tracing of the frames after it.  The last
// frame is the one written; -stats reports the rays of all F frames over the time of all of them.
#include "rt_abi.h"
#include "rt_host.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace {

struct RankResult {
    int rc = 0;
    std::string error;
    rt_stats stats{};
    double render_s = 0;          // enqueue of the tiles until this rank's part of the exchange has finished
};

// What the rank threads of one process share to agree on success before they meet in a collective.
struct RankSync {
    explicit RankSync(int n) : n_ranks(n) {}
    const int n_ranks;
    std::atomic<int> failed{0};
    std::vector<ncclComm_t>* comms = nullptr;
    std::atomic<int> aborted{0};
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    // every rank calls it once per meeting point; returns when all have
    void meet() {
        std::unique_lock<std::mutex> lock(m);
        const int gen = generation;
        if (++arrived == n_ranks) { arrived = 0; generation++; cv.notify_all(); return; }
        cv.wait(lock, [&] { return generation != gen; });
    }
    // a rank failed while others may already wait in an exchange: end every communicator's transfers (once)
    void abort_all() {
        failed.store(1);
        if (aborted.exchange(1) || !comms) return;
        for (ncclComm_t c : *comms) (void)ncclCommAbort(c);
    }
};

// Test hook: RT_RENDER_FAIL="<rank>:<where>" makes that rank fail on purpose, where = "setup" (before the ranks
// have agreed to start) or "frame" (inside the frame loop, after its first render call).
bool forced_failure(int rank, const char* where) {
    const char* env = std::getenv("RT_RENDER_FAIL");
    if (!env) return false;
    return std::string(env) == std::to_string(rank) + ":" + where;
}

// One GPU's share of an N-GPU frame.  Rank 0 ends up with the whole frame in `image`.
void run_rank(int rank, int n_ranks, ncclComm_t comm, const rt_scene_desc* desc, int tile_rows, bool bvh, bool exact,
              int n_frames, int in_flight, bool packed, RankSync* sync, std::vector<uint8_t>* image, RankResult* out) {
    auto fail = [&](const std::string& what) { out->rc = 1; out->error = "GPU " + std::to_string(rank) + ": " + what; };
    const int W = desc->width, H = desc->height;
    const size_t frame_bytes = (size_t)W * H * 4;
    rt_ctx* ctx = nullptr;
    std::vector<void*> frames((size_t)in_flight, nullptr);      // one buffer per render call in flight
    hipStream_t stream = nullptr;
    const int n_tiles = (H + tile_rows - 1) / tile_rows;
    // ---- the packed exchange: per rank, its full tiles (tile_rows rows each) at a constant stride, and at most one
    // clipped tile (the frame's last, when the height is not a multiple of tile_rows) ----
    struct Share { int full = 0; size_t first_off = 0; bool clipped = false; size_t clip_off = 0, clip_bytes = 0; size_t bytes = 0; };
    size_t tile_bytes = (size_t)tile_rows * W * 4;
    auto share_of = [&](int r) {
        Share s;
        size_t tb = 0, stride = 0;
        if (rth_packed_share(W, H, tile_rows, n_ranks, r, &s.full, &s.first_off, &tb, &stride, &s.clip_off, &s.clip_bytes)) s = Share{};
        s.clipped = s.clip_bytes > 0;
        s.bytes = (size_t)s.full * tile_bytes + s.clip_bytes;
        return s;
    };
    std::vector<Share> shares;
    std::vector<size_t> share_at;                 // GPU 0: where each peer's buffer starts in `staging`
    void* staging = nullptr;                      // GPU 0: all peers' buffers; other GPUs: their own
    // tiles <-> buffer: one strided copy for the full tiles (stride n_ranks tiles in the frame), one for the clipped tile
    auto strided = [&](void* frame, void* buf, const Share& s, bool to_buffer) -> hipError_t {
        hipError_t e = hipSuccess;
        char* f = (char*)frame + s.first_off;
        if (s.full > 0)
            e = to_buffer ? hipMemcpy2DAsync(buf, tile_bytes, f, (size_t)n_ranks * tile_bytes, tile_bytes, (size_t)s.full, hipMemcpyDeviceToDevice, stream)
                          : hipMemcpy2DAsync(f, (size_t)n_ranks * tile_bytes, buf, tile_bytes, tile_bytes, (size_t)s.full, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess && s.clipped) {
            char* b = (char*)buf + (size_t)s.full * tile_bytes;
            e = to_buffer ? hipMemcpyAsync(b, (char*)frame + s.clip_off, s.clip_bytes, hipMemcpyDeviceToDevice, stream)
                          : hipMemcpyAsync((char*)frame + s.clip_off, b, s.clip_bytes, hipMemcpyDeviceToDevice, stream);
        }
        return e;
    };
    // one frame's exchange, enqueued on `stream`
    auto exchange = [&](void* frame) -> ncclResult_t {
        if (n_ranks == 1) return ncclSuccess;
        if (packed) {
            if (rank != 0 && shares[(size_t)rank].bytes && strided(frame, staging, shares[(size_t)rank], true) != hipSuccess) return ncclUnhandledCudaError;
            ncclResult_t nr = ncclGroupStart();
            if (rank == 0) {
                for (int p = 1; p < n_ranks && nr == ncclSuccess; p++)
                    if (shares[(size_t)p].bytes) nr = ncclRecv((char*)staging + share_at[(size_t)p], shares[(size_t)p].bytes, ncclUint8, p, comm, stream);
            } else if (shares[(size_t)rank].bytes) {
                nr = ncclSend(staging, shares[(size_t)rank].bytes, ncclUint8, 0, comm, stream);
            }
            const ncclResult_t ne = ncclGroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (rank == 0)
                for (int p = 1; p < n_ranks && nr == ncclSuccess; p++)
                    if (shares[(size_t)p].bytes && strided(frame, (char*)staging + share_at[(size_t)p], shares[(size_t)p], false) != hipSuccess) nr = ncclUnhandledCudaError;
            return nr;
        }
        // every tile's rows from its owner straight into GPU 0's frame
        ncclResult_t nr = ncclGroupStart();
        for (int k = 0; k < n_tiles && nr == ncclSuccess; k++) {
            size_t off = 0, bytes = 0;
            const int owner = rth_tile_span(W, H, tile_rows, n_ranks, k, &off, &bytes);
            if (owner < 0) { nr = ncclInvalidArgument; break; }
            if (owner == 0) continue;                                        // already where it belongs
            if (rank == 0) nr = ncclRecv((char*)frame + off, bytes, ncclUint8, owner, comm, stream);
            else if (rank == owner) nr = ncclSend((const char*)frame + off, bytes, ncclUint8, 0, comm, stream);
        }
        const ncclResult_t ne = ncclGroupEnd();
        return nr == ncclSuccess ? ne : nr;
    };
    bool met = false;                             // this rank has been to the meeting point after set-up
    do {
        // ---- set-up: everything that can fail before the first collective ----
        if (hipSetDevice(rank) != hipSuccess) { fail("hipSetDevice failed"); break; }
        if (forced_failure(rank, "setup")) { fail("failure forced by RT_RENDER_FAIL (setup)"); break; }
        if (rt_upload_scene(&ctx, desc)) { fail(rt_last_error()); break; }
        if (exact && rt_set_option(ctx, RT_OPT_EXACT_ONLY, 1)) { fail(rt_last_error()); break; }
        if (bvh && rt_set_option(ctx, RT_OPT_USE_BVH, 1)) { fail(rt_last_error()); break; }
        if (in_flight > 1 && rt_set_option(ctx, RT_OPT_FRAMES_IN_FLIGHT, in_flight)) { fail(rt_last_error()); break; }
        bool ok = true;
        for (auto& f : frames)
            ok = ok && hipMalloc(&f, frame_bytes) == hipSuccess && hipMemset(f, 0, frame_bytes) == hipSuccess;
        if (!ok) { fail("cannot allocate the frame"); break; }
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { fail("cannot create a stream"); break; }
        if (packed && n_ranks > 1) {
            size_t total = 0;
            for (int r = 0; r < n_ranks; r++) {
                shares.push_back(share_of(r));
                share_at.push_back(total);
                if (r > 0) total += shares.back().bytes;
            }
            const size_t want = rank == 0 ? total : shares[(size_t)rank].bytes;
            if (want && hipMalloc(&staging, want) != hipSuccess) { fail("cannot allocate the exchange buffer"); break; }
        }
        // ---- agree: nobody starts unless everybody can ----
        if (out->rc) sync->failed.store(1);
        sync->meet();
        met = true;
        if (sync->failed.load()) break;                       // (the failing rank reports; the others just leave)
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<void*> waiting;                   // frames traced (or being traced) and not yet exchanged
        ncclResult_t nr = ncclSuccess;
        int rc = 0;
        for (int i = 0; i < n_frames && !rc && nr == ncclSuccess && !sync->failed.load(); i++) {
            void* frame = frames[(size_t)(i % in_flight)];
            rc = rt_bind_frame(ctx, frame, frame_bytes);
            if (!rc) rc = rt_render_tiles(ctx, rank, tile_rows, n_ranks, stream);
            if (!rc && i == 0 && forced_failure(rank, "frame")) { fail("failure forced by RT_RENDER_FAIL (frame)"); rc = -1; }
            if (rc) break;
            waiting.push_back(frame);
            if ((int)waiting.size() == in_flight) {
                // the oldest call's pixels, and only those, are waited for (a no-op with one call in flight,
                // whose work is on `stream` already); its exchange runs beside the calls after it
                rc = rt_join(ctx, stream, in_flight - 1);
                if (!rc) nr = exchange(waiting.front());
                waiting.erase(waiting.begin());
            }
        }
        if (!rc && !sync->failed.load()) rc = rt_join(ctx, stream, 0);
        for (size_t k = 0; k < waiting.size() && !rc && nr == ncclSuccess && !sync->failed.load(); k++) nr = exchange(waiting[k]);
        if (rc || nr != ncclSuccess) {
            if (!out->rc) fail(rc ? std::string(rt_last_error()) : std::string("RCCL exchange: ") + ncclGetErrorString(nr));
            sync->abort_all();                                // peers may already wait for this rank's tiles
            break;
        }
        if (hipStreamSynchronize(stream) != hipSuccess || sync->failed.load()) {
            if (!sync->failed.load()) { fail("stream synchronisation failed"); sync->abort_all(); }
            break;
        }
        out->render_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rt_get_stats(ctx, &out->stats)) { fail(rt_last_error()); break; }
        if (rank == 0) {
            image->resize(frame_bytes);
            void* last = frames[(size_t)((n_frames - 1) % in_flight)];
            if (hipMemcpy(image->data(), last, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess) { fail("cannot copy the frame to the host"); break; }
        }
    } while (false);
    if (!met) {                                   // (a rank that failed during set-up still keeps the appointment)
        if (out->rc) sync->failed.store(1);
        sync->meet();
    }
    if (ctx) rt_free(ctx);                  // (waits for everything the context has enqueued)
    if (stream) (void)hipStreamDestroy(stream);
    for (void* f : frames) if (f) (void)hipFree(f);
    if (staging) (void)hipFree(staging);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <scene> [out.png] [-w W] [-h H] [-depth D] [-eps E] [-tile ROWS] [-gpus N] [-frames F] [-inflight K] [-packed] [-bvh] [-exact] [-stats]\n",
                argv[0]);
        return 2;
    }
    std::string scene = argv[1], out = "output.png";
    int width = 1000, height = 1000, depth = -1, tile = 0, gpus = 0, n_frames = 1, in_flight = 1;
    float eps = 0.f;
    bool bvh = false, exact = false, stats = false, packed = false;
    int pos = 0;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "-w") width = atoi(need("-w"));
        else if (a == "-h") height = atoi(need("-h"));
        else if (a == "-depth") depth = atoi(need("-depth"));
        else if (a == "-eps") eps = (float)atof(need("-eps"));
        else if (a == "-tile") tile = atoi(need("-tile"));
        else if (a == "-gpus") gpus = atoi(need("-gpus"));
        else if (a == "-frames") n_frames = atoi(need("-frames"));
        else if (a == "-inflight") in_flight = atoi(need("-inflight"));
        else if (a == "-packed") packed = true;
        else if (a == "-bvh") bvh = true;
        else if (a == "-exact") exact = true;
        else if (a == "-stats") stats = true;
        else if (pos++ == 0) out = a;
        else { fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
    }
    if (gpus < 0 || gpus > 64) { fprintf(stderr, "-gpus must be between 1 and 64\n"); return 2; }
    if (n_frames < 1 || in_flight < 1 || in_flight > 8) { fprintf(stderr, "-frames must be at least 1, -inflight between 1 and 8\n"); return 2; }
    if ((n_frames > 1 || in_flight > 1) && gpus == 0) gpus = 1;       // (the frame loop lives in the N-GPU path)
    // Calls in flight run on a hardware queue each, and the stream the exchange is enqueued on should have one of its
    // own; HIP deals a process 4 unless told otherwise, and reads this at its first call.
    // (an environment that exports a smaller number -- HIP's default of 4, say -- is raised as well)
    if (in_flight > 1) {
        const char* have = std::getenv("GPU_MAX_HW_QUEUES");
        if (!have || std::atoi(have) < 16) setenv("GPU_MAX_HW_QUEUES", "16", 1);
    }
    printf("Ray Tracer!\n");
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    auto t_load = now();
    rth_scene* sc = nullptr;
    if (rth_load_scene(&sc, scene.c_str(), width, height, depth, eps)) {
        fprintf(stderr, "%s\n", rth_last_error());
        return 1;
    }
    const double load_s = since(t_load);
    std::vector<uint8_t> image;
    rt_stats st{};
    double secs = 0, upload_s = 0, bvh_s = 0;
    int status = 0;

    if (gpus >= 1) {
        // ---- N GPUs: one host thread and one communicator per GPU ----
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < gpus) {
            fprintf(stderr, "-gpus %d: only %d HIP device(s) visible\n", gpus, n_dev);
            rth_free_scene(sc);
            return 1;
        }
        std::vector<int> devs(gpus);
        for (int g = 0; g < gpus; g++) devs[g] = g;
        std::vector<ncclComm_t> comms(gpus);
        ncclResult_t nr = ncclCommInitAll(comms.data(), gpus, devs.data());
        if (nr != ncclSuccess) {
            fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(nr));
            rth_free_scene(sc);
            return 1;
        }
        printf("Rendering Scene...\n");
        const int rows = tile > 0 ? tile : 16;
        std::vector<RankResult> results(gpus);
        RankSync sync(gpus);
        sync.comms = &comms;
        auto t0 = now();
        {
            std::vector<std::thread> pool;
            for (int g = 0; g < gpus; g++)
                pool.emplace_back(run_rank, g, gpus, comms[g], rth_desc(sc), rows, bvh, exact, n_frames, in_flight, packed, &sync, &image, &results[g]);
            for (auto& th : pool) th.join();
        }
        const double wall = since(t0);         // upload, render, exchange and the copy to the host, all ranks
        if (!sync.aborted.load())              // (an aborted communicator is gone already)
            for (int g = 0; g < gpus; g++) (void)ncclCommDestroy(comms[g]);
        for (int g = 0; g < gpus; g++) {
            if (results[g].rc) { fprintf(stderr, "%s\n", results[g].error.c_str()); status = 1; }
            secs = std::max(secs, results[g].render_s);
            st.rays_primary += results[g].stats.rays_primary; st.rays_reflect += results[g].stats.rays_reflect;
            st.rays_shadow += results[g].stats.rays_shadow; st.hits += results[g].stats.hits;
        }
        st.n_triangles = results[0].stats.n_triangles; st.n_spheres = results[0].stats.n_spheres;
        upload_s = wall - secs;
    } else {
        // ---- one GPU, the current device ----
        auto t_upload = now();
        rt_ctx* ctx = nullptr;
        if (rt_upload_scene(&ctx, rth_desc(sc))) {
            fprintf(stderr, "%s\n", rt_last_error());
            rth_free_scene(sc);
            return 1;
        }
        upload_s = since(t_upload);
        auto t_bvh = now();
        if ((exact && rt_set_option(ctx, RT_OPT_EXACT_ONLY, 1)) || (bvh && rt_set_option(ctx, RT_OPT_USE_BVH, 1))) {
            fprintf(stderr, "%s\n", rt_last_error());
            rt_free(ctx);
            rth_free_scene(sc);
            return 1;
        }
        bvh_s = since(t_bvh);   // the build is synchronous
        printf("Rendering Scene...\n");
        auto t0 = now();
        int rc = tile > 0 ? rt_render_tiles(ctx, 0, tile, 1, nullptr) : rt_render_tile(ctx, 0, height, nullptr);
        image.resize((size_t)width * height * 4);
        if (!rc) rc = rt_download(ctx, image.data(), image.size());
        secs = since(t0);
        if (rc) { fprintf(stderr, "%s\n", rt_last_error()); status = 1; }
        if (!rc && stats && rt_get_stats(ctx, &st)) { fprintf(stderr, "%s\n", rt_last_error()); status = 1; }
        rt_free(ctx);
    }
    if (status) { rth_free_scene(sc); return 1; }
    printf("Scene render completed.\n");
    if (stats) {
        unsigned long long rays = st.rays_primary + st.rays_reflect + st.rays_shadow;
        printf("{\"render_s\": %.6f, \"frames\": %d, \"frames_in_flight\": %d, \"gpus\": %d, \"rays_primary\": %llu, \"rays_reflect\": %llu, \"rays_shadow\": %llu, \"hits\": %llu, "
               "\"Mrays_per_s\": %.3f, \"triangles\": %d, \"spheres\": %d}\n",
               secs, n_frames, in_flight, gpus >= 1 ? gpus : 1, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_reflect,
               (unsigned long long)st.rays_shadow, (unsigned long long)st.hits, rays / secs / 1e6, st.n_triangles, st.n_spheres);
    }
    printf("Writing to file...\n");
    auto t_png = now();
    if (rth_write_png(out.c_str(), image.data(), width, height)) {
        fprintf(stderr, "cannot write %s\n", out.c_str());
        rth_free_scene(sc);
        return 1;
    }
    printf("Image saved as %s.\n", out.c_str());
    if (stats)   // render_s above covers the trace and the download of the frame (N GPUs: the trace and the exchange)
        printf("{\"load_s\": %.6f, \"upload_s\": %.6f, \"bvh_build_s\": %.6f, \"png_s\": %.6f}\n", load_s, upload_s, bvh_s,
               since(t_png));
    rth_free_scene(sc);
    // RT_RENDER_FAST_EXIT: leave without running exit handlers, now that the image is written and everything this
    // program allocated has been given back.  For runs under a sanitizer whose own exit-time bookkeeping trips over
    // the GPU runtime's exit handlers (they free host memory into the sanitizer's quarantine after the sanitizer has
    // shut its device allocator down); whatever the sanitizer finds while the program runs is reported when it happens.
    if (std::getenv("RT_RENDER_FAST_EXIT")) { fflush(stdout); fflush(stderr); _exit(0); }
    return 0;
}
