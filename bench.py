#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflect) on the teapot scene at 4K,
depth 5, on N MI355X of one node.

One process per GPU (torch.distributed over RCCL for N > 1).  A step is one full frame:
every rank traces its interleaved 16-row scanline tiles through the C ABI
(rt_render_tiles) and the tiles are gathered to rank 0.  Inputs (the uploaded scene) are
resident in HBM before the timed region.  Rank 0 prints one JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself: a fresh `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`
child, before this process has touched the GPU, whose output and exit status become this
program's.  Started by torch.distributed.run (WORLD_SIZE set), WORLD_SIZE must equal --gpus.

Legs that run at every world size, through the same tile / exchange path as the headline:
  config4   campanile + teddy, 7680 x 4320, depth 6: brute force and through the LBVH
  config5   1,000,000 triangles + 64 spheres, 16 lights, depth 8, 7680 x 4320 through the LBVH
            (and, on rank 0's GPU alone, the plain FP32 scan of a 512 x 288 frame of it)

  roofline      the dominant kernel, the FP32 filter scan k_filter, in the form in which every
                ray x triangle pair takes the FP32 triangle test (a leg of K frames with
                RT_OPT_BOUNDING_TESTS = 0 after the timed region):
                algorithmic FLOP = ray x primitive pairs scanned x 42 (triangle) or 28
                (sphere), divided by the kernel's summed duration measured with HIP events
                on the launch stream, against the 157.3 TFLOP/s FP32 vector peak.
                The timed region itself runs the default: bounding-sphere tests of 2-12
                multiply-adds per ray in front of the triangle test decide most pairs, so
                its scans are reported as pairs per second ("default_scans"); priced at 42
                FLOP per pair they exceed the peak, which is no statement about the vector
                unit.
  cpu_baseline  the unmodified reference renderer (oracle/_ref, kind "reference") -- or the
                C restatement when that binary is absent (kind "port") -- timed on this
                box's host cores on sub-grids of the same frame (16 threads on every 4th pixel,
                one thread on every 16th).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# Four render calls in flight want two hardware queues each (the two streams of a pass), and the stream this program
# itself enqueues on (joins, the exchange) should not sit on one of them: what it is given would queue up behind a
# whole frame of that lane.  HIP deals a process 4 queues unless told otherwise, and reads this when it starts up --
# before torch is imported.  (4 queues: 4.7-5.1 ms per frame depending on which lane the caller's stream shares a
# queue with; 8: 4.6-5.0; 16: 4.56-4.63.)  The line this program prints records what the library found
# (config.frames_in_flight_queues).
try:
    _queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
except ValueError:
    _queues = 0
if _queues < 16:            # (also when the environment exports HIP's default of 4, as the GPU boxes of this pool do)
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_TRI, FLOP_SPH = 42.0, 28.0          # per ray-primitive test, FMA = 2
PEAK_FP32_VALU = 157.3e12                # 256 CU x 128 lanes x 2 x 2.4 GHz

# Executed FP32 operations per test of each stage of the default (bounded) scans, counted from the
# kernels' arithmetic (rt_kernels.h: filter_range / filter_z2; FMA = 2, multiply / add / reciprocal /
# square root / divide = 1, compare / min / max / select = 0).  Per kind of scan:
#   group  one (bundle of rays, triangle) question of the once-per-64-triangles judgement, per lane
#   bound  one per-ray bounding-sphere test
#   tri    one FP32 triangle test as that scan executes it (the shared origin of camera rays saves the
#          plane distance, the shared direction of a directional light the denominator and its reciprocal;
#          42 is the algorithmic price of the full test by the usual convention, 29 what it executes)
STAGE_FLOP = {
    "camera":             {"group": 25.0, "bound": 5.0, "tri": 23.0},
    "shadow_point":       {"group": 43.0, "bound": 5.0, "tri": 29.0},
    "shadow_directional": {"group": 33.0, "bound": 4.0, "tri": 23.0},
    "reflection":         {"group": 40.0, "bound": 22.0, "tri": 29.0},
    "plain_nearest":      {"group": 0.0, "bound": 0.0, "tri": 29.0},
    "plain_shadow":       {"group": 0.0, "bound": 0.0, "tri": 29.0},
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "teapot.scene"))
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--tile-rows", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-only", action="store_true")
    ap.add_argument("--no-lbvh", action="store_true", help="skip the extra LBVH-mode measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo stages through host memory and exists to "
                         "rehearse the N-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the roofline-stress leg (1,000,000 triangles + 64 spheres, 16 lights, depth 8)")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the campanile + teddy 8K leg (brute force and LBVH)")
    ap.add_argument("--config-size", default="7680x4320",
                    help="frame size of the config4 / config5 legs (tests shrink it; the configurations are quoted at 7680x4320)")
    ap.add_argument("--exchange", default="tiles", choices=["tiles", "packed"],
                    help="end-of-frame exchange for N > 1: one point-to-point transfer per tile straight into rank 0's "
                         "frame rows (tiles), or one per peer through a packed buffer (packed); same bytes either way")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="render calls in flight in the timed region (RT_OPT_FRAMES_IN_FLIGHT, 1..8): frame i+1 is traced "
                         "beside the deep levels of frame i, each in a frame buffer of its own; 1 = one frame at a time; "
                         "0 (default) = 4, or 8 from four ranks on (a rank's share of the frame is a chain of short launches: "
                         "one rank's eighth of the 4K frame takes 1.15 ms per frame with 4 in flight, 1.03 with 8)")
    ap.add_argument("--dump-frame", default=None,
                    help="rank 0 writes the last frame of the timed region (after its exchange) to this .npy file")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream: no overlap of a level's shadow launches with the next level's nearest-hit launches")
    return ap.parse_args()


def start_ranks_if_asked(args):
    """`--gpus N` without a launcher: this process becomes the parent of a torch.distributed.run job of N
    ranks running this same file with the same arguments.  Must be called before anything initialises the
    GPU (a child process, never an exec of a process that holds the device)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: start it as `python bench.py --gpus N` or "
                     f"with --nproc-per-node equal to --gpus")
        return
    if args.gpus <= 1:
        return
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


class Job:
    """This process's place in the N-rank job: rank, device, collective backend and the small reductions."""

    def __init__(self, args, torch, dist):
        self.torch, self.dist, self.backend = torch, dist, args.backend
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        n_dev = torch.cuda.device_count()
        if self.world > 1:
            if args.backend == "nccl" and n_dev < self.world:
                sys.exit(f"bench.py: {self.world} ranks over RCCL need {self.world} GPUs, {n_dev} visible "
                         f"(--backend gloo rehearses the N-rank flow with ranks sharing a GPU)")
            local_rank %= max(1, n_dev)      # (more ranks than devices: a rehearsal on a smaller box)
            torch.cuda.set_device(local_rank)
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(0)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.coll_dev = self.dev if args.backend == "nccl" else torch.device("cpu")      # where the small reductions live
        self.stream = torch.cuda.current_stream().cuda_stream
        self.host_staging = args.backend != "nccl"

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def reduce(self, values, op):
        """max / sum over ranks of a list of numbers (float64)."""
        if self.world == 1:
            return [float(v) for v in values]
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def share_text(self, text):
        """rank 0's string on every rank."""
        if self.world == 1:
            return text
        box = [text]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]


def run_frames(job, args, scene, steps, in_flight, options=()):
    """K frames of `scene` on this job's ranks through the headline's path: interleaved tiles per rank
    (rt_render_tiles), `in_flight` render calls in flight, every frame exchanged to rank 0; timed between
    barriers, the maximum over ranks.  Returns the figures and rank-local statistics."""
    from raytracer_amd import abi
    from raytracer_amd.render import FramePipeline, Renderer, make_exchange
    torch = job.torch
    t0 = time.perf_counter()
    r = Renderer(scene)
    upload_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    for opt, val in options:
        r.set_option(opt, val)
    options_s = time.perf_counter() - t0
    H, W = scene.height, scene.width
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device=job.dev) for _ in range(in_flight)]
    exchange = make_exchange(args.exchange, H, W, args.tile_rows, job.rank, job.world, job.dev, dist=job.dist, torch=torch,
                             host_staging=job.host_staging)
    pipe = FramePipeline(r, frames, exchange, job.rank, job.world, args.tile_rows, job.stream)

    def fence():
        pipe.drain()
        job.barrier()

    pipe.set_in_flight(in_flight)
    for _ in range(max(1, in_flight)):          # (every lane's workspace is in place)
        pipe.step()
    fence()
    r.reset_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step()
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()
    rays = st["rays_primary"] + st["rays_reflect"] + st["rays_shadow"]
    elapsed, = job.reduce([elapsed], "max")
    rays_total, = job.reduce([rays], "sum")
    out = {"frame": f"{W}x{H}", "depth": scene.desc.max_depth, "steps": steps, "frames_in_flight": in_flight,
           "ms_per_step": elapsed / steps * 1e3, "rays_per_frame": rays_total / steps,
           "Mrays_per_s": rays_total / elapsed / 1e6, "upload_s": upload_s, "options_s": options_s,
           "triangles": st["n_triangles"], "spheres": st["n_spheres"], "lights": scene.desc.n_lights,
           "flagged_rays": st["flagged_rays"], "filter_form": st["filter_form"]}
    if st["bvh_node_tests"]:
        out.update({"in_plane_origin_scan": "on" if st["bvh_plane_guard"] else "off (the calibration frame found no ray that needs it)",
                    "box_tests_per_ray": st["bvh_node_tests"] / max(1, rays), "leaf_tests_per_ray": st["bvh_leaf_tests"] / max(1, rays),
                    "exact_retests_per_ray": st["exact_tests"] / max(1, rays)})
    r.close()
    del frames
    torch.cuda.empty_cache()
    return out


def config4_leg(job, args, steps, in_flight):
    """campanile + teddy (23,243 triangles, 48 of them zero-area: the reference shadows every point, the frame is
    ambient light only), 7680 x 4320, depth 6, split over this job's ranks by interleaved tiles: brute force and
    through the LBVH.  A zero-area triangle is in the way of every shadow ray (raytracer.cpp:313, 553) and the library
    knows so from the upload: its shadow launches mark the rays and test nothing.  The third leg takes that knowledge
    away (RAYTRACER_AMD_TEST_VOID_NORMALS=1: every shadow ray scanned against every triangle, as in a scene without
    such a triangle)."""
    from raytracer_amd import Scene, abi
    cw, ch = (int(v) for v in args.config_size.split("x"))
    sc = Scene.load(os.path.join(ROOT, "scenes", "campanile_teddy.scene"), cw, ch, 6)
    out = {"scene": "campanile_teddy.scene", "n_gpus": job.world,
           "shadow_rays": "answered at upload: a zero-area triangle shadows every ray, none is tested",
           "brute_force": run_frames(job, args, sc, steps, in_flight),
           "lbvh": run_frames(job, args, sc, steps, in_flight, options=[(abi.RT_OPT_USE_BVH, 1)])}
    os.environ["RAYTRACER_AMD_TEST_VOID_NORMALS"] = "1"
    try:
        out["brute_force_shadow_rays_scanned"] = run_frames(job, args, sc, steps, in_flight)
    finally:
        del os.environ["RAYTRACER_AMD_TEST_VOID_NORMALS"]
    return out


def cpu_baseline(args, gpu_rays_fn):
    """Times the reference CPU renderer on bounded samples of the same workload: the frame at
    1/k x 1/k resolution is exactly every k-th pixel of the full grid in x and y (the camera
    samples pixel corners: u = x / W).  Two legs: all the threads this box grants (k = 4) and
    one thread, the reference as shipped (k = 16)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_render")

    def leg(k, threads):
        w, h = max(1, args.width // k), max(1, args.height // k)
        rays = gpu_rays_fn(w, h)
        if os.path.exists(ref):
            out = subprocess.run([ref, args.scene, str(w), str(h), str(args.depth), "-", str(threads)],
                                 check=True, capture_output=True, text=True).stdout
            secs = json.loads(out.strip().splitlines()[-1])["render_s"]
            kind = "reference"
        else:
            from raytracer_amd import Scene
            from tests import oracle_api
            sc = Scene.load(args.scene, w, h, args.depth)
            t0 = time.perf_counter()
            oracle_api.Oracle().render(sc, threads=threads)
            secs = time.perf_counter() - t0
            kind = "port"
        return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": kind,
                "sample": f"{w}x{h} sub-grid (every {k}th pixel in x and y) of the {args.width}x{args.height} frame, "
                          f"{rays} rays, {secs:.2f} s"}

    out = leg(4, min(16, os.cpu_count() or 1))
    one = leg(16, 1)
    out["single_thread"] = {"value": one["value"], "unit": one["unit"], "sample": one["sample"]}
    return out


def config5_leg(job, args, steps, in_flight):
    """The roofline-stress configuration: a tessellated height field of 1,000,000 triangles + 64 reflective
    spheres (every 4th an ellipsoid), 16 lights, depth 8 (tools/make_synthetic.py writes it as .scene / .obj
    text, no random numbers).
      plain_scan  on rank 0's GPU alone: a 512 x 288 frame with RT_OPT_BOUNDING_TESTS = 0, every ray x triangle
                  pair takes the FP32 triangle test (the scene takes the filter's per-test margin form) -- the
                  FP32 roofline fraction of the brute-force scan on a mesh 158 times the teapot's size;
      lbvh_8k     the full 7680 x 4320 frame through the GPU-built LBVH, split over this job's ranks by interleaved
                  tiles and exchanged to rank 0 like the headline's frames (brute force at that size is 3 x 10^14 pairs)."""
    import shutil
    import tempfile
    from raytracer_amd import Scene, abi
    from raytracer_amd.render import Renderer
    out = {"n_gpus": job.world}
    tmp = None
    if job.rank == 0:
        tmp = tempfile.mkdtemp(prefix="rt_config5_")
        t0 = time.perf_counter()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic.py"), tmp], check=True, stdout=subprocess.DEVNULL)
        out["host"] = {"generate_text_s": time.perf_counter() - t0}
    tmp = job.share_text(tmp)                     # (one node: every rank reads rank 0's files)
    path = os.path.join(tmp, "synthetic.scene")
    try:
        if job.rank == 0:
            t1 = time.perf_counter()
            sc = Scene.load(path, 512, 288, 8)
            t2 = time.perf_counter()
            r = Renderer(sc)
            t3 = time.perf_counter()
            out["host"].update({"parse_s": t2 - t1, "upload_s": t3 - t2})
            r.set_option(abi.RT_OPT_BOUNDING_TESTS, 0)
            r.set_option(abi.RT_OPT_PROFILE, 1)
            r.render_frame()
            r.reset_stats()
            tp = time.perf_counter()
            r.render_frame()
            st = r.stats()
            wall = time.perf_counter() - tp
            n_tri, n_sph = st["n_triangles"], st["n_spheres"]
            tri_share = n_tri / (n_tri + n_sph)
            ms = st["closest_ms"] + st["anyhit_ms"]
            tests = (st["closest_tests"] + st["anyhit_tests"]) * tri_share
            rays = st["rays_primary"] + st["rays_reflect"] + st["rays_shadow"]
            out["plain_scan"] = {"frame": "512x288", "gpus": 1, "triangles": n_tri, "spheres": n_sph, "filter_form": st["filter_form"],
                                 "rays": rays, "ray_triangle_tests": tests, "filter_kernels_ms": ms, "frame_ms": wall * 1e3,
                                 "achieved_tflops": tests * FLOP_TRI / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                 "frac_of_peak": tests * FLOP_TRI / (ms * 1e-3) / PEAK_FP32_VALU if ms > 0 else 0.0,
                                 "reference_precision_kernels_ms": st["exact_ms"],
                                 "note": "42 FLOP per ray x triangle test; reference_precision_kernels_ms holds the listed pairs "
                                         "and the spheres (FP32 pre-test, survivors in the reference's arithmetic)"}
            # the same small frame with the default bounding tests
            r.set_option(abi.RT_OPT_BOUNDING_TESTS, 1)
            r.set_option(abi.RT_OPT_PROFILE, 0)
            r.render_frame()
            r.reset_stats()
            tp = time.perf_counter()
            r.render_frame()
            r.stats()
            out["bounded_scan"] = {"frame": "512x288", "gpus": 1, "frame_ms": (time.perf_counter() - tp) * 1e3}
            r.close()
        job.barrier()
        cw, ch = (int(v) for v in args.config_size.split("x"))
        sc8 = Scene.load(path, cw, ch, 8)
        out["lbvh_8k"] = run_frames(job, args, sc8, steps, in_flight, options=[(abi.RT_OPT_USE_BVH, 1)])
        out["lbvh_8k"]["bvh_build_s"] = out["lbvh_8k"].pop("options_s")
    finally:
        job.barrier()
        if job.rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def pmc_of_kernel(mode, bounding_test):
    """The committed PMC passes of this same command (profiles/*pmc_summary.csv, written by tools/pmc_passes.sh +
    tools/pmc_summary.py), for the scan kernel k_filter<MODE, rays per lane, form, bounding test, counting = false[, in-plane-origin machinery]>:
    (file name, {counter: sum over the DISPATCHES launches of one pass}).  (None, {}) when there is no such file or kernel."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.csv")), key=os.path.basename)   # (named per round: the last one)
    if not files:
        return None, {}
    found = {}
    for r in csv.DictReader(open(files[-1])):
        name = r["kernel"].replace(" ", "")
        if not (name.startswith("k_filter<") and name.endswith(">")):
            continue
        targs = name[len("k_filter<"):-1].split(",")
        if len(targs) in (5, 6) and targs[0] == str(mode) and targs[3] == str(bounding_test) and targs[4] == "false":
            found.setdefault(name, {})[r["counter"]] = float(r["sum_over_dispatches"])
    if not found:
        return os.path.basename(files[-1]), {}
    # (a sixth argument tells the instance with the in-plane-origin machinery from the one without: the timed configuration
    # runs the one with the most launches in that command)
    got = max(found.values(), key=lambda g: g.get("DISPATCHES", 0.0))
    return os.path.basename(files[-1]), got


def measured_traffic(mode, bounding_test):
    """HBM bytes per launch of that kernel from those passes: FETCH_SIZE is in KiB and reads half the bytes of a wide
    stream on gfx950 (doubled here), WRITE_SIZE is in KiB and exact.  None when the passes do not hold them."""
    _, got = pmc_of_kernel(mode, bounding_test)
    if not all(k in got for k in ("FETCH_SIZE", "WRITE_SIZE", "DISPATCHES")) or got["DISPATCHES"] <= 0:
        return None
    return (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0 / got["DISPATCHES"]


def pmc_occupancy(mode, bounding_test, kernel_ms_avg, clock_khz, n_cu):
    """How full the chip was while that kernel ran, from the same passes and THIS run's average launch duration:
    waves resident per SIMD (SQ_WAVE_CYCLES counts in units of 4 cycles, summed over waves) and the share of SIMD
    cycles in which a vector instruction executed (SQ_ACTIVE_INST_VALU, same unit), over n_cu x 4 SIMDs x clock x
    duration; SQ_WAIT_ANY / SQ_WAVE_CYCLES is the share of a resident wave's time spent waiting."""
    name, got = pmc_of_kernel(mode, bounding_test)
    need = ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "DISPATCHES")
    if not all(k in got for k in need) or got["DISPATCHES"] <= 0 or not clock_khz or kernel_ms_avg <= 0:
        return None
    simd_cycles = n_cu * 4 * clock_khz * 1e3 * kernel_ms_avg * 1e-3
    return {"source": f"profiles/{name} (counters of a separate rocprofv3 --pmc run of this command; durations from this run)",
            "waves_per_simd": 4.0 * got["SQ_WAVE_CYCLES"] / got["DISPATCHES"] / simd_cycles,
            "valu_busy": 4.0 * got["SQ_ACTIVE_INST_VALU"] / got["DISPATCHES"] / simd_cycles,
            "wait_share_of_wave_cycles": got["SQ_WAIT_ANY"] / got["SQ_WAVE_CYCLES"] if got["SQ_WAVE_CYCLES"] else None}


def main():
    args = parse()
    start_ranks_if_asked(args)              # (--gpus N without a launcher: this process only waits for its ranks)
    import torch
    import torch.distributed as dist
    from raytracer_amd import Scene, abi
    from raytracer_amd.render import FramePipeline, Renderer, make_exchange

    job = Job(args, torch, dist)
    world, rank, dev, coll_dev = job.world, job.rank, job.dev, job.coll_dev

    scene = Scene.load(args.scene, args.width, args.height, args.depth)
    r = Renderer(scene)
    if args.exact_only:
        r.set_option(abi.RT_OPT_EXACT_ONLY, 1)
    in_flight = max(1, min(8, args.frames_in_flight)) if args.frames_in_flight > 0 else (8 if world >= 4 else 4)
    # one frame buffer per render call in flight: frame i is exchanged while frame i+1 is traced
    frames = [torch.zeros((args.height, args.width, 4), dtype=torch.uint8, device=dev) for _ in range(in_flight)]
    stream = job.stream
    exchange = make_exchange(args.exchange, args.height, args.width, args.tile_rows, rank, world, dev, dist=dist, torch=torch,
                             host_staging=job.host_staging)
    pipe = FramePipeline(r, frames, exchange, rank, world, args.tile_rows, stream)
    step, drain, set_in_flight = pipe.step, pipe.drain, pipe.set_in_flight

    def fence():
        drain()
        job.barrier()

    if args.no_overlap:
        r.set_option(abi.RT_OPT_OVERLAP, 0)
    # One frame at a time first (its time is the latency of a frame), then the timed region with frames in flight.
    step()
    fence()
    tl = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    one_at_a_time_ms = (time.perf_counter() - tl) / args.steps * 1e3
    set_in_flight(in_flight)
    for _ in range(max(args.warmup, in_flight if in_flight > 1 else 0)):      # (every lane's workspace is in place)
        step()
    fence()
    r.reset_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st_timed = r.stats()
    lane_deal = {"lanes": st_timed["lanes"], "hardware_queues_found": st_timed["lane_hw_queues_found"],
                 "lanes_on_distinct_queues": "yes" if not st_timed["lanes_share_a_queue"] else "no",
                 "callers_stream_on_a_queue_of_its_own": {0: "yes", 1: "no"}.get(st_timed["caller_stream_shares_a_queue"], "unknown"),
                 "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    if args.dump_frame and rank == 0:
        import numpy as np
        np.save(args.dump_frame, pipe.last_frame().cpu().numpy())
    set_in_flight(1)

    # Kernel durations for the roofline: the same K frames once more with HIP events around every
    # trace-kernel launch (recorded by the library on the launch stream).  Event timing serialises
    # the library's two streams, so that a kernel's duration is its own and not that of two
    # kernels sharing the chip; the headline above is measured without it.
    r.reset_stats()
    r.set_option(abi.RT_OPT_PROFILE, 1)
    fence()
    tp = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    serial_elapsed = time.perf_counter() - tp
    st_default = r.stats()
    # ... and once more with the bounding tests off: the scan in which every pair takes the FP32
    # triangle test, the kernel the FP32 roofline is quoted for.
    if not args.exact_only:
        r.set_option(abi.RT_OPT_BOUNDING_TESTS, 0)
        step()
        fence()
        r.reset_stats()
        fence()
        for _ in range(args.steps):
            step()
        fence()
        st = r.stats()
        r.set_option(abi.RT_OPT_BOUNDING_TESTS, 1)
    else:
        st = st_default
    r.set_option(abi.RT_OPT_PROFILE, 0)

    # One more frame of the default configuration in the counting instances of the scan kernels
    # (RT_OPT_WORK_COUNTERS): the work the timed frames execute, stage by stage.
    work = None
    if not args.exact_only:
        r.reset_stats()
        r.set_option(abi.RT_OPT_WORK_COUNTERS, 1)
        step()
        fence()
        work = r.stats()["scan_work"]
        r.set_option(abi.RT_OPT_WORK_COUNTERS, 0)

    # The same frames again through the optional LBVH (same rays, same pixels; reported beside
    # the brute-force headline, which is what the roofline describes).
    lbvh = None
    if not args.no_lbvh and not args.exact_only:
        r.set_option(abi.RT_OPT_USE_BVH, 1)
        step()
        fence()
        tb = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        lb_one_ms = (time.perf_counter() - tb) / args.steps * 1e3
        set_in_flight(in_flight)
        for _ in range(max(1, in_flight if in_flight > 1 else 0)):
            step()
        fence()
        r.reset_stats()
        fence()
        tb = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        lb_elapsed = time.perf_counter() - tb
        sb = r.stats()
        set_in_flight(1)
        r.set_option(abi.RT_OPT_USE_BVH, 0)
        lb_rays = sb["rays_primary"] + sb["rays_reflect"] + sb["rays_shadow"]
        if world > 1:
            tt = torch.tensor([lb_elapsed, float(lb_rays), lb_one_ms], dtype=torch.float64, device=coll_dev)
            mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            lb_elapsed, lb_rays_total, lb_one_ms = float(mx[0].item()), float(sm[1].item()), float(mx[2].item())
        else:
            lb_rays_total = float(lb_rays)
        lbvh = {"value": lb_rays_total / lb_elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": lb_elapsed / args.steps * 1e3,
                "frames_in_flight": in_flight, "one_frame_at_a_time_ms": lb_one_ms,
                "box_tests_per_ray": sb["bvh_node_tests"] / max(1, lb_rays),
                "leaf_tests_per_ray": sb["bvh_leaf_tests"] / max(1, lb_rays),
                "exact_retests_per_ray": sb["exact_tests"] / max(1, lb_rays),
                "in_plane_origin_scan": "on" if sb["bvh_plane_guard"] else "off (the calibration frame found no ray that needs it)",
                "note": "frames byte-identical to brute force (tests/test_gpu_bvh.py); the in-plane-origin hit that no box hierarchy finds "
                        "(raytracer.cpp:319-321) is covered by a scan behind the traversals, kept only when a calibration frame of "
                        "this scene and camera shows a ray that takes that branch (RT_OPT_IN_PLANE_ORIGIN in include/rt_abi.h)"}

    rays_local = st_timed["rays_primary"] + st_timed["rays_reflect"] + st_timed["rays_shadow"]
    if world > 1:
        t = torch.tensor([elapsed, one_at_a_time_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, one_at_a_time_ms = float(t[0].item()), float(t[1].item())
        cnt = torch.tensor([rays_local, st_timed["rays_primary"], st_timed["rays_reflect"], st_timed["rays_shadow"]],
                           dtype=torch.int64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_total, n_primary, n_reflect, n_shadow = [int(v) for v in cnt.tolist()]
    else:
        rays_total, n_primary, n_reflect, n_shadow = (rays_local, st_timed["rays_primary"], st_timed["rays_reflect"],
                                                      st_timed["rays_shadow"])

    # The headline's renderer gives its workspaces back before the 8K legs take theirs.
    n_cu, clock_khz = r.device_info()
    r.close()
    del frames, pipe
    torch.cuda.empty_cache()
    leg_steps = max(1, min(3, args.steps))
    # The headline once more with the in-plane-origin question asked of every ray whatever the calibration frame found
    # (RT_OPT_IN_PLANE_ORIGIN = 1): what the question costs, and the figure for a scene that needs it.
    always_asked = None
    if not args.exact_only and not args.no_lbvh:
        from raytracer_amd import abi as _abi
        a = run_frames(job, args, scene, args.steps, in_flight, options=[(_abi.RT_OPT_IN_PLANE_ORIGIN, 1)])
        always_asked = {"ms_per_step": a["ms_per_step"], "Mrays_per_s": a["Mrays_per_s"], "steps": a["steps"],
                        "frames_in_flight": a["frames_in_flight"],
                        "note": "same frames; the timed configuration asks the question only if the calibration frame of the "
                                "scene and camera shows a pair that takes that branch (config.in_plane_origin_question)"}
    config4 = config4_leg(job, args, leg_steps, in_flight) if not args.no_config4 and not args.exact_only else None
    config5 = config5_leg(job, args, leg_steps, in_flight) if not args.no_config5 and not args.exact_only else None

    if rank == 0:
        n_tri, n_sph = st["n_triangles"], st["n_spheres"]
        flop_per_test = (FLOP_TRI * n_tri + FLOP_SPH * n_sph) / max(1, n_tri + n_sph)
        kernels = {"k_filter<0, 8, form, 0> (nearest hit, every pair takes the FP32 triangle test)":
                       (st["closest_ms"], st["closest_tests"], st["closest_launches"]),
                   "k_filter<1, 8, form, 0> (shadow, every pair takes the FP32 triangle test)":
                       (st["anyhit_ms"], st["anyhit_tests"], st["anyhit_launches"])}
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, tests, launches = kernels[dom]
        achieved = tests * flop_per_test / (ms * 1e-3) if ms > 0 else 0.0
        both_ms = st["closest_ms"] + st["anyhit_ms"]
        both = (st["closest_tests"] + st["anyhit_tests"]) * flop_per_test / (both_ms * 1e-3) if both_ms > 0 else 0.0
        n_prim = n_tri + n_sph
        sd = st_default
        cam_tests = sd["rays_primary"] * n_prim if sd["camera_launches"] else 0

        def scan(ms_sum, pairs, n_launches):
            return {"ms_per_step": ms_sum / args.steps, "launches": n_launches,
                    "pairs_per_s": pairs / (ms_sum * 1e-3) if ms_sum > 0 else 0.0,
                    "at_42_flop_per_pair_tflops": pairs * flop_per_test / (ms_sum * 1e-3) / 1e12 if ms_sum > 0 else 0.0}
        default_scans = {
            "camera_rays": scan(sd["camera_ms"], cam_tests, sd["camera_launches"]),
            "reflection_rays": scan(sd["closest_ms"] - sd["camera_ms"], sd["closest_tests"] - cam_tests,
                                    sd["closest_launches"] - sd["camera_launches"]),
            "shadow_rays": scan(sd["anyhit_ms"], sd["anyhit_tests"], sd["anyhit_launches"]),
            "filter_ms_per_step": (sd["closest_ms"] + sd["anyhit_ms"]) / args.steps,
            "reference_precision_kernels_ms_per_step": sd["exact_ms"] / args.steps,
            "exact_retests_per_ray": sd["exact_tests"] / max(1, rays_local),
            "note": "the timed configuration: bounding-sphere tests in front of the triangle test; the 42-FLOP price "
                    "of a pair is the triangle test's, so these figures are not roofline fractions"}
        # Executed work of the timed configuration: counts of one frame (they do not vary from frame to
        # frame) x the stage prices above, over the scan's duration per frame in the serialised leg.
        executed = None
        if work is not None:
            def merged(kinds):
                out = {}
                for k in kinds:
                    for name, v in work[k].items():
                        out[name] = out.get(name, 0) + v
                return out

            def executed_scan(kinds, ms_per_frame, nominal_pairs):
                w = merged(kinds)
                flop = sum(work[k]["group_tests"] * STAGE_FLOP[k]["group"] + work[k]["ray_bound_tests"] * STAGE_FLOP[k]["bound"] +
                           work[k]["tri_tests"] * STAGE_FLOP[k]["tri"] for k in kinds)
                tflops = flop / (ms_per_frame * 1e-3) / 1e12 if ms_per_frame > 0 else 0.0
                return {"ms_per_step": ms_per_frame, "nominal_pairs_per_step": nominal_pairs,
                        "executed_per_step": {"tasks": w["tasks"], "groups_of_64_judged": w["groups"],
                                              "groups_live": w["groups_live"], "group_tests": w["group_tests"],
                                              "batches_of_4_evaluated": w["batches"], "per_ray_bounding_tests": w["ray_bound_tests"],
                                              "batches_with_a_triangle_test": w["batches_tested"],
                                              "fp32_triangle_tests": w["tri_tests"],
                                              "fp32_triangle_tests_in_lanes_that_asked": w["tri_tests_wanted"],
                                              "pairs_listed": w["listed"]},
                        "executed_flop_per_step": flop, "executed_tflops": tflops, "frac_of_peak": tflops * 1e12 / PEAK_FP32_VALU}
            sms, steps = sd["scan_ms"], max(1, args.steps)
            executed = {
                "camera_rays": executed_scan(["camera"], sms["camera"] / steps, cam_tests // steps),
                "reflection_rays": executed_scan(["reflection"], sms["reflection"] / steps,
                                                 (sd["closest_tests"] - cam_tests) // steps),
                "shadow_rays": executed_scan(["shadow_point", "shadow_directional"],
                                             (sms["shadow_point"] + sms["shadow_directional"]) / steps,
                                             sd["anyhit_tests"] // steps),
                "stage_flop": STAGE_FLOP,
                "note": "counts from one frame in the counting instances of the scan kernels (RT_OPT_WORK_COUNTERS), "
                        "durations from the serialised leg (HIP events); whole-wave counts: a test a wave executes "
                        "for one asking lane occupies all 64"}
        device_peak = (n_cu * 128 * 2 * clock_khz * 1e3) if clock_khz else None
        line = {
            "metric": "Mrays/s (primary+shadow+reflect) teapot 4K depth-5",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "one_frame_at_a_time_ms": one_at_a_time_ms,
            "one_frame_at_a_time_value": rays_total / args.steps / (one_at_a_time_ms * 1e-3) / 1e6,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 filter + f64/f32 reference-precision re-test",
            "data": "synthetic (teapot.obj mesh + 2 floor triangles, authored scene file)",
            "config": {"workload": f"{os.path.basename(args.scene)} {args.width}x{args.height} depth {args.depth}",
                       "streams": "one" if args.no_overlap else "two (shadow launches beside the next level's nearest-hit launches)",
                       "frames_in_flight": in_flight,
                       "frames_in_flight_note": "render call i+1 is enqueued while call i runs (RT_OPT_FRAMES_IN_FLIGHT): the short "
                                                "launches of a frame's deep levels share the chip with the next frame's large ones; "
                                                "every frame is traced in full into a buffer of its own and exchanged, none is skipped "
                                                "or reused; ms_per_step is time per frame at that throughput, one_frame_at_a_time_ms the "
                                                "same K frames with one call in flight",
                       "frames_in_flight_queues": lane_deal,
                       "triangles": n_tri, "spheres": n_sph, "lights": scene.desc.n_lights,
                       "filter_form": {1: "static margins", 2: "per-test margins"}.get(st_timed["filter_form"], st_timed["filter_form"]),
                       "in_plane_origin_question": ("asked" if st_timed["bvh_plane_guard"] else
                                                    "not asked: the calibration frame of this scene and camera, traced with the question "
                                                    "asked of every ray, found no pair that takes that branch of Triangle::intersection "
                                                    "(raytracer.cpp:319-321; RT_OPT_IN_PLANE_ORIGIN in include/rt_abi.h); frames equal "
                                                    "the reference's either way"),
                       "tile_rows": args.tile_rows, "parallelism": f"interleaved scanline tiles x{world}",
                       "rays_per_frame": {"primary": n_primary // args.steps, "reflect": n_reflect // args.steps,
                                          "shadow": n_shadow // args.steps,
                                          "note": "traced = reference-equivalent: every ray the reference's traceRay / "
                                                  "traceShadowRay would be called with, no path pruned"}},
            "roofline": None,
        }
        plain = {"kernel": dom, "achieved": achieved / 1e12, "peak": PEAK_FP32_VALU / 1e12, "unit": "TFLOP/s",
                 "frac": achieved / PEAK_FP32_VALU,
                 "what": "the scan in which every ray x triangle pair takes the FP32 triangle test (RT_OPT_BOUNDING_TESTS = 0), "
                         "priced at the algorithmic 42 FLOP per test; NOT the timed configuration (it takes %.1f ms per frame)"
                         % (both_ms / args.steps),
                 # (the committed PMC pass is a 1-GPU run: its per-launch bytes describe N = 1 only)
                 "traffic": measured_traffic(0 if "<0," in dom else 1, 0) if world == 1 and launches else None,
                 "traffic_source": "replayed from the committed PMC summary under profiles/ (not measured in this run)",
                 "kernel_ms_avg": ms / max(1, launches), "launches": launches,
                 "filter_form": {1: "static margins", 2: "per-test margins"}.get(st["filter_form"], st["filter_form"]),
                 "flop_per_test": flop_per_test, "tests_per_launch": tests / max(1, launches),
                 "both_filter_kernels_tflops": both / 1e12,
                 "filter_ms_per_step": both_ms / args.steps,
                 "reference_precision_kernels_ms_per_step": st["exact_ms"] / args.steps,
                 "exact_retests_per_ray": st["exact_tests"] / max(1, rays_local),
                 "nominal_tests_per_step": (st["closest_tests"] + st["anyhit_tests"]) / max(1, args.steps),
                 "executed_reference_precision_tests_per_step": st["exact_tests"] / max(1, args.steps)}
        roof = {"bound": "valu_fp32", "peak": PEAK_FP32_VALU / 1e12, "unit": "TFLOP/s",
                "peak_from_device_properties": {"compute_units": n_cu, "clock_mhz": clock_khz / 1e3 if clock_khz else None,
                                                "tflops": device_peak / 1e12 if device_peak else None},
                "timing": "HIP events on the launch stream, K frames after the timed region with the library's two streams "
                          "serialised (the default configuration takes %.2f ms per frame serialised)" % (serial_elapsed / args.steps * 1e3),
                "traffic": None,
                "default_scans": default_scans, "plain_scan": plain}
        if executed is not None:
            names = {"camera_rays": "k_filter<0, %d, form, 1> (camera rays)" % 8,
                     "reflection_rays": "k_filter<0, %d, form, 4> (reflection rays)" % 4,
                     "shadow_rays": "k_filter<1, %d, form, 2> (shadow rays)" % 8}
            pmc_key = {"camera_rays": (0, 1), "reflection_rays": (0, 4), "shadow_rays": (1, 2)}
            dom_scan = max(names, key=lambda k: executed[k]["ms_per_step"])
            # the same kernel by the per-unit formula (42 FLOP x every ray x triangle pair of its launches, no culling):
            # more than the peak, because the bounding hierarchy decides most pairs without the triangle test
            dom_ms = executed[dom_scan]["ms_per_step"]
            nominal_frac = (executed[dom_scan]["nominal_pairs_per_step"] * flop_per_test / (dom_ms * 1e-3) / PEAK_FP32_VALU
                            if dom_ms > 0 else None)
            launches_per_step = {"camera_rays": sd["scan_launches"]["camera"], "reflection_rays": sd["scan_launches"]["reflection"],
                                 "shadow_rays": sd["scan_launches"]["shadow_point"] + sd["scan_launches"]["shadow_directional"]}[dom_scan] / steps
            rays_of = {"camera_rays": sd["rays_primary"], "reflection_rays": sd["rays_reflect"], "shadow_rays": sd["rays_shadow"]}[dom_scan] / steps
            ray_bytes = {"camera_rays": 24.0, "reflection_rays": 48.0, "shadow_rays": 24.0}[dom_scan]
            algorithmic_bytes = ((rays_of * ray_bytes + executed[dom_scan]["executed_per_step"]["pairs_listed"] * 8.0) / max(1.0, launches_per_step)
                                 + n_tri * 96.0)
            traffic = measured_traffic(*pmc_key[dom_scan]) if world == 1 else None
            roof.update({"kernel": names[dom_scan] + ": the scan kernel of the timed configuration with the largest share of "
                                                      "the frame; executed FP32 operations of its three stages (bundle judgement, "
                                                      "per-ray bounding test, triangle test) over its duration",
                         "achieved": executed[dom_scan]["executed_tflops"], "frac": executed[dom_scan]["frac_of_peak"],
                         "nominal_frac": nominal_frac,
                         "nominal_frac_note": "culled: 42 FLOP x (rays x triangles) of the kernel's launches over its duration, as if "
                                              "every pair took the triangle test; above 1 because the bounding tests decide most pairs",
                         "launches_per_step": launches_per_step, "kernel_ms_avg": dom_ms / max(1.0, launches_per_step),
                         # (the committed PMC pass is a 1-GPU run: its per-launch bytes describe N = 1 only)
                         "traffic": traffic,
                         "traffic_source": "HBM bytes per launch, replayed from the committed PMC summary under profiles/ "
                                           "(not measured in this run)",
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "algorithmic_bytes_note": "rays x 48 B (24 B for camera and shadow rays: one vector) + listed pairs x 8 B, per "
                                                   "launch, + the triangle tables once (96 B per triangle)",
                         "traffic_ratio": traffic / algorithmic_bytes if traffic and algorithmic_bytes else None,
                         "pmc": pmc_occupancy(*pmc_key[dom_scan], dom_ms / max(1.0, launches_per_step), clock_khz, n_cu) if world == 1 else None,
                         "timed_configuration": executed})
        else:
            roof.update({"kernel": plain["kernel"], "achieved": plain["achieved"], "frac": plain["frac"]})
        line["roofline"] = roof
        if lbvh is not None:
            line["lbvh"] = lbvh
        if config4 is not None:
            line["config4"] = config4
        if always_asked is not None:
            line["in_plane_origin_question_always_asked"] = always_asked
        if config5 is not None:
            line["config5"] = config5
        line["config"]["exchange"] = ("one transfer per tile, straight into rank 0's frame rows" if args.exchange == "tiles"
                                      else "one packed transfer per peer") if world > 1 else "none (one rank)"
        # (rank 0 times the CPU reference while the other ranks wait at the barrier below)
        if not args.no_cpu_baseline:
            def gpu_rays(w, h):
                sc = Scene.load(args.scene, w, h, args.depth)
                rr = Renderer(sc)
                rr.render_frame()
                s2 = rr.stats()
                rr.close()
                return s2["rays_primary"] + s2["rays_reflect"] + s2["rays_shadow"]
            line["cpu_baseline"] = cpu_baseline(args, gpu_rays)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
