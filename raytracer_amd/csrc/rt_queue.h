// The data a pass moves between its kernels: ray queues, per-level hit records, the order of the pixels in the
// queue, the counters -- and the kernel that makes the camera rays (k_primary).
#pragma once

#include "rt_device.h"

#define RT_BLOCK 256
#define RT_WAVE 64
#define RT_CAND_CAP 64          // per-lane candidate staging slots in LDS (shared by the lane's rays)
#define RT_BATCH 4              // triangles filtered between two capacity checks
#ifndef RT_FILTER_MIN_WAVES
#define RT_FILTER_MIN_WAVES 1     // waves per SIMD the filter kernel is compiled for (a higher value caps its VGPRs)
#endif
#define RT_RPL 8                // rays per lane in the filter kernel
#ifndef RT_RPL_SHARED
#define RT_RPL_SHARED 8         // ... in its instances for rays that share a point or a direction
#endif
#ifndef RT_GROUPS_AHEAD
#define RT_GROUPS_AHEAD 2       // groups of 64 triangles whose bounding records a wave requests together
#endif
#ifndef RT_RPL_ANY
#define RT_RPL_ANY 4            // ... in its instance with a bounding test for rays with nothing in common (256 rays per
                                // task ask for fewer triangles than 512; the scans of camera and shadow rays lose by it)
#endif
// a staged candidate (u16) = triangle index relative to a chunk base | ray slot << shift
#define RT_PAIR_LISTS 64        // sub-lists of the pair list at most (see TraceArgs::pair_count)
// A trace launch's counters (the header: RT_PAIR_LISTS fill counters of the pair list on a 128-byte line each, then a
// line with the task counter, "some ray is flagged" and the launch's number of rays) and its per-ray flags.  A workspace
// keeps three headers and two flag arrays per kind of query and launches take them in turn, so that every launch finds
// its own at zero without a fill in front of it: launch x clears header x + 1 and, if launch x - 1 flagged rays, that
// launch's flags (see k_brute).
#define RT_FLAGS_HEADER (RT_PAIR_LISTS * 128 + 128)
#define RT_SLOT_SHIFT(RPL) ((RPL) <= 2 ? 15 : ((RPL) <= 4 ? 14 : 13))

struct RayQueue {               // structure of arrays, one entry per ray
    double *ox, *oy, *oz, *dx, *dy, *dz;
    // FP32 scan copies (null: none kept): what the bounded scans of k_filter hold per ray -- origin, direction
    // and the three per-ray values of the bounding test -- written once by the kernel that makes the ray
    // (k_primary, k_hit) instead of being derived from the doubles again by every task that scans a part of
    // the triangles for it.  Three arrays of scan_stride entries, one after the other: float4 (px py pz hx),
    // float4 (vx vy vz hy), float hz -- three coalesced loads per ray from three uniform bases.
    float* scan;
    uint32_t scan_stride;
};
__device__ __forceinline__ void scan_store(const RayQueue& q, unsigned int i, float px, float py, float pz, float vx, float vy,
                                           float vz, float hx, float hy, float hz) {
    float4* a = (float4*)q.scan;
    float4* b = a + q.scan_stride;
    float* c = (float*)(b + q.scan_stride);
    a[i] = make_float4(px, py, pz, hx);
    b[i] = make_float4(vx, vy, vz, hy);
    c[i] = hz;
}

// The per-ray values of the bounding tests, in ONE place: the kernels that write the scan copies and the
// scan's own conversion of the doubles must produce the same bits (the filter's margins are derived for
// exactly these roundings, and the two paths are held to equal pair lists by the tests).
// Camera rays (cone test): the unit direction.
__device__ __forceinline__ void scan_h_camera(double vx, double vy, double vz, float* ux, float* uy, float* uz) {
    const double inv_len = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
    *ux = (float)(vx * inv_len); *uy = (float)(vy * inv_len); *uz = (float)(vz * inv_len);
}
// Rays with nothing in common (distance of a sphere's centre from the ray's line, expanded): P.V, |P|^2,
// 1 / |V|^2 from the rounded values the test multiplies.
__device__ __forceinline__ void scan_h_any(float fx, float fy, float fz, float gx, float gy, float gz,
                                           float* ux, float* uy, float* uz) {
    *ux = (float)((double)fx * gx + (double)fy * gy + (double)fz * gz);
    *uy = (float)((double)fx * fx + (double)fy * fy + (double)fz * fz);
    *uz = (float)(1.0 / ((double)gx * gx + (double)gy * gy + (double)gz * gz));
}
// Shadow ray towards a point light: distance to the light, rounded up.
__device__ __forceinline__ float scan_len_point(const double* light_pos, double px, double py, double pz) {
    const double lx = light_pos[0] - px, ly = light_pos[1] - py, lz = light_pos[2] - pz;
    return (float)(sqrt(lx * lx + ly * ly + lz * lz) * (1.0 + 1e-6));
}
// Shadow ray towards a directional light: its origin in the light's plane (frame F: e1, e2, offsets), and the
// square of that point.
__device__ __forceinline__ void scan_plane_point(const double* F, double px, double py, double pz,
                                                 float* ux, float* uy, float* uz) {
    *ux = (float)(px * F[0] + py * F[1] + pz * F[2] + F[6]);
    *uy = (float)(px * F[3] + py * F[4] + pz * F[5] + F[7]);
    *uz = (float)((double)*ux * (double)*ux + (double)*uy * (double)*uy);
}

struct LevelStore {             // one entry per hit of a level
    int32_t* parent;            // index of the ray in this level's queue (= hit index one level up, or the pixel)
    int32_t* obj;               // Scene::objects index that was hit
    double* phong;              // [3][cap] phongShade result
    double* child;              // [3][cap] value returned by the reflected ray (0 when it misses): the level-by-level fold
    int32_t* next;              // the hit of this hit's reflected ray one level down, or -1 (the same memory as `child`:
                                // the fold that walks the chains needs no stored child values)
};
#define RT_FOLD_MAX_LEVELS 8    // deepest chain k_fold_all walks (deeper scenes fold level by level)

struct HitAux {                 // per hit of the current level only
    double *nx, *ny, *nz;       // shading normal
    double *vx, *vy, *vz;       // the last object's transformed ray direction (raytracer.cpp:518-520)
};

struct PassDesc {               // rows traced by one pass, see rt_render_tiles
    int32_t y_start, tile_rows, stride_rows, q0;   // pass row q -> y = y_start + (q / tile_rows) * stride_rows + q % tile_rows
};

struct DevCounters {
    unsigned long long rays_primary, rays_reflect, rays_shadow, hits;
    unsigned long long closest_tests, anyhit_tests, exact_tests;
    unsigned long long bvh_node_tests, bvh_leaf_tests;   // written through BvhCounters (same layout)
    unsigned long long in_plane_origin_hits;
    unsigned long long flagged_rays;                     // rays that took k_brute's full scan (per launch and part 0 only)
};

// ---------------------------------------------------------------------------
// Camera::generateRay (raytracer.cpp:441-447) for the pixels of a pass, plus the
// black background of pixels whose ray hits nothing (raytracer.cpp:546).
// ---------------------------------------------------------------------------
//
// Queue order: which pixel a queue index stands for is free (every ray carries its pixel), and
// the scans bound the rays of a wave together, so neighbours in the queue should be neighbours
// on the screen in both directions.  The pass's rows are cut into bands of RT_TILE rows, a band
// into tiles of RT_TILE x RT_TILE pixels, a tile into four 8 x 8 blocks (64 consecutive queue
// indices = one ray slot of a wave).  What does not fill a tile (the band's last columns, the
// pass's last rows) stays in scanline order.
// ---------------------------------------------------------------------------
#ifndef RT_TILE
#define RT_TILE 16
#endif
__host__ __device__ __forceinline__ void queue_pixel(int p, int W, int rows, int* row, int* x) {
    const int T = RT_TILE, band = T * W, bands = rows / T;
    if (p >= bands * band) { const int r = p - bands * band; *row = bands * T + r / W; *x = r % W; return; }
    const int b = p / band, i = p % band, wide = (W / T) * T * T;
    if (i >= wide) { const int w = W % T, j = i - wide; *row = b * T + j / w; *x = (W / T) * T + j % w; return; }
    const int t = i / (T * T), j = i % (T * T), block = j >> 6, k = j & 63;
    const int bx = (block & 1) | ((block >> 1) & 2) | ((block >> 2) & 4);       // blocks in Z order
    const int by = ((block >> 1) & 1) | ((block >> 2) & 2) | ((block >> 3) & 4);
    *row = b * T + by * 8 + (k >> 3);
    *x = t * T + bx * 8 + (k & 7);
}

__global__ void __launch_bounds__(RT_BLOCK)
k_primary(const DevScene* __restrict__ sc, PassDesc pd, int n_pix, RayQueue q,
          unsigned long long* __restrict__ keys, int32_t* __restrict__ pixel_of_ray,
          uint32_t* __restrict__ frame, uint32_t* __restrict__ counts, int n_counts) {
    int W = sc->width, H = sc->height;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += gridDim.x * blockDim.x) {
        int row, x;
        queue_pixel(p, W, n_pix / W, &row, &x);
        int qrow = pd.q0 + row;
        int y = pd.y_start + (qrow / pd.tile_rows) * pd.stride_rows + qrow % pd.tile_rows;
        float u = (float)x / (float)W;
        float v = (float)y / (float)H;
        d3 ll = ld3(sc->cam_ll), lr = ld3(sc->cam_lr), ul = ld3(sc->cam_ul), ur = ld3(sc->cam_ur);
        d3 bottom = add3(scl3(ll, (double)(1 - u)), scl3(lr, (double)u));
        d3 top = add3(scl3(ul, (double)(1 - u)), scl3(ur, (double)u));
        d3 c = add3(scl3(bottom, (double)(1 - v)), scl3(top, (double)v));
        d3 eye = ld3(sc->cam_eye);
        d3 dir = sub3(c, eye);
        // (no origin is stored: every reader of a camera ray takes the eye, TraceArgs::camera_rays / HitArgs::camera_rays)
        q.dx[p] = dir.x; q.dy[p] = dir.y; q.dz[p] = dir.z;
        if (q.scan) {
            float ux, uy, uz;
            scan_h_camera(dir.x, dir.y, dir.z, &ux, &uy, &uz);
            scan_store(q, (unsigned int)p, (float)eye.x, (float)eye.y, (float)eye.z, (float)dir.x, (float)dir.y, (float)dir.z, ux, uy, uz);
        }
        keys[p] = RT_MISS_KEY;
        int pix = (H - 1 - y) * W + x;          // row 0 of the frame is the top scanline (raytracer.cpp:225)
        pixel_of_ray[p] = pix;
        frame[pix] = 0xff000000u;               // R,G,B = 0, alpha = 255
    }
    // the pass's queue lengths: this queue's, and zero for the levels below until their rays are counted
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[0] = (uint32_t)n_pix;
        for (int k = 1; k < n_counts; k++) counts[k] = 0u;
    }
}

