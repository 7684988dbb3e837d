// Optional LBVH over the regular triangles of each transform group.
//
// The hierarchy is one more CONSERVATIVE filter in front of the reference-precision test:
// traversal collects the triangles whose padded box the ray's half-line can reach, runs the
// same FP32 triangle filter as the brute-force scan on them, and gives the survivors to the
// same reference-precision test (tri_intersect_exact).  It skips a box only when the ray
// cannot reach it or when everything in it lies behind a hit that test has already accepted
// (see k_bvh_trace), so the nearest hit (index and distance bits) is the brute-force one.  The one reference behaviour a box hierarchy cannot reproduce is the
// in-plane-origin case of Triangle::intersection (raytracer.cpp:319-321: a == 0 with a
// grazing ray "hits" the infinite plane, wherever the triangle is); rt_stats counts those
// events in brute-force mode so a caller can tell whether a scene has any.
//
// Build (GPU): Morton codes of the centroids -> rocPRIM radix sort -> Karras' parallel
// hierarchy (2012) on (code, index) keys -> bottom-up box refit with one atomic counter per
// internal node.  Every ray that cannot be traversed safely (stack overflow, origin out of
// the filter's bounds) is flagged and takes k_brute's full scan instead.
#pragma once

#include "rt_kernels.h"

#define RT_BVH_BLOCK 256
#ifndef RT_BVH_STACK
#define RT_BVH_STACK 24         // per-lane traversal stack entries kept in LDS
#endif
#define RT_BVH_SPILL 48         // ... and deeper ones in a per-thread strip of global memory
#ifndef RT_BVH_STAGE
#define RT_BVH_STAGE 4          // per-lane staged candidates (LDS)
#endif
#ifndef RT_BVH_STEPS
#define RT_BVH_STEPS 8          // traversal steps between two drain phases
#endif
#ifndef RT_BVH_ROTATION_PASSES
#define RT_BVH_ROTATION_PASSES 3   // passes of tree rotations after the build (see k_bvh_rotate): box tests per ray 30.4 -> 20.6 -> 18.6 -> 18.5 with 0, 1, 2, 4 passes on the teapot
#endif
#define RT_BVH_NO_CHILD (-2147483647 - 1)   // child code of a leaf whose slot holds no triangle (TriExact::hole): never entered

struct __attribute__((aligned(16))) BvhNode {
    float lo[2][3], hi[2][3];   // boxes of the two children
    int32_t child[2];           // >= 0: internal node (index into the group's node array); < 0: triangle slot -1 - child
    int32_t pad[2];
};

struct GroupBvh {
    unsigned long long node_offset;   // the group's n - 1 internal nodes start here in the node array; root first
    int32_t n;                  // triangles in the group
    int32_t first_slot;         // n == 1: the only triangle's slot
};

// ---- build ----------------------------------------------------------------

__device__ __forceinline__ unsigned int expand_bits10(unsigned int v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 30-bit Morton code of each triangle's centroid inside the group's centroid box.
__global__ void k_bvh_morton(const TriExact* __restrict__ exact, int base, int n, float3 cmin, float3 cscale,
                             unsigned int* __restrict__ keys, int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TriExact& T = exact[base + i];
    float c[3];
    for (int k = 0; k < 3; k++) c[k] = (float)(T.v1[k] + (T.e1[k] + T.e2[k]) * (1.0 / 3.0));
    float q[3] = {(c[0] - cmin.x) * cscale.x, (c[1] - cmin.y) * cscale.y, (c[2] - cmin.z) * cscale.z};
    unsigned int code = 0;
    for (int k = 0; k < 3; k++) {
        float v = fminf(fmaxf(q[k] * 1024.0f, 0.0f), 1023.0f);
        code |= expand_bits10((unsigned int)v) << (2 - k);
    }
    keys[i] = code;
    vals[i] = i;
}

// Common-prefix length of sorted keys i and j; equal codes are told apart by their position.
__device__ __forceinline__ int bvh_delta(const unsigned int* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned int a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((unsigned int)(i ^ j));
    return __clz(a ^ b);
}

// Karras 2012: internal node i of n - 1; leaves are sorted positions 0 .. n-1.
__global__ void k_bvh_hierarchy(const unsigned int* __restrict__ keys, const int* __restrict__ vals, int n, int base,
                                BvhNode* __restrict__ nodes, int* __restrict__ parent_of_node,
                                int* __restrict__ parent_of_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = bvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = bvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    // child codes: leaves carry the triangle's slot
    int left, right;
    if (lo == gamma) { left = -1 - (base + vals[gamma]); parent_of_leaf[gamma] = 2 * i; }
    else { left = gamma; parent_of_node[gamma] = 2 * i; }
    if (hi == gamma + 1) { right = -1 - (base + vals[gamma + 1]); parent_of_leaf[gamma + 1] = 2 * i + 1; }
    else { right = gamma + 1; parent_of_node[gamma + 1] = 2 * i + 1; }
    nodes[i].child[0] = left; nodes[i].child[1] = right;
    nodes[i].pad[0] = nodes[i].pad[1] = 0;
    if (i == 0) parent_of_node[0] = -1;
}

__device__ __forceinline__ float round_down(double v) {
    float f = (float)v;
    return ((double)f > v) ? nextafterf(f, -INFINITY) : f;
}
__device__ __forceinline__ float round_up(double v) {
    float f = (float)v;
    return ((double)f < v) ? nextafterf(f, INFINITY) : f;
}

// Leaf boxes (padded) and the bottom-up union.  The second thread to arrive at a node
// continues upwards; the boxes it reads were published by the first behind a device-scope fence.
__global__ void k_bvh_refit(const TriExact* __restrict__ exact, const int* __restrict__ vals, int n, int base,
                            BvhNode* nodes, const int* __restrict__ parent_of_node, const int* __restrict__ parent_of_leaf,
                            unsigned int* arrive, float pad_abs) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const TriExact& T = exact[base + vals[k]];
    float lo[3], hi[3];
    const float pad = pad_abs + T.box_pad;
    for (int a = 0; a < 3; a++) {
        double p0 = T.v1[a], p1 = T.v1[a] + T.e1[a], p2 = T.v1[a] + T.e2[a];
        lo[a] = round_down(fmin(p0, fmin(p1, p2))) - pad;
        hi[a] = round_up(fmax(p0, fmax(p1, p2))) + pad;
        // a hole in the slot range (see TriExact::hole) adds nothing to the boxes above it
        if (T.hole) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
    }
    int link = parent_of_leaf[k];
    if (T.hole && link >= 0) nodes[link >> 1].child[link & 1] = RT_BVH_NO_CHILD;     // (k_bvh_hierarchy has finished: a launch earlier)
    while (link >= 0) {
        const int node = link >> 1, side = link & 1;
        for (int a = 0; a < 3; a++) {
            __hip_atomic_store(&nodes[node].lo[side][a], lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodes[node].hi[side][a], hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        if (atomicAdd(&arrive[node], 1u) == 0u) return;     // the sibling subtree is not finished yet
        __threadfence();
        for (int a = 0; a < 3; a++) {
            float olo = __hip_atomic_load(&nodes[node].lo[side ^ 1][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float ohi = __hip_atomic_load(&nodes[node].hi[side ^ 1][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo[a] = fminf(lo[a], olo);
            hi[a] = fmaxf(hi[a], ohi);
        }
        link = parent_of_node[node];
    }
}

// Tree rotations (Kensler 2008) on the finished binary tree, one bottom-up pass per launch: at every node
// X = {A, B} whose two subtrees are complete, the four exchanges of a child with a grandchild on the other side
// (B <-> A0, B <-> A1, A <-> B0, A <-> B1) are priced by the surface area of the one node they rebuild, and the
// best one is applied when it shrinks that node.  The Morton order splits space by bit pattern, not by where
// the surfaces are; rotations repair the worst of it (the same leaves under every node above X, so no box
// above X changes).  Climbing is k_bvh_refit's: the second thread to arrive at a node owns it.
__device__ __forceinline__ float box_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ float union_area(const float* lo0, const float* hi0, const float* lo1, const float* hi1) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = fminf(lo0[a], lo1[a]); hi[a] = fmaxf(hi0[a], hi1[a]); }
    return box_area(lo, hi);
}
__device__ __forceinline__ void load_node(const BvhNode* nodes, int i, BvhNode* out) {
    for (int s = 0; s < 2; s++) {
        for (int a = 0; a < 3; a++) {
            out->lo[s][a] = __hip_atomic_load(&nodes[i].lo[s][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out->hi[s][a] = __hip_atomic_load(&nodes[i].hi[s][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out->child[s] = __hip_atomic_load(&nodes[i].child[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void store_side(BvhNode* nodes, int i, int side, const float* lo, const float* hi, int child) {
    for (int a = 0; a < 3; a++) {
        __hip_atomic_store(&nodes[i].lo[side][a], lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&nodes[i].hi[side][a], hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(&nodes[i].child[side], child, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rotate_at(BvhNode* nodes, int x, int n_internal, unsigned int* n_rotations) {
    BvhNode X;
    load_node(nodes, x, &X);
    if (X.child[0] == RT_BVH_NO_CHILD || X.child[1] == RT_BVH_NO_CHILD) return;
    float best = 0.0f;
    int pick = -1;
    BvhNode C[2];
    bool inner[2];
    for (int s = 0; s < 2; s++) {
        // side s is rebuilt: its node keeps one of its children and takes X's other child
        const int c = X.child[s];
        inner[s] = c >= 0 && c < n_internal;
        if (!inner[s]) continue;
        load_node(nodes, c, &C[s]);
        if (C[s].child[0] == RT_BVH_NO_CHILD || C[s].child[1] == RT_BVH_NO_CHILD) { inner[s] = false; continue; }
        const float now = box_area(X.lo[s], X.hi[s]);
        for (int keep = 0; keep < 2; keep++) {
            const float then = union_area(C[s].lo[keep], C[s].hi[keep], X.lo[s ^ 1], X.hi[s ^ 1]);
            const float gain = now - then;
            if (gain > best && gain > 1e-4f * now) { best = gain; pick = 2 * s + keep; }
        }
    }
    if (pick < 0) return;
    const int s = pick >> 1, keep = pick & 1, go = keep ^ 1;     // C[s] keeps child `keep`, gives child `go` to X, takes X's other child
    const int c = X.child[s];
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = fminf(C[s].lo[keep][a], X.lo[s ^ 1][a]);
        hi[a] = fmaxf(C[s].hi[keep][a], X.hi[s ^ 1][a]);
    }
    // the rebuilt node: slot `go` now holds X's other child
    store_side(nodes, c, go, X.lo[s ^ 1], X.hi[s ^ 1], X.child[s ^ 1]);
    // X: side s keeps the rebuilt node with its new box, the other side takes the grandchild
    store_side(nodes, x, s, lo, hi, c);
    store_side(nodes, x, s ^ 1, C[s].lo[go], C[s].hi[go], C[s].child[go]);
    if (n_rotations) atomicAdd(n_rotations, 1u);
}

__global__ void k_bvh_rotate(int n, BvhNode* nodes, const int* __restrict__ parent_of_node, const int* __restrict__ parent_of_leaf,
                             unsigned int* arrive, unsigned int* n_rotations) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int link = parent_of_leaf[k];
    while (link >= 0) {
        const int node = link >> 1;
        __threadfence();
        if (atomicAdd(&arrive[node], 1u) == 0u) return;     // the sibling subtree is not finished yet
        __threadfence();
        rotate_at(nodes, node, n - 1, n_rotations);
        link = parent_of_node[node];
    }
}

// Parent links from the child codes (after a pass of rotations; the links of hole leaves, which never move, stay).
// leaf_of_slot: sorted position of the leaf that holds slot base + j.
__global__ void k_bvh_links(int n, int base, const BvhNode* __restrict__ nodes, const int* __restrict__ leaf_of_slot,
                            int* __restrict__ parent_of_node, int* __restrict__ parent_of_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    for (int s = 0; s < 2; s++) {
        const int c = nodes[i].child[s];
        if (c == RT_BVH_NO_CHILD) continue;
        if (c >= 0) parent_of_node[c] = 2 * i + s;
        else parent_of_leaf[leaf_of_slot[(-1 - c) - base]] = 2 * i + s;
    }
    if (i == 0) parent_of_node[0] = -1;
}
__global__ void k_bvh_leaf_of_slot(int n, const int* __restrict__ vals, int* __restrict__ leaf_of_slot) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) leaf_of_slot[vals[k]] = k;
}

// Four-wide nodes (what RT_OPT_USE_BVH = 1 traverses; 2 walks the binary tree): node i stands for binary node i with each internal child replaced by
// that child's two children -- up to four boxes per step, half as many dependent node fetches per ray.  Every
// binary node gets its wide form (one thread each, no compaction); a traversal that starts at the root only
// ever visits those that stand for every second level.  Boxes and child codes are the binary tree's own.
struct __attribute__((aligned(16))) BvhNode4 {
    float lo[4][3], hi[4][3];
    int32_t child[4];           // as BvhNode::child; RT_BVH_NO_CHILD: no such child
    int32_t pad[4];
};

__global__ void k_bvh_widen(const BvhNode* __restrict__ nodes, int n_internal, BvhNode4* __restrict__ wide) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    BvhNode4 W;
    int k = 0;
    const BvhNode N = nodes[i];
    for (int side = 0; side < 2; side++) {
        const int c = N.child[side];
        if (c == RT_BVH_NO_CHILD) continue;
        if (c < 0 || c >= n_internal) {                         // a leaf (or a code this kernel does not follow): the child itself
            for (int a = 0; a < 3; a++) { W.lo[k][a] = N.lo[side][a]; W.hi[k][a] = N.hi[side][a]; }
            W.child[k++] = c;
        } else {
            const BvhNode M = nodes[c];
            for (int s2 = 0; s2 < 2; s2++) {
                if (M.child[s2] == RT_BVH_NO_CHILD) continue;
                for (int a = 0; a < 3; a++) { W.lo[k][a] = M.lo[s2][a]; W.hi[k][a] = M.hi[s2][a]; }
                W.child[k++] = M.child[s2];
            }
        }
    }
    for (; k < 4; k++) {
        for (int a = 0; a < 3; a++) { W.lo[k][a] = 3.0e38f; W.hi[k][a] = -3.0e38f; }
        W.child[k] = RT_BVH_NO_CHILD;
    }
    for (int q = 0; q < 4; q++) W.pad[q] = 0;
    wide[i] = W;
}

// ---- traversal --------------------------------------------------------------

struct BvhCounters { unsigned long long node_tests, leaf_tests; };

// The brute-force filter's test for ONE ray and a triangle record fetched per lane (same
// arithmetic as filter_z2; `robust` selects the per-test margin f[13 + mode]).
// *r_lo_out: a lower bound of the ray parameter the reference would compute for the pair.
__device__ __forceinline__ bool filter_keep1(const float* __restrict__ f, int axis, float px, float py, float pz,
                                             float vx, float vy, float vz, const GroupFilter& Q, const bool robust,
                                             const int mode, float* r_lo_out) {
    float b = f[0] * vx;
    b = __builtin_fmaf(f[1], vy, b);
    b = __builtin_fmaf(f[2], vz, b);
    float a = __builtin_fmaf(-f[0], px, f[3]);
    a = __builtin_fmaf(-f[1], py, a);
    a = __builtin_fmaf(-f[2], pz, a);
    const float rinv = __builtin_amdgcn_rcpf(b);
    const float r = a * rinv;
    const float pu = axis == 0 ? py : (axis == 1 ? pz : px), pv = axis == 0 ? pz : (axis == 1 ? px : py);
    const float vu = axis == 0 ? vy : (axis == 1 ? vz : vx), vv = axis == 0 ? vz : (axis == 1 ? vx : vy);
    const float hu = __builtin_fmaf(r, vu, pu), hv = __builtin_fmaf(r, vv, pv);
    const float s = __builtin_fmaf(f[4], hu, __builtin_fmaf(f[5], hv, f[6]));
    const float t = __builtin_fmaf(f[7], hu, __builtin_fmaf(f[8], hv, f[9]));
    const float u = __builtin_fmaf(f[10], hu, __builtin_fmaf(f[11], hv, f[12]));
    float m = __builtin_fminf(__builtin_fminf(s, t), u);
    if (robust) m = __builtin_fmaf(f[13 + mode], fabsf(rinv), m);
    const bool reject = (m < 0.0f) | (r < Q.r_lo);
    // (the in-plane-origin branch is the reference's only for a grazing ray, raytracer.cpp:319: both forms ask for one)
    const bool in_plane = (fabsf(a) <= Q.a_zero) & (fabsf(b) <= Q.b_hi);
    // |r - reference's r| <= (Ea + |r| Eb) / |b| + rounding of the division and the product
    *r_lo_out = r - ((Q.e_a + fabsf(r) * Q.e_b) * fabsf(rinv) * 1.25f + fabsf(r) * 1e-6f);
    return (!reject) | in_plane;
}

// Where the half-line P + tV, t >= 0, is inside the (already padded) box: false when it never
// is, else *t_near = a lower bound of the entry parameter.  NaNs (0 * inf on a slab boundary)
// drop out of min/max, which only widens the interval.
__device__ __forceinline__ bool ray_reaches_box(const float* lo, const float* hi, float px, float py, float pz,
                                                float ix, float iy, float iz, float* t_near) {
    const float x0 = (lo[0] - px) * ix, x1 = (hi[0] - px) * ix;
    const float y0 = (lo[1] - py) * iy, y1 = (hi[1] - py) * iy;
    const float z0 = (lo[2] - pz) * iz, z1 = (hi[2] - pz) * iz;
    float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    // widen by rounding slack; by multiplication, so that an infinite bound (zero direction
    // component, box entirely to one side) stays infinite instead of turning into inf - inf
    tnear *= (tnear > 0.0f) ? (1.0f - 1e-5f) : (1.0f + 1e-5f);
    tfar *= (tfar > 0.0f) ? (1.0f + 1e-5f) : (1.0f - 1e-5f);
    *t_near = tnear;
    return !(tnear > tfar) && !(tfar < 0.0f);
}

// One ray per lane, nodes visited nearer child first.  The wave alternates between a few
// traversal steps and a drain phase in which every lane that has staged triangles (survivors
// of the FP32 filter) runs the reference-precision test on them -- together, so that the
// double-precision work is not serialised lane by lane.  What the exact test finds bounds
// the rest of the search:
//   nearest hit, identity-transform group: the metric of Scene::traceRay (raytracer.cpp:522-524)
//     is |dir| * r there, monotone in the ray parameter r, so boxes and triangles that begin
//     behind an accepted hit can neither hold the nearest hit nor tie with it;
//   nearest hit, transformed group: the metric mixes spaces; nothing is pruned;
//   shadow ray: the first accepted hit off the source object settles it (raytracer.cpp:553).
// Results are merged exactly as k_exact_pairs does (atomicMin of the key / atomicOr of the
// light's bit), so the other kernels of the launch need not know which path produced them.
// Every array the traversal reads is a kernel argument (not a pointer fetched from the scene
// header), so that the loads are global-memory loads rather than generic ones.
struct BvhArgs {
    const GroupBvh* bvh;            // per group
    const BvhNode* nodes;           // all groups' nodes; GroupBvh::node_offset indexes into it
    const BvhNode4* wide;           // the same nodes four-wide (WIDE instances)
    const float* filt;              // FP32 filter records, 16 floats per regular triangle
    const TriExact* exact;
    const Group* groups;
    const double* xforms;
    int* spill;                     // deep stack entries: [RT_BVH_SPILL][threads of the launch]
    BvhCounters* counters;
    // Per block of the launch, 4 entries: box tests, leaf tests, reference-precision tests executed (and one
    // spare), added up by the host when statistics are read.  (One set of counters for the whole launch had
    // every wave end with three atomics on one 128-byte line: 24,000 per launch, one after the other.)
    unsigned long long* tally;
};

#ifndef RT_BVH_MIN_WAVES
#define RT_BVH_MIN_WAVES 1      // waves per SIMD the traversal is compiled for (a higher value caps its registers)
#endif
template <int MODE, bool WIDE = false>
__global__ void __launch_bounds__(RT_BVH_BLOCK, RT_BVH_MIN_WAVES)
k_bvh_trace(const DevScene* __restrict__ sc, BvhArgs B, TraceArgs A) {
    const GroupBvh* __restrict__ bvh = B.bvh;
    const float* __restrict__ filt = B.filt;
    BvhCounters* counters = B.counters;
    int* __restrict__ spill = B.spill;
    __shared__ int stack[RT_BVH_STACK * RT_BVH_BLOCK];
    __shared__ unsigned int stage[RT_BVH_STAGE * RT_BVH_BLOCK];
    int* my_stack = stack + threadIdx.x;                 // entry k at my_stack[k * RT_BVH_BLOCK]
    // entries RT_BVH_STACK and deeper: entry k at my_spill[(k - RT_BVH_STACK) * n_threads]
    const size_t n_threads = (size_t)gridDim.x * RT_BVH_BLOCK;
    int* my_spill = spill + (size_t)blockIdx.x * RT_BVH_BLOCK + threadIdx.x;
    auto stack_get = [&](int k) -> int {
        return k < RT_BVH_STACK ? my_stack[k * RT_BVH_BLOCK] : my_spill[(size_t)(k - RT_BVH_STACK) * n_threads];
    };
    auto stack_put = [&](int k, int v) {
        if (k < RT_BVH_STACK) my_stack[k * RT_BVH_BLOCK] = v;
        else my_spill[(size_t)(k - RT_BVH_STACK) * n_threads] = v;
    };
    unsigned int* my_stage = stage + threadIdx.x;
    const int lane = threadIdx.x & (RT_WAVE - 1);
    const unsigned int n_lights = sc->n_lights;
    const unsigned int n_hits = __builtin_amdgcn_readfirstlane(*A.n_items);
    const unsigned int n_rays = (MODE == 0) ? n_hits : n_hits * n_lights;
    const bool robust = sc->robust_filter != 0;
    const int n_groups = sc->n_groups;
    const float eps = (MODE == 0) ? sc->eps : sc->shadow_eps;
    const TriExact* __restrict__ exact = B.exact;
    unsigned long long n_node = 0, n_leaf = 0;
    unsigned int n_exact = 0;

    const unsigned int stride = gridDim.x * blockDim.x;
    const unsigned int n_round = (n_rays + stride - 1) / stride * stride;
    for (unsigned int ray = blockIdx.x * blockDim.x + threadIdx.x; ray < n_round; ray += stride) {
        const bool valid = ray < n_rays;
        d3 wpos = mk3(0, 0, 0), wvec = mk3(0, 0, 1);
        int source = -1;
        // per-ray state
        bool active = valid;                            // still has boxes to visit or triangles staged
        if (MODE == 1 && A.spheres == 2 && valid) {     // (a sphere is in the way already: k_spheres_first)
            unsigned int hit, light;
            shadow_ray_index(ray, n_hits, &hit, &light);
            active = ((A.occluded[(size_t)hit * A.occl_words + (light >> 5)] >> (light & 31)) & 1u) == 0u;
        }
        if (active) load_world_ray<MODE>(A, sc, ray, n_hits, &wpos, &wvec, &source);
        bool bail = false;                              // this ray takes the full reference-precision scan instead
        bool need_group = true;                         // the current group is finished (or none was entered yet)
        int g = -1, cur = 0, sp = 0, n_staged = 0;
        unsigned long long best = RT_MISS_KEY;          // MODE 0: smallest key found here
        float t_limit_identity = INFINITY;              // identity groups share the ray's coordinates
        // per-group context
        d3 pos = wpos, vec = wvec;
        float px = 0, py = 0, pz = 0, vx = 0, vy = 0, vz = 1, ix = 0, iy = 0, iz = 1, t_limit = INFINITY;
        const Group* G = B.groups;
        const BvhNode* nodes = nullptr;
        const BvhNode4* wide = nullptr;
        int n_internal = 0;
        bool prune = false;

        unsigned int rounds = 0;
        while (__any(active)) {
            // (a sound tree is walked in far fewer rounds; one that is not must not hold the wave for ever: the full scan instead)
            if (++rounds > (1u << 20)) { if (active) { bail = true; active = false; n_staged = 0; } break; }
#pragma unroll 1
            for (int step = 0; step < RT_BVH_STEPS; step++) {
                if (!active) break;
#ifdef RT_BVH_UTIL
                // diagnostic build: lane-steps worked (rt_stats.bvh_node_tests) against the lane-steps the wave spent on them
                // (rt_stats.bvh_leaf_tests: 64 per step that any lane takes)
                {
                    const unsigned long long here = __ballot(true);
                    n_node++;
                    if (lane == (int)__builtin_ctzll(here)) n_leaf += 64;
                    // ... and the steps in which some lanes stand at a leaf and others at a node (rt_stats.exact_tests, x 64)
                    const unsigned long long at_leaf = __ballot(!need_group && cur < 0), at_node = __ballot(!need_group && cur >= 0);
                    if (lane == (int)__builtin_ctzll(here) && at_leaf != 0ull && at_node != 0ull) n_exact += 64;
                }
#endif
                if (need_group) {
                    if (n_staged > 0) break;            // the staged triangles belong to the group just left: drain first
                    if (prune) t_limit_identity = t_limit;
                    int n = 0;
                    do { g++; n = (g < n_groups) ? bvh[g].n : 0; } while (g < n_groups && n <= 0);
                    if (g >= n_groups) { active = false; break; }
                    G = B.groups + g;
                    const GroupFilter& Q = G->q[MODE];
                    prune = (MODE == 0) && G->identity != 0;
                    pos = wpos; vec = wvec;
                    if (MODE == 0) to_object_space(B.xforms + 48 * (size_t)G->xf, G->identity, &pos, &vec);
                    px = (float)pos.x; py = (float)pos.y; pz = (float)pos.z;
                    vx = (float)vec.x; vy = (float)vec.y; vz = (float)vec.z;
                    const float ob = Q.origin_bound;
                    const bool trusted = fabsf(px) <= ob && fabsf(py) <= ob && fabsf(pz) <= ob &&
                                         (vx * vx + vy * vy + vz * vz) <= Q.dir_len2_bound;
                    if (!trusted) { bail = true; active = false; break; }
                    ix = 1.0f / vx; iy = 1.0f / vy; iz = 1.0f / vz;
                    t_limit = prune ? t_limit_identity : INFINITY;
                    nodes = B.nodes + bvh[g].node_offset;
                    if (WIDE) { wide = B.wide + bvh[g].node_offset; n_internal = n - 1; }
                    cur = (n == 1) ? (-1 - bvh[g].first_slot) : 0;
                    sp = 0;
                    need_group = false;
                    continue;
                }
                if (cur < 0) {
                    // leaf: the FP32 triangle filter; survivors wait in the stage for the drain phase
                    if (n_staged == RT_BVH_STAGE) break;
                    const int slot = -1 - cur;
                    const int axis = slot >= G->fax[2] ? 2 : (slot >= G->fax[1] ? 1 : 0);     // (fax[k]: where axis k's slots begin)
#ifndef RT_BVH_UTIL
                    n_leaf++;
#endif
                    float r_lo;
                    if (filter_keep1(filt + 16 * (size_t)slot, axis, px, py, pz, vx, vy, vz, G->q[MODE], robust, MODE, &r_lo) &&
                        !(r_lo > t_limit)) {
                        my_stage[n_staged * RT_BVH_BLOCK] = (unsigned int)slot;
                        n_staged++;
                    }
                    if (sp == 0) need_group = true;
                    else cur = stack_get(--sp);
                    continue;
                }
                if (WIDE) {
                    // (a child code outside the group's nodes cannot come from a sound tree: the ray takes the full scan)
                    if (cur >= n_internal) { bail = true; active = false; break; }
                    const BvhNode4& N4 = wide[cur];
                    float d[4];
                    int c[4];
                    int n_hit = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float tn;
                        const bool r = ray_reaches_box(N4.lo[k], N4.hi[k], px, py, pz, ix, iy, iz, &tn);
                        c[k] = N4.child[k];
                        const bool h = r & !(tn > t_limit) & (c[k] != RT_BVH_NO_CHILD);
                        d[k] = h ? tn : INFINITY;
                        n_hit += h ? 1 : 0;
#ifndef RT_BVH_UTIL
                        n_node += (c[k] != RT_BVH_NO_CHILD) ? 1 : 0;
#endif
                    }
                    // nearest first: a five-exchange sorting network on (entry parameter, child); boxes not reached sort last
#define RT_BVH_CSWAP(i, j) { const bool sw = d[j] < d[i]; const float td = sw ? d[j] : d[i]; d[j] = sw ? d[i] : d[j]; d[i] = td; \
                             const int tc = sw ? c[j] : c[i]; c[j] = sw ? c[i] : c[j]; c[i] = tc; }
                    RT_BVH_CSWAP(0, 1) RT_BVH_CSWAP(2, 3) RT_BVH_CSWAP(0, 2) RT_BVH_CSWAP(1, 3) RT_BVH_CSWAP(1, 2)
#undef RT_BVH_CSWAP
                    if (n_hit == 0) {
                        if (sp == 0) need_group = true;
                        else cur = stack_get(--sp);
                    } else {
                        if (sp + 3 > RT_BVH_STACK + RT_BVH_SPILL) { bail = true; active = false; break; }
                        if (n_hit > 3) stack_put(sp++, c[3]);
                        if (n_hit > 2) stack_put(sp++, c[2]);
                        if (n_hit > 1) stack_put(sp++, c[1]);
                        cur = c[0];
                    }
                    continue;
                }
                const BvhNode& N = nodes[cur];
                n_node += 2;
                float tn0, tn1;
                // (no short-circuit: both boxes are tested whatever the answers, without a branch between them)
                const bool r0 = ray_reaches_box(N.lo[0], N.hi[0], px, py, pz, ix, iy, iz, &tn0);
                const bool r1 = ray_reaches_box(N.lo[1], N.hi[1], px, py, pz, ix, iy, iz, &tn1);
                const bool h0 = r0 & !(tn0 > t_limit) & (N.child[0] != RT_BVH_NO_CHILD);
                const bool h1 = r1 & !(tn1 > t_limit) & (N.child[1] != RT_BVH_NO_CHILD);
                if (h0 && h1) {
                    const int near = tn1 < tn0 ? 1 : 0;
                    if (sp == RT_BVH_STACK + RT_BVH_SPILL) { bail = true; active = false; break; }
                    stack_put(sp++, N.child[near ^ 1]);
                    cur = N.child[near];
                } else if (h0) {
                    cur = N.child[0];
                } else if (h1) {
                    cur = N.child[1];
                } else if (sp == 0) {
                    need_group = true;
                } else {
                    cur = stack_get(--sp);
                }
            }
            // drain: reference-precision tests of the staged triangles, all lanes together
            while (__any(active && n_staged > 0)) {
                if (active && n_staged > 0) {
                    const int slot = (int)my_stage[(--n_staged) * RT_BVH_BLOCK];
                    const TriExact& T = exact[slot];
                    const float t = tri_intersect_exact(T, pos, vec, eps);
#ifndef RT_BVH_UTIL
                    n_exact++;
#endif
                    if (MODE == 0) {
                        d3 world;
                        const float d = hit_metric(B.xforms + 48 * (size_t)G->xf, G->identity, pos, vec, t, &world);
                        if (t != RT_MISS_T && d < RT_MISS_T) {
                            const unsigned long long key = make_key(d, T.obj);
                            if (key < best) best = key;
                            if (prune) t_limit = fminf(t_limit, t + fabsf(t) * 1e-4f);
                        }
                    } else if (t != RT_MISS_T && T.obj != source) {
                        unsigned int hit, light;
                        shadow_ray_index(ray, n_hits, &hit, &light);
                        atomicOr(&A.occluded[(size_t)hit * A.occl_words + (light >> 5)], 1u << (light & 31));
                        active = false;                 // occluded: nothing further to find
                    }
                }
            }
        }
        if (valid && bail) { A.ray_flags[ray] = 1; *A.any_flagged = 1u; }
        if (MODE == 0 && valid && best != RT_MISS_KEY) atomicMin(&A.keys[ray], best);
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_node += __shfl_xor(n_node, off); n_leaf += __shfl_xor(n_leaf, off); n_exact += __shfl_xor(n_exact, off);
    }
    // (every thread of the block gets here: the ray loop's trip count is the same for all of them)
    __shared__ unsigned long long block_tally[3];
    if (threadIdx.x < 3) block_tally[threadIdx.x] = 0ull;
    __syncthreads();
    if (lane == 0) {
        if (n_node) atomicAdd(&block_tally[0], n_node);
        if (n_leaf) atomicAdd(&block_tally[1], n_leaf);
        if (n_exact) atomicAdd(&block_tally[2], (unsigned long long)n_exact);
    }
    __syncthreads();
    // (launches of one kind of query follow each other on one stream, and each kind has its own tally: the
    // block's entries are its own)
    if (threadIdx.x < 3 && block_tally[threadIdx.x] != 0ull) B.tally[4 * (size_t)blockIdx.x + threadIdx.x] += block_tally[threadIdx.x];
    (void)counters;
}
