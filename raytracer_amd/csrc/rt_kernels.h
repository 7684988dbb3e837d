// The trace loop as a wavefront pipeline.  One "level" is one generation of
// rays (level 0 = camera rays, level k = k-th reflection).  Per level:
//   nearest hit  of every ray in the level's queue  (k_filter<0> -> k_exact_pairs<0>, k_brute<0>)
//   k_hit        compacts the rays that hit something, in the order of the rays
//                (k_hit_counts: hits per block), computes hit point,
//                normal and reflection, and writes the next level's ray queue
//   shadow rays  one per (hit, light)               (k_filter<1> -> k_exact_pairs<1>, k_brute<1>)
//   k_phong      Phong sum over the unshadowed lights
// and after the last level k_fold walks the levels deepest-first to evaluate
//   phong_0 + kr_0 * (phong_1 + kr_1 * (...))            (raytracer.cpp:542-543)
// in the reference's order.  Queue lengths live in device memory; nothing
// is read back by the host between kernels.
//
// The trace kernels scan every primitive for every ray.  Regular triangles
// go through bounding tests (can a bundle of neighbouring rays, can a lane's rays
// reach the triangle's bounding sphere at all: filter_range) and an FP32 filter (16 floats per triangle, wave-uniform, so the
// compiler fetches them with scalar loads and they never occupy vector
// registers); pairs the filter cannot rule out are staged per lane in LDS, listed
// in global memory and re-evaluated in the reference's mixed double/float
// arithmetic, so every accept/reject decision and every hit distance is the
// reference's own.
#pragma once

#include "rt_queue.h"
#include "rt_trace_args.h"
#include "rt_filter.h"
#include "rt_exact.h"
#include "rt_levels.h"
