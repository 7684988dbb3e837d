#!/bin/bash
# Configurations 2 and 4 through the C++ driver's frame loop (steady state: the workspace exists, frames in
# flight): rays per second of 20 frames, brute force and LBVH, one frame at a time and three in flight.
cd "$GRAFT_REPO_ROOT"
R=raytracer_amd/rt_render
for k in 1 3; do
  echo "config 2 (ellipse 1920x1080 depth 5), $k in flight"
  timeout -k 10 120 $R scenes/ellipse.scene gpurun_out/c2.png -w 1920 -h 1080 -depth 5 -frames 20 -inflight $k -stats | grep render_s || exit 1
  for mode in "" "-bvh"; do
    echo "config 4 (campanile + teddy 7680x4320 depth 6) $mode, $k in flight"
    timeout -k 10 300 $R scenes/campanile_teddy.scene gpurun_out/c4.png -w 7680 -h 4320 -depth 6 -frames 6 -inflight $k $mode -stats | grep render_s || exit 1
  done
done
