#!/bin/bash
# Per-launch durations of k_exact_pairs<0> over camera-ray frames (depth 0: the launch's input does not depend on what the
# kernel writes), for the default library and for every variants/libraytracer_amd_<name>.so named on the command line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
one() {
  rm -rf gpurun_out/prof_x
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_x -- python3 tools/frame_loop.py 4 scenes/teapot.scene 3840 2160 0 > gpurun_out/prof_x.log 2>&1 || { tail -5 gpurun_out/prof_x.log; exit 1; }
  f=$(find gpurun_out/prof_x -name '*kernel_trace.csv' | head -1)
  echo "== $1"; python3 tools/trace_durations.py "$f" "k_exact_pairs<0>" | tr '\n' ' '; echo; python3 tools/trace_durations.py "$f" "k_exact_pairs<1>" | tr '\n' ' '; echo
}
one default || exit 1
for v in "$@"; do
  export RAYTRACER_AMD_LIB=$PWD/variants/libraytracer_amd_$v.so
  one $v || exit 1
done
