#!/bin/bash
# Runs the AddressSanitizer build of the driver (make asan) over the bundled scenes
# and the driver's modes, on a machine with a GPU.  Only reports that name
# "ERROR: AddressSanitizer" count: at process exit the HIP runtime frees through
# the sanitizer after its device allocator has shut down, which trips an internal
# CHECK of the sanitizer in some runs (exit status 1, buffered stdout lost) and says
# nothing about this library.
export ASAN_OPTIONS=detect_leaks=0
out=${1:-gpurun_out}
bad=0
for sc in mixed_xf teapot campanile_teddy synthetic_small normals; do
  for opt in "" "-bvh" "-exact" "-tile 7"; do
    timeout -k 10 120 asan_build/rt_render scenes/$sc.scene $out/asan.png -w 160 -h 90 -depth 3 $opt > $out/asan_out.log 2> $out/asan_err.log
    if grep -q "ERROR: AddressSanitizer" $out/asan_err.log; then
      echo "ASAN ERROR: $sc $opt"; grep -A14 "ERROR: AddressSanitizer" $out/asan_err.log
      bad=1
    fi
  done
done
echo "asan_cli: bad=$bad"
exit $bad
