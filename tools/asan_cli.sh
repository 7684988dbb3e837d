#!/bin/bash
# Runs the AddressSanitizer build of the driver (make asan) over the bundled scenes
# and the driver's modes, on a machine with a GPU.  A run passes only when the driver
# exits with status 0, has written its image and its stderr holds no sanitizer report;
# anything else fails the sweep -- there is no second try: 124 / 137 are reported as hangs
# (the time limit), other statuses as failures, and the stderr of every failing run is kept
# as asan_fail_<pass>_<scene>_<n>.log for the stack it holds.
#
# Two passes over the same 35 commands:
#   run   default sanitizer options; the driver leaves through _exit once the image is written and
#         everything it allocated has been freed (RT_RENDER_FAST_EXIT).  This pass is about the
#         program while it runs.  Without it the sanitizer runtime aborts some runs AFTER main has
#         returned: "CHECK failed: sanitizer_allocator_device.h ... dev_runtime_unloaded_" --
#         libamdhip64's __cxa_finalize -> libhsa-runtime64 operator delete -> the sanitizer's
#         quarantine recycling a device chunk after it has shut its device allocator down: the
#         order of two exit handlers neither of which is this program's.
#   exit  normal exit, every exit handler runs, with quarantine_size_mb=0 (freed chunks are
#         recycled at once, so nothing of the GPU runtime's is left parked when it unloads).
#         This pass covers the exit path of the two libraries themselves.
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
out=${1:-gpurun_out}
mkdir -p $out
bad=0
runs=0
for pass in run exit; do
  if [ $pass = run ]; then
    export ASAN_OPTIONS=detect_leaks=0:symbolize=1:print_stacktrace=1 RT_RENDER_FAST_EXIT=1
  else
    export ASAN_OPTIONS=detect_leaks=0:symbolize=1:print_stacktrace=1:quarantine_size_mb=0
    unset RT_RENDER_FAST_EXIT
  fi
  for sc in mixed_xf teapot campanile_teddy synthetic_small normals; do
    n=0
    for opt in "" "-bvh" "-exact" "-tile 7" "-gpus 1 -tile 5 -packed" "-frames 5 -inflight 3" "-frames 3 -inflight 4 -bvh"; do
      n=$((n + 1)); runs=$((runs + 1))
      rm -f $out/asan.png
      timeout -k 10 120 asan_build/rt_render scenes/$sc.scene $out/asan.png -w 160 -h 90 -depth 3 $opt > $out/asan_out.log 2> $out/asan_err.log
      rc=$?
      why=""
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then why="HANG (time limit, status $rc)"
      elif [ $rc -ne 0 ]; then why="FAILED (status $rc)"
      elif grep -q -e "ERROR: AddressSanitizer" -e "AddressSanitizer: CHECK failed" $out/asan_err.log; then why="SANITIZER REPORT (status 0)"
      elif ! grep -q "Image saved" $out/asan_out.log || [ ! -s $out/asan.png ]; then why="NO IMAGE (status 0)"
      fi
      if [ -n "$why" ]; then
        echo "asan_cli: pass $pass: $sc $opt: $why"
        cp $out/asan_err.log $out/asan_fail_${pass}_${sc}_$n.log
        grep -m1 -A24 -e "ERROR: AddressSanitizer" -e "CHECK failed" $out/asan_err.log
        bad=$((bad + 1))
        # a hang may have left the GPU in no state for another run
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "asan_cli: stopping after a hang"; echo "asan_cli: runs=$runs bad=$bad"; exit 1; fi
      fi
    done
  done
  echo "asan_cli: pass $pass done: runs=$runs bad=$bad"
done
echo "asan_cli: runs=$runs bad=$bad retried=0"
[ $bad -eq 0 ]
