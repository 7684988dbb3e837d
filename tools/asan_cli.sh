#!/bin/bash
# Runs the AddressSanitizer build of the driver (make asan) over the bundled scenes
# and the driver's modes, on a machine with a GPU.  A run passes only when the driver
# exits with status 0 and its stderr holds no sanitizer report; anything else fails
# the sweep: 124 / 137 are reported as hangs (the time limit), other statuses as
# failures, and the stderr of every failing run is kept as asan_fail_<scene>_<n>.log
# for the stack it holds.
#
# One failure is retried rather than counted: "CHECK failed: sanitizer_allocator_device.h
# ... dev_runtime_unloaded_".  Its stack is libamdhip64's __cxa_finalize -> libhsa-runtime64
# operator delete -> the sanitizer's quarantine recycling a DEVICE chunk (one that hipFree gave
# back earlier, parked in the quarantine) after the device runtime has been unloaded.  It shows
# on the runs that free many device buffers before exit (-bvh: the build's temporaries), with
# rt_free and hipFree called before main returns, so it is the order of the sanitizer's and the
# HIP runtime's exit handlers, not a use of freed memory by this library.  The retry runs the
# same command with quarantine_size_mb=0 (chunks are recycled at hipFree time, while the runtime
# is alive): it must then pass, or the run counts as failed.
export ASAN_OPTIONS=detect_leaks=0:symbolize=1:print_stacktrace=1
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
out=${1:-gpurun_out}
mkdir -p $out
bad=0
runs=0
retried=0
for sc in mixed_xf teapot campanile_teddy synthetic_small normals; do
  n=0
  for opt in "" "-bvh" "-exact" "-tile 7" "-gpus 1 -tile 5" "-frames 5 -inflight 3" "-frames 3 -inflight 4 -bvh"; do
    n=$((n + 1)); runs=$((runs + 1))
    timeout -k 10 120 asan_build/rt_render scenes/$sc.scene $out/asan.png -w 160 -h 90 -depth 3 $opt > $out/asan_out.log 2> $out/asan_err.log
    rc=$?
    why=""
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then why="HANG (time limit, status $rc)"
    elif [ $rc -ne 0 ]; then why="FAILED (status $rc)"
    elif grep -q -e "ERROR: AddressSanitizer" -e "AddressSanitizer: CHECK failed" $out/asan_err.log; then why="SANITIZER REPORT (status 0)"
    elif ! grep -q "Image saved" $out/asan_out.log; then why="NO IMAGE (status 0, stdout lost)"
    fi
    if [ -n "$why" ] && [ $rc -eq 1 ] && grep -q "dev_runtime_unloaded_" $out/asan_err.log && ! grep -q "ERROR: AddressSanitizer" $out/asan_err.log; then
      cp $out/asan_err.log $out/asan_exit_check_${sc}_$n.log
      ASAN_OPTIONS=$ASAN_OPTIONS:quarantine_size_mb=0 timeout -k 10 120 asan_build/rt_render scenes/$sc.scene $out/asan.png -w 160 -h 90 -depth 3 $opt > $out/asan_out.log 2> $out/asan_err.log
      rc=$?
      if [ $rc -eq 0 ] && grep -q "Image saved" $out/asan_out.log && ! grep -q -e "ERROR: AddressSanitizer" -e "CHECK failed" $out/asan_err.log; then
        echo "asan_cli: $sc $opt: exit-time quarantine CHECK of the sanitizer runtime; passes with quarantine_size_mb=0"
        why=""; retried=$((retried + 1))
      else
        why="FAILED again with quarantine_size_mb=0 (status $rc)"
      fi
    fi
    if [ -n "$why" ]; then
      echo "asan_cli: $sc $opt: $why"
      cp $out/asan_err.log $out/asan_fail_${sc}_$n.log
      grep -m1 -A24 -e "ERROR: AddressSanitizer" -e "CHECK failed" $out/asan_err.log
      bad=$((bad + 1))
      # a hang may have left the GPU in no state for another run
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "asan_cli: stopping after a hang"; echo "asan_cli: runs=$runs bad=$bad"; exit 1; fi
    fi
  done
done
echo "asan_cli: runs=$runs bad=$bad retried_exit_check=$retried"
[ $bad -eq 0 ]
