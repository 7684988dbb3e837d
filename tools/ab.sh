#!/bin/bash
# tools/scan_ms.py for the default library and for every variants/libraytracer_amd_<name>.so named on the command line.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 120 python3 tools/scan_ms.py ${SCENE_ARGS:-} || exit 1
for v in "$@"; do
  RAYTRACER_AMD_LIB=$PWD/variants/libraytracer_amd_$v.so timeout -k 10 120 python3 tools/scan_ms.py ${SCENE_ARGS:-} || exit 1
done
