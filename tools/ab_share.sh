#!/bin/bash
# tools/sim_world.py (one rank's share of an N-GPU frame, timed on one GPU) for the default library and for the named variants.
set -u
cd "$GRAFT_REPO_ROOT"
echo default; timeout -k 10 120 python3 tools/sim_world.py ${WORLDS:-1 4 8} 2>&1 | grep world || exit 1
for v in "$@"; do
  echo $v; RAYTRACER_AMD_LIB=$PWD/variants/libraytracer_amd_$v.so timeout -k 10 120 python3 tools/sim_world.py ${WORLDS:-1 4 8} 2>&1 | grep world || exit 1
done
