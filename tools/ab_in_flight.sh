#!/bin/bash
# tools/in_flight_probe.py for the default library and for every variants/libraytracer_amd_<name>.so named.
set -u
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python3 tools/in_flight_probe.py ${WORLDS:-1 8} 2>/dev/null || exit 1
for v in "$@"; do
  echo "variant $v"
  RAYTRACER_AMD_LIB=$PWD/variants/libraytracer_amd_$v.so timeout -k 10 200 python3 tools/in_flight_probe.py ${WORLDS:-1 8} 2>/dev/null || exit 1
done
