for v in 0 1 0 1; do
  export RAYTRACER_AMD_LEAN_SCANS=$v
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-config4 --no-config5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=l['roofline']['default_scans']; print('lean=$v', round(l['ms_per_step'],3), round(l['one_frame_at_a_time_ms'],3), 'scans cam/refl/shadow', round(d['camera_rays']['ms_per_step'],3), round(d['reflection_rays']['ms_per_step'],3), round(d['shadow_rays']['ms_per_step'],3), 'lbvh', round(l['lbvh']['ms_per_step'],3))" || exit 1
done
