"""Registers, scratch and occupancy of the scan / traversal kernels as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  Usage: kernel_resources.py [extra hipcc flags ...]"""
import re, subprocess, sys
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
       "-Iinclude", "-Iraytracer_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kernel_resources.so",
       "raytracer_amd/csrc/rt_api.hip"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip().strip('[]')
    if not any(k in name for k in ('k_filter', 'k_bvh_trace', 'k_exact', 'k_hit', 'k_phong', 'k_fold', 'k_brute')):
        continue
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    f = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    scratch, occ, lds = f(r'ScratchSize \[bytes/lane\]'), f(r'Occupancy \[waves/SIMD\]'), f(r'LDS Size \[bytes/block\]')
    print(f"{dn[:64]:64s} VGPR {f('VGPRs'):>4} SGPR {f('SGPRs'):>4} scratch {scratch:>4} occ {occ:>2} "
          f"sgpr-spill {f('SGPRs Spill'):>3} vgpr-spill {f('VGPRs Spill'):>3} lds {lds}")
