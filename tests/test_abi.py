"""The two shared libraries load and export exactly what include/*.h declares.
No compute call is made here (there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z_]+)\s*\(", text)))


def test_header_lists_match_python_lists():
    assert declared_functions("rt_abi.h") == sorted(abi.DEVICE_SYMBOLS)
    assert declared_functions("rt_host.h") == sorted(abi.HOST_SYMBOLS)


def test_host_library_exports():
    lib = C.CDLL(abi.host_lib_path())
    for name in declared_functions("rt_host.h"):
        assert hasattr(lib, name), name


def test_device_library_exports():
    assert os.path.exists(abi.device_lib_path()), "run `make hip` (or __graft_entry__.build())"
    lib = C.CDLL(abi.device_lib_path())
    for name in declared_functions("rt_abi.h"):
        assert hasattr(lib, name), name


def test_struct_sizes_match_c():
    # sizes computed from the C declarations: doubles and int32 only, natural alignment
    assert C.sizeof(abi.RtMaterial) == 13 * 8
    assert C.sizeof(abi.RtTransform) == 48 * 8
    assert C.sizeof(abi.RtLight) == 6 * 8 + 8
    assert C.sizeof(abi.RtSceneDesc) == 24 + 18 * 8 + 16 + 9 * 8
    assert C.sizeof(abi.RtStats) == 11 * 8 + 3 * 8 + 3 * 4 + 5 * 4 + 8 + 2 * 4 + 6 * 11 * 8 + 6 * 8 + 6 * 4 + 4 * 4


def test_struct_sizes_match_the_compiler(tmp_path):
    """... and what the C compiler makes of include/rt_abi.h itself."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rt_abi.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rt_material), '
                   'sizeof(rt_transform), sizeof(rt_light), sizeof(rt_scene_desc), sizeof(rt_stats), sizeof(rt_scan_work)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(t) for t in (abi.RtMaterial, abi.RtTransform, abi.RtLight, abi.RtSceneDesc, abi.RtStats, abi.RtScanWork)]
    assert got == want


def test_upload_without_gpu_fails_loudly():
    """Without a HIP device rt_upload_scene must fail with RT_ERR_DEVICE, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracer_amd import Scene
    from raytracer_amd.render import Renderer, DeviceError
    sc = Scene.load(os.path.join(ROOT, "scenes", "inputs", "input-01"), 8, 8)
    with pytest.raises(DeviceError) as e:
        Renderer(sc)
    assert "-2" in str(e.value)
