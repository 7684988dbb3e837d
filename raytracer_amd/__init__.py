"""MI355X-native trace loop for the logicx24/RayTracer scenes.

`abi`     ctypes declarations of the two C ABIs (include/rt_abi.h, include/rt_host.h)
`scene`   host-side scene loader handle (librt_host.so)
`render`  device-side renderer handle (libraytracer_amd.so) and the multi-GPU tile driver
"""
from . import abi  # noqa: F401
from .scene import Scene, write_png  # noqa: F401
