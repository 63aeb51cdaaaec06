"""rust-raytracer_amd — MI355X-native trace loop behind gyng/rust-raytracer's Scene/Camera API.

The directory name carries a hyphen, so import it with
    importlib.import_module("rust-raytracer_amd")
Only the path's needs live here: csrc/ (HIP kernels + the C ABI of include/rtx.h), the ctypes
rendering of that ABI (abi.py), the scene flattener (desc.py) and the host-side mirror of the
reference's Renderer (renderer.py).
"""
from . import abi, desc  # noqa: F401
from .desc import SceneBuilder, camera, options  # noqa: F401
from .renderer import Renderer, Scene, octree_dump  # noqa: F401,E402
from . import host, sharding  # noqa: F401,E402
