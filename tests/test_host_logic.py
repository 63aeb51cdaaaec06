"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/rtx.h declares, and the octree it builds (an independent implementation) is identical,
bit for bit, to the oracle's reference-order octree."""
import ctypes
import os
import re

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rtx):
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", hdr))
    assert declared == set(rtx.abi.EXPORTS), declared ^ set(rtx.abi.EXPORTS)
    lib = ctypes.CDLL(rtx.abi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert rtx.abi.load().rtx_version().decode().startswith("rtx ")
    assert rtx.abi.load().rtx_strerror(2) == b"no HIP device"


def test_struct_sizes_match_header(rtx):
    a = rtx.abi
    assert ctypes.sizeof(a.Material) == 8 + 5 * 8 + 12 * 8 + 4 * 8
    assert ctypes.sizeof(a.Texture) == 8 + 7 * 8
    assert ctypes.sizeof(a.Light) == 8 + 7 * 8
    assert ctypes.sizeof(a.Camera) == 10 * 8 + 8
    assert ctypes.sizeof(a.Options) == 56
    assert ctypes.sizeof(a.Stats) == 5 * 8 + 4 * 8 + 8 + 8 + 48 + 8 + 8 + 16 * 8      # + exact_shadow_rerenders, ms_shade, ms_shadow_light[16]


@pytest.mark.parametrize("which", ["sphere", "cornell", "mesh", "cube", "empty", "mesh_big"])
def test_octree_matches_oracle_bit_for_bit(rtx, orc, which):
    if which == "sphere":
        flat, _ = scenes.sphere_scene(rtx)
    elif which == "cornell":
        flat, _ = scenes.cornell_scene(rtx)
    elif which == "mesh":
        flat, _ = scenes.mesh_scene(rtx)
    elif which == "mesh_big":
        flat, _ = scenes.mesh_scene(rtx, n_u=96, n_v=64)
    elif which == "cube":
        flat, _ = scenes.cube_obj_scene(rtx)
    else:
        flat = rtx.SceneBuilder().build()
    depth, bbox, shape, entries = rtx.octree_dump(flat)
    osc = orc.Scene(flat)
    info = osc.octree_info()
    obbox, oshape, oentries = osc.octree_dump()
    assert depth == info["depth"]
    assert bbox.shape == obbox.shape and shape.shape == oshape.shape
    assert np.array_equal(bbox.view(np.uint64), obbox.view(np.uint64))     # bit-exact boxes
    assert np.array_equal(shape, oshape)
    assert np.array_equal(entries, oentries)


def test_invalid_descriptions_are_rejected(rtx):
    lib = rtx.abi.load()
    b = rtx.SceneBuilder()
    b.sphere((0, 0, 0), 1.0, 5)          # material 5 does not exist
    flat = b.build()
    h = ctypes.c_void_p()
    assert lib.rtx_scene_create(ctypes.byref(flat.desc), 0, ctypes.byref(h)) == rtx.abi.RTX_ERR_INVALID_ARGUMENT
    assert b"material" in lib.rtx_last_error()
    assert lib.rtx_scene_create(None, 0, ctypes.byref(h)) == rtx.abi.RTX_ERR_INVALID_ARGUMENT
    assert lib.rtx_scene_destroy(None) == rtx.abi.RTX_OK


def test_no_device_means_error_not_fallback(rtx):
    """Without a GPU the product must refuse to render (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    flat, _ = scenes.sphere_scene(rtx)
    with pytest.raises(rtx.abi.RtxError) as e:
        rtx.Scene(flat)
    assert e.value.status == rtx.abi.RTX_ERR_NO_DEVICE
