"""Host-side BVH of the product (beifong_amd/csrc/bf_bvh.cpp: binned SAH + collapse to four-wide nodes), checked on
the CPU through a small harness (tests/native/bvh_check.cpp): every triangle in exactly one leaf, triangles inside
their leaf's box, child boxes inside the parent's (up to the padding), leaf size, depth bound, and the worst-case
traversal stack the kernels size their spill columns from."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from beifong_amd import meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    out = tmp_path_factory.mktemp("bvh") / "libbvh_check.so"
    src = [os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "beifong_amd", "csrc", "bf_bvh.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", str(out)] + src, check=True)
    lib = C.CDLL(str(out))
    lib.bvh_check.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32 * 8)]
    lib.bvh16_check.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32 * 8)]
    lib.bvh4q_check.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32 * 8)]
    return lib


def _check(lib, v, f):
    tri = np.ascontiguousarray(np.asarray(v, np.float32)[np.asarray(f)].reshape(-1, 9))
    out = (C.c_uint32 * 8)()
    rc = lib.bvh_check(tri.shape[0], tri.ctypes.data, C.byref(out))
    assert rc == 0, f"invariant {rc} violated"
    return dict(zip(("nodes4", "leaves", "max_leaf", "depth4", "stack_need", "depth2", "nodes2", "leaf_cap"), list(out)))


def test_bus_and_car_meshes(checker):
    for v, f in (meshgen.bus(20000, seed=1), meshgen.car_body(30000, seed=2)[:2], meshgen.motorbike(10000, seed=5)):
        s = _check(checker, v, f)
        assert s["max_leaf"] <= s["leaf_cap"] and s["depth2"] <= 31
        assert s["depth4"] <= s["depth2"] and s["stack_need"] <= 3 * s["depth4"]
        assert s["leaves"] >= len(f) / s["leaf_cap"]
        assert s["nodes4"] < s["nodes2"]                    # the collapse removes interior nodes


def test_random_soup_and_degenerate_inputs(checker):
    v, f = meshgen.triangle_soup(5000, seed=3)
    _check(checker, v, f)
    # all centroids coincide: the builder must fall back to index splits and still bound the depth
    v = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (3000, 1))
    f = np.arange(9000).reshape(-1, 3)
    s = _check(checker, v, f)
    assert s["depth2"] <= 31
    # a long sliver line: worst case for SAH depth
    x = np.arange(20000, dtype=np.float32)[:, None] * np.float32(1e-3)
    v = np.concatenate([np.concatenate([x, 0 * x, 0 * x], 1), np.concatenate([x + 5e-4, 0 * x + 1e-4, 0 * x], 1),
                        np.concatenate([x, 0 * x, 0 * x + 1e-4], 1)], 1).reshape(-1, 3)
    f = np.arange(60000).reshape(-1, 3)
    s = _check(checker, v, f)
    assert s["depth2"] <= 31 and s["stack_need"] <= 93
    # tiny inputs
    for n in (1, 2, 3, 5):
        v, f = meshgen.triangle_soup(n, seed=n)
        _check(checker, v, f)


def _check16(lib, v, f):
    tri = np.ascontiguousarray(np.asarray(v, np.float32)[np.asarray(f)].reshape(-1, 9))
    out = (C.c_uint32 * 8)()
    rc = lib.bvh16_check(tri.shape[0], tri.ctypes.data, C.byref(out))
    assert rc == 0, f"sixteen-wide invariant {rc} violated"
    return dict(zip(("nodes16", "leaves", "max_leaf", "depth16", "stack_need", "internal"), list(out)[:6]))


def test_sixteen_wide_collapse_for_the_tail_kernel(checker):
    """bf::collapse_bvh16 (the tail kernel's row traversal): same leaf order as the four-wide tree, leaves
    of at most 16 contiguous slots, a much shallower tree, and a worst-case stack that fits the kernels' 512 entries."""
    for v, f in (meshgen.bus(20000, seed=1), meshgen.car_body(30000, seed=2)[:2], meshgen.motorbike(10000, seed=5),
                 meshgen.triangle_soup(5000, seed=3)):
        s4 = _check(checker, v, f)
        s = _check16(checker, v, f)
        assert s["max_leaf"] <= 16 and s["leaves"] >= len(f) / 16
        assert s["depth16"] < s4["depth4"]
        assert s["stack_need"] <= 16 * s["depth16"] and s["stack_need"] <= 512
    for n in (1, 2, 3, 5, 16, 17, 40):
        v, f = meshgen.triangle_soup(n, seed=n)
        s = _check16(checker, v, f)
        assert s["nodes16"] == (0 if n <= 16 else s["internal"])


def test_quantised_nodes_contain_the_fp32_boxes(checker):
    """bf::quantise_bvh4 (wf_trace's 64-byte nodes): every 8-bit child box, evaluated in fp32 as the kernels evaluate it,
    contains the fp32 box it replaces and grows it by at most one quantum (1 / 255 of the node's extent, or twice that
    when the extent sits just below a power of two); child references unchanged; unused slots inverted."""
    for v, f in (meshgen.bus(20000, seed=1), meshgen.car_body(30000, seed=2)[:2], meshgen.motorbike(10000, seed=5),
                 meshgen.triangle_soup(5000, seed=3), meshgen.triangle_soup(3, seed=9)):
        tri = np.ascontiguousarray(np.asarray(v, np.float32)[np.asarray(f)].reshape(-1, 9))
        out = (C.c_uint32 * 8)()
        rc = checker.bvh4q_check(tri.shape[0], tri.ctypes.data, C.byref(out))
        assert rc == 0, f"quantised-node invariant {rc} violated"
        assert out[1] <= 1e6 / 255 * 2.01
