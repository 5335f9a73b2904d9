"""CPU-side checks of the C-ABI library: it loads and exports every symbol
include/beifong_hip.h declares; no compute call is made without a GPU."""
import ctypes as C
import os
import re

import pytest

from beifong_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_match_binding_list():
    hdr = open(os.path.join(ROOT, "include", "beifong_hip.h")).read()
    decl = set(re.findall(r"\b(bf_[a-z_]+)\s*\(", hdr))
    assert decl == set(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    _ensure_built()
    lib = C.CDLL(capi.LIB_PATH)
    for name in capi.EXPORTED_SYMBOLS:
        assert hasattr(lib, name), name
    assert lib.bf_version() == capi.BF_ABI_VERSION


def test_struct_sizes_match_header():
    # compile a tiny C program against the header and compare sizeof()
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "beifong_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(bf_material),sizeof(bf_shape),sizeof(bf_emitter),sizeof(bf_sensor),sizeof(bf_scene_desc),' \
          'sizeof(bf_launch),sizeof(bf_path_record),sizeof(bf_stats),sizeof(bf_scene_info),sizeof(bf_batch));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()
    sizes = [C.sizeof(t) for t in (capi.bf_material, capi.bf_shape, capi.bf_emitter, capi.bf_sensor, capi.bf_scene_desc,
                                   capi.bf_launch, capi.bf_path_record, capi.bf_stats, capi.bf_scene_info, capi.bf_batch)]
    assert [int(x) for x in out] == sizes


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(capi.BeifongError):
        capi.load_library(str(tmp_path / "nope.so"))


def test_no_device_is_an_error_not_a_fallback():
    _ensure_built()
    lib = capi.load_library()
    if lib.bf_device_count() > 0:
        pytest.skip("a GPU is present")
    from beifong_amd import scenes
    sd, _ = scenes.trans_rad(4)
    with pytest.raises(capi.BeifongError, match="no HIP device"):
        capi.Scene(sd)


def test_missing_rccl_is_unsupported_not_a_crash():
    """bf_allreduce_device when librccl cannot be loaded: BF_ERR_UNSUPPORTED with the loader's message (the
    message was built from TWO dlerror() calls, the second of which returns NULL -> std::string(NULL) -> abort).  RCCL is
    loaded once per process, so the forced not-found path (BF_RCCL_LIB) runs in a child process; no GPU is touched: the
    call fails before any device work."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from beifong_amd import capi\n"
        "lib = capi.load_library()\n"
        "lib.bf_allreduce_device.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p), C.c_uint64, C.POINTER(C.c_void_p)]\n"
        "lib.bf_allreduce_device.restype = C.c_int\n"
        "devs = (C.c_int * 2)(0, 1)\n"
        "buf = (C.c_float * 4)()\n"
        "bufs = (C.c_void_p * 2)(C.addressof(buf), C.addressof(buf))\n"
        "st = lib.bf_allreduce_device(devs, 2, bufs, 4, None)\n"
        "print(st, lib.bf_last_error().decode())\n"
    )
    env = dict(os.environ, BF_RCCL_LIB="/nonexistent/librccl_missing.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    st, msg = r.stdout.strip().split(" ", 1)
    assert int(st) == capi.BF_ERR_UNSUPPORTED, r.stdout
    assert "librccl_missing.so" in msg and "not found" in msg
