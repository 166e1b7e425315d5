"""The C-ABI library loads without a GPU and exports every symbol include/y3.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "y3.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(y3_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from yolo_v3_tf2_amd import _lib
    declared = _header_functions()
    assert declared, "no functions parsed from include/y3.h"
    assert sorted(_lib.SYMBOLS) == declared


def test_library_exports_every_declared_symbol():
    from yolo_v3_tf2_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build liby3hip.so first (python __graft_entry__.py)"
    lib = _lib.load()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert lib.y3_version() >= 100
    assert lib.y3_device_count() >= 0


def test_error_convention_without_compute():
    """Invalid arguments come back as negative status + message, never an exception or abort."""
    from yolo_v3_tf2_amd import _lib
    lib = _lib.load()
    st = lib.y3_net_create(None, 0, None, 0, None, 0, None, 0, 0, None, 80, None)
    assert st == -1 and b"y3_net_create" in lib.y3_last_error()
    assert lib.y3_nms_workspace_bytes(0, 0) == 0
    # sort keys [B][next pow2 of N] u64 + kept-list spill [B][N][4] f32
    assert lib.y3_nms_workspace_bytes(2, 10647) == 2 * 16384 * 8 + 2 * 10647 * 16
    st = lib.y3_nms_padded(None, None, 1, 10, 100, 0.5, 0.1, None, None, None, 0, None)
    assert st == -1
    with pytest.raises(_lib.Y3Error):
        _lib.check(st, "y3_nms_padded")


def test_no_cpu_fallback():
    """Without a GPU the product path refuses to run instead of silently computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from yolo_v3_tf2_amd import _lib, runtime
    with pytest.raises(_lib.Y3Error):
        _lib.require_gpu()
    with pytest.raises(_lib.Y3Error):
        runtime.nms_padded(torch.zeros(1, 4, 4), torch.zeros(1, 4), 10, 0.5, 0.1)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "yolo-v3-tf2_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liby3oracle" not in txt, f


def test_rccl_not_loadable_is_a_status_not_a_crash():
    """librccl missing: y3_comm_get_unique_id / y3_comm_init_rank return Y3_ERR_COMM with a message (the dlopen
    failure once dereferenced a null dlerror()).  Own process: the binding is resolved once per process."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r)\n"
        "import yolo_v3_tf2_amd\n"
        "from yolo_v3_tf2_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = C.create_string_buffer(128)\n"
        "st = lib.y3_comm_get_unique_id(buf)\n"
        "msg = lib.y3_last_error()\n"
        "assert st == -6 and msg and b'librccl not found' in msg, (st, msg)\n"
        "h = C.c_void_p()\n"
        "st = lib.y3_comm_init_rank(buf, 1, 0, C.byref(h))\n"
        "assert st == -6 and not h.value and lib.y3_last_error(), st\n"
        "print('ok')\n" % ROOT)
    env = dict(os.environ, Y3_RCCL_LIB="/nonexistent/librccl.so.0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_exception_barrier_turns_a_failed_allocation_into_a_status():
    """a C++ exception must not cross extern "C" (it would terminate the host process of a ctypes / cgo / JNI
    caller).  Y3_TEST_FAIL_ALLOC=1 makes the object allocations of y3_net_create and y3_comm_init_rank throw std::bad_alloc the way
    operator new does: both calls come back with Y3_ERR_OOM and a message, the process lives, and the very next call works.  Own
    process: a crash here must fail the test, not the test runner."""
    import subprocess
    import sys
    code = (
        "import os, sys, ctypes as C; sys.path.insert(0, %r)\n"
        "import yolo_v3_tf2_amd\n"
        "from yolo_v3_tf2_amd import _lib\n"
        "lib = _lib.load()\n"
        "t = (_lib.TensorDesc * 2)(_lib.TensorDesc(3, 1), _lib.TensorDesc(32, 1))\n"
        "k = (C.c_int32 * 1)(0)\n"
        "c = (_lib.ConvDesc * 1)(_lib.ConvDesc(3, 1, 3, 32, 1, 1, 0, 0, 3, -1, -1, 1, 1, 1))\n"
        "a = (_lib.AuxDesc * 1)()\n"
        "o = (C.c_int32 * 3)(1, 1, 1)\n"
        "h = C.c_void_p()\n"
        "os.environ['Y3_TEST_FAIL_ALLOC'] = '1'\n"
        "st = lib.y3_net_create(t, 2, k, 1, c, 1, a, 0, 0, o, 0, C.byref(h))\n"
        "msg = lib.y3_last_error()\n"
        "assert st == -3 and not h.value and b'y3_net_create' in msg and b'bad_alloc' in msg, (st, msg)\n"
        "buf = C.create_string_buffer(128)\n"
        "hc = C.c_void_p()\n"
        "st = lib.y3_comm_init_rank(buf, 1, 0, C.byref(hc))\n"
        "assert st == -3 and not hc.value and b'y3_comm_init_rank' in lib.y3_last_error(), st\n"
        "os.environ['Y3_TEST_FAIL_ALLOC'] = '0'\n"
        "st = lib.y3_net_create(t, 2, k, 1, c, 1, a, 0, 0, o, 0, C.byref(h))\n"
        "assert st in (0, -5), (st, lib.y3_last_error())    # created, or 'no HIP device' on the CPU tier: an ordinary status either way\n"
        "if st == 0: lib.y3_net_destroy(h)\n"
        "st = lib.y3_net_create(t, 2, k, 1, c, -1, a, 0, 0, o, 0, C.byref(h))\n"
        "assert st == -1, st                                # a negative count is an argument error, not a length_error from std::vector\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_no_extern_c_entry_without_the_barrier():
    """Every y3_status entry point defined in csrc/*.cpp is a function-try-block closed by Y3_CATCH (the barrier is structural, not
    a convention someone has to remember)."""
    for f in ("y3_api.cpp", "comm.cpp"):
        src = open(os.path.join(ROOT, "yolo-v3-tf2_amd", "csrc", f)).read()
        src = src[src.index('extern "C" {'):]
        defs = re.findall(r"^y3_status (y3_\w+)\((?:[^{};]|\n)*?\)\n(try \{|\{)", src, flags=re.M)
        assert defs, f
        for name, opener in defs:
            if name == "y3_comm_info":      # two stores through caller pointers: nothing in it can throw
                continue
            assert opener == "try {", (f, name)
            assert 'Y3_CATCH("%s")' % name in src, (f, name)


def test_tile_built_reports_the_default_tile_set():
    """y3_tile_built (no GPU needed): the library holds exactly the tiles of the Python-side tables, and exactly the tiles some plan can select;
    every other id of rounds 1-4 (ablations, stream-K, residual prefetch, pipelined / tap-row-reuse / four-wave bf16 tiles, and the tuner
    candidates no table kept) is retired."""
    from yolo_v3_tf2_amd import _lib
    f32 = [t for t in range(len(_lib.TILES) + 16) if _lib.tile_built(_lib.Y3_DTYPE_F32, t)]
    bf16 = [t for t in range(len(_lib.TILES_BF16) + 4) if _lib.tile_built(_lib.Y3_DTYPE_BF16, t)]
    x2 = [t for t in range(len(_lib.TILES_X3) + 16) if _lib.tile_built(_lib.Y3_DTYPE_F32X2, t)]
    x3 = [t for t in range(len(_lib.TILES_X3) + 16) if _lib.tile_built(_lib.Y3_DTYPE_F32X3, t)]
    assert not _lib.tile_built(_lib.Y3_DTYPE_F32, -1) and not _lib.tile_built(_lib.Y3_DTYPE_F32, len(_lib.TILES))
    assert not _lib.tile_built(7, 0)
    assert x3 == list(_lib.TILES_X3_BUILT) and x2 == list(_lib.TILES_X2_BUILT)
    assert f32 == [t for t in range(len(_lib.TILES)) if _lib.TILES[t][0] > 0]
    assert bf16 == [t for t in range(len(_lib.TILES_BF16)) if _lib.TILES_BF16[t][0] > 0]
    assert x3 == [t for t in range(len(_lib.TILES_X3)) if _lib.TILES_X3[t][0] > 0 and t not in (26, 27)]      # 26, 27: two-plane mode only
    assert all(_lib.TILES_X3[t][0] > 0 for t in x2)
    # every tile a committed tuning table names is in the default set
    import glob
    import json
    named = {}
    for f in glob.glob(os.path.join(ROOT, "yolo-v3-tf2_amd", "tuning", "*.json")):
        tag = os.path.basename(f).split("_")[0]
        dt = {"f32": _lib.Y3_DTYPE_F32, "bf16": _lib.Y3_DTYPE_BF16, "f32x3": _lib.Y3_DTYPE_F32X3, "f32x2": _lib.Y3_DTYPE_F32X2}[tag]
        for sig, t in json.load(open(f))["tiles"].items():
            assert t < 0 or _lib.tile_built(dt, t), (f, sig, t)
            if t >= 0:
                named.setdefault(tag, set()).add(t)
    # ... and nothing else is: every tile id the library builds is named by a packaged table or picked by the library's heuristics
    # (_lib.HEURISTIC_TILES mirrors choose_tile* / the head-decode fallback of csrc/y3_api.cpp) -- no kernel ships that no plan can select
    for tag, built in (("f32", f32), ("bf16", bf16), ("f32x3", x3), ("f32x2", x2)):
        assert set(built) == named.get(tag, set()) | set(_lib.HEURISTIC_TILES[tag]), (tag, sorted(set(built) ^ (named.get(tag, set()) | set(_lib.HEURISTIC_TILES[tag]))))

