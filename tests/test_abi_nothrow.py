"""No C++ exception crosses the C ABI (the callers of VO_utility.h:96-117 expect a status).  tests/cpp/abi_nothrow.cpp
replaces the global `operator new` with one that throws std::bad_alloc for an allocation size the test arms, and drives
  * on CPU: the host-only entries (uvo_select_estimation_method's median scratch; uvo_rodrigues; context creation without a device);
  * on the GPU box: an operator that stages through a std::vector on the calling thread (uvo_triangulate_points) and a lane WORKER's
    allocation (the pose stage of a pipelined mono frame) -- the frame fails with UVO_CAPACITY instead of std::terminate, and the
    context gives the first run's results afterwards."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from ergo_uvo_amd import _lib
    _lib.build()
    exe = str(tmp_path / "abi_nothrow")
    libdir = os.path.join(ROOT, "ergo_uvo_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "abi_nothrow.cpp"),
                           "-o", exe, "-L", libdir, "-luvo_hip", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined", "-pthread"])
    return exe


def test_host_only_entries_return_a_status_under_a_failing_operator_new(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe, "host"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr


def test_every_extern_c_body_is_guarded():
    """Every extern "C" definition of ctx.hip that can allocate is a function-try-block ending in one of the UVO_ABI_CATCH handlers; the
    accessors and the two parameter initialisers that remain cannot throw (they return a member or a constant / store into the caller's struct)."""
    import re
    src = open(os.path.join(ROOT, "ergo_uvo_amd", "csrc", "ctx.hip")).read()
    defs = re.findall(r'^extern "C" [^;{]*?\b(uvo_\w+)\([^;{]*?\)\s*(try\s*\{|\{)', src, flags=re.M | re.S)
    plain = sorted(name for name, how in defs if not how.startswith("try"))
    assert plain == sorted(["uvo_last_error", "uvo_ctx_stream", "uvo_ctx_warning", "uvo_ctx_pending", "uvo_timing_count", "uvo_timing_name",
                            "uvo_params_default_stereo", "uvo_params_default_mono"]), plain           # (memset + stores)
    assert sum(1 for _, how in defs if how.startswith("try")) == src.count("} UVO_ABI_CATCH") >= 50
    for f in sorted(os.listdir(os.path.join(ROOT, "ergo_uvo_amd", "csrc"))):
        if f.endswith(".hip") and f != "ctx.hip":
            assert 'extern "C"' not in open(os.path.join(ROOT, "ergo_uvo_amd", "csrc", f)).read(), f     # the ABI lives in ctx.hip alone


@pytest.mark.gpu
def test_operators_and_lane_workers_return_a_status_under_a_failing_operator_new(tmp_path, mono_small):
    exe = _build(tmp_path)
    paths = []
    for k in (0, 1):
        path = str(tmp_path / f"img{k}.raw")
        np.ascontiguousarray(mono_small[k], np.uint8).tofile(path)
        paths.append(path)
    h, w = mono_small[0].shape
    p = subprocess.run([exe, "gpu", paths[0], paths[1], str(w), str(h)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
