"""The dependency-free .npz reader/writer of the optional OpenCV dumper (tools/opencv_oracle/npz_io.h) round-trips numpy's files,
and the dumper's inputs are deterministic; the dumper itself needs OpenCV and is never built here."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_npz_io_roundtrip(tmp_path):
    src = tmp_path / "rt.cpp"
    src.write_text('#include "npz_io.h"\nint main(int c, char** v) { npz::File f = npz::load(v[1]); npz::save(v[2], f); return (int)f.size() == 6 ? 0 : 1; }\n')
    exe = tmp_path / "rt"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tools", "opencv_oracle"), str(src), "-o", str(exe)])
    rng = np.random.default_rng(0)
    a = dict(img=rng.integers(0, 256, (37, 53)).astype(np.uint8), rgb=rng.integers(0, 256, (5, 7, 3)).astype(np.uint8), K=rng.normal(size=(3, 3)),
             pts=rng.normal(size=(11, 2)).astype(np.float32), idx=np.arange(9, dtype=np.int32), empty=np.zeros((0, 2), np.float32))
    np.savez(tmp_path / "in.npz", **a)
    assert subprocess.call([str(exe), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")]) == 0
    b = np.load(tmp_path / "out.npz")
    for k, v in a.items():
        assert b[k].dtype == v.dtype and b[k].shape == v.shape and np.array_equal(b[k], v), k


def test_dumper_inputs_are_deterministic():
    sys.path.insert(0, os.path.join(ROOT, "tools", "opencv_oracle"))
    import make_inputs
    a, b = make_inputs.build_inputs(), make_inputs.build_inputs()
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert a["left0"].shape == (360, 640) and a["pnp_X"].shape == (800, 3) and a["e_x1"].dtype == np.float32


def test_dumper_source_cites_the_reference_call_sites():
    text = open(os.path.join(ROOT, "tools", "opencv_oracle", "opencv_oracle.cpp")).read()
    for needle in ("xfeatures2d::SURF::create", "knnMatch", "triangulatePoints", "solvePnPRansac", "findEssentialMat", "recoverPose",
                   "findHomography", "decomposeHomographyMat", "getOptimalNewCameraMatrix", "INTER_AREA", "createCLAHE", "VOU:", "VO:"):
        assert needle in text, needle


def test_orb_pattern_extractor_reads_an_annotated_initialiser(tmp_path):
    """tools/orb_pattern_from_opencv_source.py on a stand-in source file (seeded integers -- not OpenCV's table -- in orb.cpp's layout, with
    its block and line comments): 1024 integers come back in order, in the text format the C++ surface parses; a short table is refused."""
    import importlib.util, random
    spec = importlib.util.spec_from_file_location("orbpat", os.path.join(ROOT, "tools", "orb_pattern_from_opencv_source.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    rnd = random.Random(5)
    vals = [rnd.randint(-13, 13) for _ in range(1024)]
    lines = [f"    {vals[4*k]},{vals[4*k+1]}, {vals[4*k+2]},{vals[4*k+3]} /*mean (0.{k}), correlation ({k % 7})*/," for k in range(256)]
    src = "static int other_[4] = { 1, 2, 3, 4 };\nstatic int bit_pattern_31_[256*4] =\n{\n" + "\n".join(lines) + "  // the end\n};\nstatic void f() { int x[2] = {5, 6}; }\n"
    cpp, out = tmp_path / "orb.cpp", tmp_path / "pattern.txt"
    cpp.write_text(src)
    assert m.extract(src) == vals
    assert m.main(["x", str(cpp), str(out)]) == 0
    back = [int(v) for v in out.read_text().replace(",", " ").split()]
    assert back == vals
    with pytest.raises(ValueError):
        m.extract(src.replace(lines[7] + "\n", ""))
