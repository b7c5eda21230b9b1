"""bench.py's roofline.traffic comes from two rocprofv3 counter passes it runs as child processes before its first GPU call
(measure_hbm_traffic).  CPU-side checks of that plumbing with a stand-in `rocprofv3` on PATH: the per-launch figure is the mean over
the launches after the first two of the counter summed over its per-XCD rows, FETCH_SIZE counted twice (MI355X_MICROARCH.md's gfx950
reading), and every way the passes can fail ends in (None, reason) -- the bench then falls back to the committed summary, it never dies."""
import os
import stat
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE = textwrap.dedent("""\
    #!%(py)s
    import os, sys
    a = sys.argv[1:]
    counter = a[a.index("--pmc") + 1]; out = a[a.index("-d") + 1]
    mode = os.environ.get("FAKE_ROCPROF", "ok")
    if mode == "fail":
        sys.exit(3)
    if mode == "hang":
        import time; time.sleep(60)
    os.makedirs(os.path.join(out, "host"), exist_ok=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,SGPR_Count,Counter_Name,Counter_Value"]
    base = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 10.0}[counter]
    for d in range(1, 7):                                   # six launches of the kernel, eight XCD rows each; two of another kernel
        for x in range(8):
            v = (base + (100 if d <= 2 else 0) + d) / 8.0   # the first two launches read more (cold caches): they are skipped
            rows.append(f"{d},{d},1,1,1,1,1,1,\\"void uvo::k_hessian_nms_all<64, 32>(uvo::LanePair)\\",512,0,0,64,32,{counter},{v}")
        rows.append(f"{d + 100},{d + 100},1,1,1,1,1,2,\\"uvo::k_other(int)\\",256,0,0,8,8,{counter},99999")
    if mode != "empty":
        open(os.path.join(out, "host", "1_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
""")


def _with_fake(tmp_path, monkeypatch, mode):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE % {"py": sys.executable})
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("FAKE_ROCPROF", mode)
    for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR"):
        monkeypatch.delenv(k, raising=False)
    import bench
    return bench


def test_counter_passes_are_averaged_per_launch_and_fetch_counts_twice(tmp_path, monkeypatch):
    bench = _with_fake(tmp_path, monkeypatch, "ok")
    nbytes, det = bench.measure_hbm_traffic(timeout_s=30)
    fetch = 1000.0 + (3 + 4 + 5 + 6) / 4.0                     # launches 3..6, the eight XCD rows of each summed
    write = 10.0 + (3 + 4 + 5 + 6) / 4.0
    assert det["launches_averaged"] == 4 and abs(det["FETCH_SIZE_KiB_per_launch"] - fetch) < 0.06 and abs(det["WRITE_SIZE_KiB_per_launch"] - write) < 0.06
    assert nbytes == int((2 * fetch + write) * 1024)


def test_failing_hanging_or_empty_passes_are_reported_not_raised(tmp_path, monkeypatch):
    for mode, needle in (("fail", "failed"), ("empty", "failed"), ("hang", "did not finish")):
        bench = _with_fake(tmp_path, monkeypatch, mode)
        nbytes, det = bench.measure_hbm_traffic(timeout_s=2 if mode == "hang" else 30)
        assert nbytes is None and needle in det["skipped"], (mode, det)


def test_no_profiler_or_a_profiled_parent_skips_the_passes(tmp_path, monkeypatch):
    import bench
    monkeypatch.setenv("PATH", str(tmp_path))                  # nothing called rocprofv3 here
    nbytes, det = bench.measure_hbm_traffic(timeout_s=5)
    assert nbytes is None and "not on PATH" in det["skipped"]
    bench = _with_fake(tmp_path, monkeypatch, "ok")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    nbytes, det = bench.measure_hbm_traffic(timeout_s=5)
    assert nbytes is None and "profiler already" in det["skipped"]
