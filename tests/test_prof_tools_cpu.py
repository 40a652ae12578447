"""The profiling tools that produce the tracked evidence (tools/prof_window.py, tools/prof_timeline.py) and bench.py's reader of
the window table, on a synthetic rocprofv3 kernel trace: steps are delimited by `adam_flat_kernel`, the streams are separated by
queue, a trace without the marker is an ERROR (round 3 committed an empty table because the tool crashed silently), and the first
per-function row is what bench.py's `roofline` names."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, steps, with_marker=True):
    rows, t = [], 1000
    for s in range(steps):
        for name, dur, q in (("void (anonymous namespace)::sa_wgrad_kernel<128, 64, true>(float const*)", 200_000, "1"),
                             ("void (anonymous namespace)::sa_wgrad_kernel<64, 64, false>(float const*)", 100_000, "1"),
                             ("void (anonymous namespace)::tf_rows_kernel<false>(TfRowsArgs)", 12_000, "1"),
                             ("Cijk_Ailk_Bljk_SB_MT64x32x32", 15_000, "1"),
                             ("void at::native::vectorized_elementwise_kernel<4, X>(int)", 5_000, "1"),
                             ("void (anonymous namespace)::fps_bucket_kernel<10, true>(float const*)", 900_000, "3")):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur, "Queue_Id": q})
            t += dur + 1000 if q == "1" else 0
        if with_marker:
            rows.append({"Kernel_Name": "(anonymous namespace)::adam_flat_kernel(float*)", "Start_Timestamp": t, "End_Timestamp": t + 40_000,
                         "Queue_Id": "1"})
            t += 50_000
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"], quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)


def test_window_table_names_the_largest_main_stream_function(tmp_path):
    tr = tmp_path / "trace.csv"
    _trace(tr, steps=4)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_window.py"), str(tr), "3", "10"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith("timed window: 3 steps") and "6 kernels/step" in lines[0] and "side stream(s)" in lines[0]
    assert "rocBLAS/MIOpen 0.015 ms/step" in out.stdout and "at::native 0.005 ms/step" in out.stdout
    i = lines.index("main_ms/step,calls/step,avg_us,function")
    first = next(csv.reader([lines[i + 1]]))
    assert first[3] == "sa_wgrad_kernel" and abs(float(first[0]) - 0.3) < 1e-6 and float(first[1]) == 2.0   # both instantiations merged
    assert any(l.endswith("side,\"void (anonymous namespace)::fps_bucket_kernel<10, true>(float const*)\"") for l in lines)
    # bench.py reads the same row
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_x_timed_window_kernels.csv").write_text(out.stdout)
    (prof / "r00_empty_timed_window_kernels.csv").write_text("")
    code = ("import sys, os; sys.argv=['bench.py']; os.environ['WORLD_SIZE']='1'; import importlib.util as u; "
            f"s=u.spec_from_file_location('b', r'{os.path.join(ROOT, 'bench.py')}'); m=u.module_from_spec(s); s.loader.exec_module(m); "
            f"m.ROOT=r'{tmp_path}'; print(m.window_table_top()[:2])")
    got = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert got.returncode == 0 and "('sa_wgrad_kernel', 'r99_x_timed_window_kernels.csv')" in got.stdout, got.stderr[-2000:]


def test_a_trace_without_the_step_marker_is_an_error(tmp_path):
    tr = tmp_path / "trace.csv"
    _trace(tr, steps=4, with_marker=False)
    for tool, args in (("prof_window.py", [str(tr), "3"]), ("prof_timeline.py", [str(tr), "0"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True)
        assert out.returncode != 0 and "adam_flat_kernel" in out.stderr and out.stdout.strip() == "", (tool, out.stdout[:200])
    _trace(tr, steps=2)   # fewer steps than asked for
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_window.py"), str(tr), "3"], capture_output=True, text=True)
    assert out.returncode != 0 and "need at least 4" in out.stderr
