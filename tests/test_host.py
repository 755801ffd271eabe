"""Host-side logic that needs no GPU: the worker pool that spreads the per-frame line-graph tails over host threads."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_pool_runs_every_index_once(tmp_path):
    exe = str(tmp_path / "host_pool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "host_pool_stress.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
