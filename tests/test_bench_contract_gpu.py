"""bench.py as the driver runs it: one JSON line on stdout with the keys of the bench contract (task statement, section 4), at a reduced
sample count so that the test takes seconds; the N = 2 launch line of the contract with the gloo backend on the one GPU of the box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

ENV = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_n1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "16", "--cpu-spp", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=ENV, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "Msamples/s" and j["unit"] == "Msamples/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["dtype"] == "f32" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - j["config"]["samples_per_step"] / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * j["value"]
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4
    assert ro["kernel"].startswith("lrt::k_render<1, 1024, true") and ro["avg_launch_ms"] > 0 and ro["avg_launch_ms"] <= j["ms_per_step"] * 1.001
    assert ro["traffic"] is None or ro["traffic_source"]            # traffic only from a profile of THIS build of the kernels (kernel_source_id)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "spp" in cb["sample"] and "orc_render_scalar" in cb["build"]
    assert j["rmse_vs_oracle"]["value"] <= j["rmse_vs_oracle"]["tolerance"]
    hg = j["hg_phase"]                                            # BASELINE.json config 3 as worded (HG phase) rides in the default line
    assert hg["g"] == 0.7 and hg["value"] > 0 and "HG phase" in hg["workload"]
    assert j["host_visible"]["value"] > 0 and j["host_visible"]["value"] <= j["value"] * 1.05


@pytest.mark.parametrize("config,samples", [("c3", 1920 * 1080 * 8), ("c2", 1080 * 1080 * 8), ("c5", 1920 * 1080 * 16)])
def test_bench_line_two_ranks_gloo(config, samples):
    """the driver's N = 2 launch line (gloo instead of RCCL: both ranks share the box's one GPU): film all-reduce + develop for the
    forward configurations, gradient all-reduce for the PRB one (c5: ld sampler, 8 -> 16 spp; tent filter: weight film over tiles + halo)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29700 + (os.getpid() + len(config) * 7 + samples) % 200),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "8", "--backend", "gloo", "--config", config]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "cpu_baseline" not in j and "hg_phase" not in j
    assert j["config"]["samples_per_step"] == samples and j["value"] > 0
    assert "2 GPUs" in j["config"]["parallelism"]


@pytest.mark.parametrize("config", ["het", "mis"])
def test_bench_f4_configs(config):
    """SURVEY.md 8f row 4 has bench configurations of its own: a short run of each prints a line whose roofline names
    the heterogeneous-media / volpathmis kernel and prices its wider record."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "1", "--warmup", "1", "--spp", "4", "--width", "256", "--height", "256", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = last_json(r.stdout); ro = j["roofline"]
    assert ro["kernel"].startswith("lrt::k_render<101," if config == "het" else "lrt::k_render<5,") and ro["record_bytes"] == (104 if config == "het" else 168)
    assert j["value"] > 0 and ro["records_per_sample"] > 0 and j["data"].startswith("synthetic")
