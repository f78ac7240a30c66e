#!/usr/bin/env python3
"""Benchmark of the hip_ad_rgb hot path on BASELINE.json's metric: Msamples/s at fixed spp.

A "step" is one complete render of the workload (ray generation -> wavefront loop -> film ->
develop, plus the RCCL film all-reduce when N > 1).  N = 1 workload: BASELINE config C3
(scenes/Liver-SingleMesh, plain `volpath`, 1920x1080, 512 spp, max_depth 12).  For N > 1 the SAME
image is sharded by 32x32 pixel tiles over the ranks (strong scaling) and the per-rank raw films are
summed with one all-reduce before develop.

Prints ONE JSON line on rank 0 (contract: see the task statement).
"""
import argparse
import json
import glob, os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STATE_BYTES = 88               # bytes of one path record across the SoA streams (csrc/device_types.h)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--spp", type=int, default=512)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default=os.path.join(ROOT, "scenes", "Liver-SingleMesh", "mitsuba3", "scene.xml"))
    p.add_argument("--integrator", default="volpath")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", help="process-group backend; gloo allows a multi-rank rehearsal on a single GPU")
    p.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0: calibrate to ~15 s)")
    return p.parse_args()


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes expose
    all cores of the host but grant a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max": n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0: n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def main():
    a = parse()
    import torch
    import liverrenderer_amd as mi
    from liverrenderer_amd.distributed import render_distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hip_ad_rgb back-end has no CPU fallback)")
    local_rank %= max(torch.cuda.device_count(), 1)            # several ranks share a GPU only in a gloo rehearsal
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl": dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else: dist.init_process_group(a.backend)

    scene = mi.load_file(a.scene, integrator=a.integrator, spp=a.spp, res_width=a.width, res_height=a.height)
    h, w, _ = scene.film_shape()
    C = scene.raw_channels()
    n_samples = w * h * a.spp

    def step(seed):
        if world > 1:
            img, raw = render_distributed(scene, spp=a.spp, seed=seed)
            return img
        dev = torch.device("cuda", local_rank)
        film = torch.zeros((h, w, C), dtype=torch.float32, device=dev)
        image = torch.empty((h, w, C - 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        scene.render_to_device(film.data_ptr(), image.data_ptr(), spp=a.spp, seed=seed, device=local_rank)
        return image

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(1000 + i)
    fence()
    t0 = time.perf_counter()
    kern_ms = iters = shadows = launches = records = 0.0
    for i in range(a.steps):
        img = step(i)
        st = scene.stats()
        kern_ms += st["kernel_ms"]; iters += st["n_iter"]; shadows += st["n_shadow"]; launches += st["n_launches"]; records += st["n_records"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = n_samples * a.steps / dt / 1e6

    # ---- roofline of the dominant kernel (k_render: ONE launch per step), this rank's launch.  Algorithmic bytes per
    # launch: every queued path record is written once and read once (88 B each way; fresh camera paths run their first
    # trip in registers and loop trips the look-ahead retires early move no record, so n_records <= n_iter - n_samples),
    # plus 4*C bytes per pixel of film.  Duration: HIP events around the launch on the library's stream (kernel_ms).
    n_rank = st["n_samples"]
    alg_bytes = 2.0 * STATE_BYTES * records + 4.0 * C * w * h * a.steps / max(world, 1)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # HBM traffic per launch from the PMC passes of the same workload (scripts/profile_bench.sh -> profiles/*traffic.json)
    traffic = None
    workload = f"C3 Liver-SingleMesh {a.integrator} {w}x{h} {a.spp} spp max_depth 12 (homogeneous medium, isotropic phase, envmap)"
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            tj = json.load(open(tf))
            if tj.get("workload") == workload and world == 1: traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "lrt::k_render<%d, 1024, true, false>" % (0 if a.integrator == "path" else 1),
                "launches_per_step": launches / a.steps, "avg_launch_ms": kern_ms / max(launches, 1),
                "alg_bytes_per_launch": alg_bytes / max(launches, 1), "iterations_per_sample": iters / (n_rank * a.steps),
                "records_per_sample": records / (n_rank * a.steps)}

    out = {"metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "reference scene files (scene.xml, liver2.obj, tissue_n.png, cavidade_latitude.exr)",
           "config": {"workload": workload,
                      "width": w, "height": h, "spp": a.spp, "samples_per_step": n_samples,
                      "parallelism": "1 GPU" if world == 1 else f"32x32 pixel tiles over {world} GPUs + RCCL film all-reduce"},
           "roofline": roofline}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # ---- CPU baseline: the oracle (our restatement of the reference's CPU path) on the host cores,
        # same scene/resolution at a reduced spp, and the GPU-vs-oracle RMSE at that spp (same seed).
        import orc
        cores = usable_cores()
        if a.cpu_spp <= 0:                       # calibrate on 1 spp so that the sample takes ~15 s of wall time
            cs = mi.load_file(a.scene, integrator=a.integrator, spp=1, res_width=a.width, res_height=a.height)
            t1 = time.perf_counter(); orc.OrcScene(cs).render(threads=cores, spp=1, seed=0); c1 = time.perf_counter() - t1
            a.cpu_spp = int(min(a.spp, max(1, round(15.0 / max(c1, 1e-3)))))
        cs = mi.load_file(a.scene, integrator=a.integrator, spp=a.cpu_spp, res_width=a.width, res_height=a.height)
        o = orc.OrcScene(cs)
        t1 = time.perf_counter()
        cimg = o.render(threads=cores, spp=a.cpu_spp, seed=0)
        ct = time.perf_counter() - t1
        gimg = cs.render(spp=a.cpu_spp, seed=0)
        rmse = float(np.sqrt(np.mean((gimg.astype(np.float64) - cimg.astype(np.float64)) ** 2)))
        out["cpu_baseline"] = {"value": round(w * h * a.cpu_spp / ct / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "sample": f"same scene and resolution at {a.cpu_spp} spp ({w * h * a.cpu_spp} samples, {ct:.1f} s)"}
        out["rmse_vs_oracle"] = {"value": rmse, "spp": a.cpu_spp, "tolerance": 1e-4,
                                 "note": "same seed: per-lane radiance is bit-identical, the film differs by float-atomic order only"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
