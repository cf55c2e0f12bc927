#!/usr/bin/env python3
"""Benchmark of the MI355X modulated-SIREN path: Mpixels/s reconstructed (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input per GPU:
``ModulatedSiren.forward`` (encoder -> modulator -> fused SIREN trunk) on the 400 tiles
(32x32 in, 24x24 out) of ``--slices`` 320x320 slice(s) (default 1 = BASELINE.json configs[1]),
tiles already resident in HBM, outputs left in HBM.  Patches are independent, so with N GPUs
every rank processes its own slices (weak scaling); the only collective on the path is the RCCL
broadcast of the weight blob from rank 0 at load time, outside the timed region.

Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement for the fields).
torch is used for process-group plumbing (RCCL/gloo) and device synchronisation only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SUSTAINED_F16_MFMA_TFLOPS = 1476.0
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (= vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--slices", type=int, default=1, help="320x320 slices per GPU per step")
    ap.add_argument("--activation", default="sine", choices=["sine", "morlet"])
    ap.add_argument("--model", default="baseline", choices=["baseline", "deep_residual"],
                    help="baseline = 5 x 256 (BASELINE configs 1-4); deep_residual = 10 x 512, latent 128, residual "
                         "(config 5; own semantics, precision defaults to bf16)")
    ap.add_argument("--precision", default=None, choices=["fp32", "f16x3", "bf16", "f16"],
                    help="trunk arithmetic: f16x3 = split-fp16, 3 f16 MFMAs per product, fp32-equivalent accuracy "
                         "(default); fp32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2 = consecutive steps alternate between two HIP streams (independent slices overlap)")
    ap.add_argument("--pipeline", default="forward", choices=["forward", "reconstruct"],
                    help="forward = ModulatedSiren.forward on resident tiles (the metric's timed region); "
                         "reconstruct = slice -> tiles -> black filter -> forward -> weighted fold -> slice, all on the device")
    ap.add_argument("--brain-mask", action="store_true",
                    help="elliptical brain-like mask on the synthetic slices (43 %% of the tiles become black; only "
                         "--pipeline reconstruct skips them, as the reference's black-patch filter does)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU baseline sample")
    ap.add_argument("--check", action="store_true", help="also verify one batch against the oracle")
    return ap.parse_args()


def effective_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota and by
    the GPU box's per-GPU CPU share (16) -- oversubscribing MKL with 256 threads on a 16-core share
    is ~15x slower than using the share."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 16))


def traffic_bytes(kernel):
    """HBM/fabric bytes per launch of the dominant kernel from the committed PMC passes (collected with
    tools/profile.sh: separate --pmc runs, FETCH_SIZE doubled per the gfx950 correction); None if the
    profile on record is for another kernel."""
    try:
        d = json.load(open(os.path.join(REPO, "profiles", "r1", "traffic.json")))
        return d["bytes_per_launch"] if d.get("kernel") == kernel else None
    except Exception:
        return None


def cpu_baseline(sd, tiles, activation, budget_s):
    """The torch-CPU twin (oracle/torch_twin.py, "port") timed on this box's host cores."""
    import torch

    from oracle import torch_twin as tw

    cores = effective_cores()
    torch.set_num_threads(cores)
    t = tw.to_tensors(sd)
    x = torch.from_numpy(tiles)
    tw.forward_tiles(t, x[:64], num_layers=5, activation=activation)  # warm-up (baseline model only)
    n, t0 = 0, time.perf_counter()
    while True:
        tw.forward_tiles(t, x, num_layers=5, activation=activation)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    px = n * 320 * 320
    return {
        "value": px / el / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port",
        "sample": f"{n} x (one 320x320 slice = 400 tiles -> 400x24x24) through oracle/torch_twin.py "
                  f"(torch {torch.__version__} CPU, {torch.get_num_threads()} threads), {el:.1f} s",
    }


def main():
    args = parse()
    deep = args.model == "deep_residual"
    if args.precision is None:
        args.precision = "bf16" if deep else "f16x3"
    H, L, Z = (512, 10, 128) if deep else (256, 5, 256)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    from mri_inr_amd import ModulatedSiren, synthetic as syn
    from mri_inr_amd import _lib
    from mri_inr_amd.dist import broadcast_state_dict

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; MSIREN_BENCH_BACKEND=gloo lets several ranks share one card (rehearsal of the
    # N > 1 code path on a 1-GPU box: RCCL refuses two ranks on the same device)
    backend = os.environ.get("MSIREN_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- model: random-init weights of the named architecture; rank 0's copy is broadcast (RCCL) ----
    # deep residual model: modulations centred on 0.25 keep the 10-layer residual stream in the regime
    # where 16-bit operands are meaningful (with O(1) modulations it is chaotic: even fp32 is only 2e-4)
    kw = dict(modulator_bias_center=0.25, encoder_gain=10.0) if deep else dict(trained_like=True)
    sd = syn.make_state_dict(seed=7, dim_hidden=H, num_layers=L, latent_dim=Z, **kw) if rank == 0 else None
    if world > 1:
        sd = broadcast_state_dict(sd, src=0, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
    model = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24,
                           device=f"cuda:{local_rank}", activation=args.activation, precision=args.precision, residual=deep)
    model.load_state_dict(sd)
    model.to(f"cuda:{local_rank}").eval()
    lib, h = model._lib, model._h

    # ---- synthetic input: slice k = default_rng(1000+k).random((320,320)), tiled 32/16 on the device ----
    n_sl = args.slices
    imgs = np.stack([syn.make_slice(rank * n_sl + k, brain_mask=args.brain_mask) for k in range(n_sl)])
    B = n_sl * 400
    d_img = model.device_array(imgs.shape).copy_from(imgs)
    d_tiles = model.device_array((B, 32, 32))
    d_outs = [model.device_array((B, 24, 24)) for _ in range(2)]
    d_out = d_outs[0]
    d_recons = [model.device_array((n_sl, 320, 320)) for _ in range(2)] if args.pipeline == "reconstruct" else None
    _lib.check(lib.msiren_set_streams(h, args.streams))
    _lib.check(lib.msiren_image_to_patches_dev(h, d_img.ptr, n_sl, 320, 320, d_tiles.ptr))
    model.sync()

    nstep = [0]

    def step():
        # consecutive steps are independent slices: alternate the output buffer with the stream
        if d_recons is not None:
            _lib.check(lib.msiren_reconstruct_slices_dev(h, d_img.ptr, n_sl, 320, 320, d_recons[nstep[0] & 1].ptr))
        else:
            _lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_outs[nstep[0] & 1].ptr))
        nstep[0] += 1

    def fence():
        model.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    import ctypes as C

    _lib.check(lib.msiren_profile_enable(h, 1))  # HIP events around every trunk launch, on its stream
    _lib.check(lib.msiren_timer_start(h))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    model.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev_ms = C.c_float()
    _lib.check(lib.msiren_timer_stop(h, C.byref(dev_ms)))
    launches, trunk_ms = C.c_int64(), C.c_double()
    _lib.check(lib.msiren_profile_read(h, C.byref(launches), C.byref(trunk_ms)))
    _lib.check(lib.msiren_profile_enable(h, 0))
    elapsed = t1 - t0
    overlapped_trunk_ms = trunk_ms.value / max(launches.value, 1)
    n_over = int(launches.value)
    if args.streams > 1:
        # With two streams the launches of consecutive steps overlap, so a launch's own duration says
        # little about the kernel.  The roofline figure is therefore taken from a short single-stream
        # phase of the same process (kernel alone on the device), after the timed region.
        _lib.check(lib.msiren_set_streams(h, 1))
        for _ in range(3):
            step()
        model.sync()
        _lib.check(lib.msiren_profile_enable(h, 1))
        for _ in range(max(10, args.steps // 4)):
            step()
        _lib.check(lib.msiren_profile_read(h, C.byref(launches), C.byref(trunk_ms)))
        _lib.check(lib.msiren_profile_enable(h, 0))
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    px_per_step = world * n_sl * 320 * 320
    value = px_per_step * args.steps / elapsed / 1e6
    # the slice pipeline skips black tiles (mean < 1e-10, tiling.py:184-198): only evaluated tiles count as work
    evaluated = B
    if d_recons is not None:
        evaluated = int((d_tiles.numpy().reshape(B, -1).mean(axis=1, dtype=np.float32) >= np.float32(1e-10)).sum())
    flops_launch = model.flops_per_coord() * evaluated * 576
    trunk_avg_s = trunk_ms.value / max(launches.value, 1) / 1e3
    achieved = flops_launch / trunk_avg_s / 1e12

    if args.precision == "f16x3":
        # 3 fp16 MFMAs per algorithmic multiply-add: the bound for ALGORITHMIC FLOPs is the dense fp16
        # MFMA peak / 3.  The fp32-MFMA peak the north star names is reported next to it.
        dtype, peak = "f16x3", F16_MFMA_PEAK_TFLOPS / 3.0
        dtype_note = "split-fp16: hi/lo fp16 operands, three fp16 MFMAs per product, fp32 accumulate; fp32-equivalent accuracy (1e-4 gate)"
        kernel = "siren_trunk_f16x3_kernel<%d,4>" % (1 if args.activation == "morlet" else 0)
    elif args.precision in ("bf16", "f16"):
        dtype, peak = args.precision, F16_MFMA_PEAK_TFLOPS
        dtype_note = f"{args.precision} MFMA operands, fp32 accumulate"
        kernel = "siren_trunk_x1_kernel<%d,%d,%d,3>" % (args.precision == "bf16", args.activation == "morlet", deep)
    else:
        dtype, peak = "f32", FP32_MFMA_PEAK_TFLOPS
        dtype_note = "fp32 MFMA (exact fp32 products and accumulation)"
        kernel = "siren_trunk_f32_kernel<%d,%d,%d,0>" % (H, args.activation == "morlet", deep)
    result = {
        "metric": "Mpixels/sec reconstructed (320x320 slice, hidden=256, 5 layers)",
        "value": value, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "dtype_note": dtype_note, "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[{4 if deep else 1}]{' (deep residual 10x512, own semantics)' if deep else ''}: {n_sl} x 320x320 slice per GPU per step -> {B} tiles 32x32 -> "
                        f"ModulatedSiren.forward (encoder+modulator+fused trunk, {args.activation}) -> {B}x24x24; "
                        "tiles and outputs resident in HBM",
            "slices_per_gpu_per_step": n_sl, "patches_per_step_per_gpu": B, "patches_evaluated_per_step_per_gpu": evaluated,
            "coords_per_patch": 576,
            "dim_hidden": H, "num_layers": L, "latent_dim": Z, "residual": deep, "activation": args.activation, "precision": args.precision, "streams": args.streams,
            "pipeline": args.pipeline, "brain_mask": bool(args.brain_mask),
            "parallelism": f"patch-shard x{world}",
        },
        "roofline": {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic_bytes(kernel), "kernel": kernel,
            "flops_per_launch": flops_launch, "avg_launch_ms": trunk_avg_s * 1e3, "launches": int(launches.value),
            "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
            # tools/mfma_peak_probe.hip: all 256 CUs issuing only fp16 MFMAs sustain 1476 TFLOP/s under the
            # board power limit (profiles/r1/10_*); context for `frac`, which is against the nominal peak
            "sustained_fp16_mfma_tflops_measured": SUSTAINED_F16_MFMA_TFLOPS,
            "measured": "HIP event pairs on the kernel's stream; single-stream phase of this run (kernel alone)"
                        if args.streams > 1 else "HIP event pairs on the kernel's stream inside the timed region",
            "timed_region_avg_launch_ms": overlapped_trunk_ms, "timed_region_launches": n_over,
            "pipelined_tflops": flops_launch * world * args.steps / elapsed / 1e12 / world,
            "note": "achieved = algorithmic FLOPs (525824 per coordinate) / mean kernel time; for f16x3 the "
                    "kernel issues 3x that many fp16 MFMA FLOPs, hence peak = 2500/3",
        },
        "device_ms_per_step": dev_ms.value / args.steps,
    }

    if rank == 0:
        if args.check and d_recons is not None:
            from oracle import siren_oracle as orc

            ref = orc.reconstruct_slice(sd, imgs[0], num_layers=L, activation=args.activation, dtype=np.float64)
            got = d_recons[(nstep[0] - 1) & 1].numpy()[0]
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        elif args.check:
            from oracle import siren_oracle as orc

            got = d_out.numpy()[:64]
            tiles_h = d_tiles.numpy()[:64]
            z = orc.encoder_forward(sd, tiles_h, dtype=np.float64)
            mods = orc.modulator_forward(sd, z, num_layers=L, dtype=np.float64)
            ref = orc.siren_forward(sd, mods, num_layers=L, activation=args.activation, residual=deep,
                                    dtype=np.float64).reshape(-1, 24, 24)
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        if world == 1 and not args.no_cpu_baseline and not deep:
            result["cpu_baseline"] = cpu_baseline(sd, d_tiles.numpy()[:400], args.activation, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
        info = model.device_info()
        result["device"] = info["name"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
