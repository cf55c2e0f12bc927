#!/usr/bin/env python3
"""Benchmark of the MI355X modulated-SIREN path: Mpixels/s reconstructed (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            # N > 1: starts its N ranks itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input per GPU:
``ModulatedSiren.forward`` (encoder -> modulator -> fused SIREN trunk) on the 400 tiles
(32x32 in, 24x24 out) of each 320x320 slice of the batch, tiles already resident in HBM, outputs left
in HBM.  Two ways of sizing the batch:

  --slices S         S slices per GPU per step (default 1 = BASELINE.json configs[1]); per-GPU work is
                     fixed as N grows                                          -> "scaling": "weak"
  --total-slices T   a fixed batch of T slices per step (64 = BASELINE.json configs[2]), contiguous
                     slice blocks sharded over the ranks (dist.shard_range)    -> "scaling": "strong"

Patches are independent given the weights, so the path has no data-path collective: every rank
evaluates its own slices; the only collective is the broadcast of the weights from rank 0 at load
time (RCCL over xGMI), outside the timed region.  Started without a launcher and with --gpus N > 1
the script launches its own N ranks (mri_inr_amd/launch.py: one child process per GPU, torchrun's
environment contract); under torchrun it uses the environment it finds.  WORLD_SIZE != --gpus is
an error in every case.

Backends (MSIREN_BENCH_BACKEND): "rccl" (default) -- communicator, broadcast, barrier and MAX-reduce
through libmsiren's C ABI (include/msiren.h, "multi-GPU"): no torch.distributed, one HIP runtime in
the process; "nccl" -- the same over torch.distributed (RCCL); "gloo" -- torch.distributed on the CPU,
which lets several ranks share one card to rehearse the N > 1 path on a 1-GPU box.

Prints ONE JSON line on rank 0 (README / DESIGN.md §5 describe the fields).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (= vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no sparsity)
MIN_WARMUP_S = 0.4             # untimed: the card needs a few hundred ms under load before its clock settles
MIN_ROOFLINE_LAUNCHES = 200


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--slices", type=int, default=1, help="320x320 slices per GPU per step (weak scaling)")
    ap.add_argument("--total-slices", type=int, default=0,
                    help="fixed batch of this many slices per step, sharded over the ranks (strong scaling; 64 = BASELINE configs[2])")
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
    ap.add_argument("--no-extras", action="store_true", help="skip the host->host and slice->slice rates after the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU baseline sample")
    ap.add_argument("--check", action="store_true", help="also verify one batch against the oracle")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="self-launched ranks: wall limit of the whole job")
    return ap.parse_args(argv)


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
    tools/profile.sh: separate --pmc runs, FETCH_SIZE doubled per the gfx950 correction); None if no profile on
    record is for this kernel."""
    for rnd in ("r3", "r2", "r1"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", rnd, "traffic.json")))
        except Exception:
            continue
        for e in d.get("kernels", [d]):
            if e.get("kernel") == kernel:
                return e["bytes_per_launch"]
    return None


def cpu_baseline(sd, tiles, activation, budget_s):
    """The torch-CPU twin (oracle/torch_twin.py, "port") timed on this box's host cores: best of 5 full slices after
    2 warm-up slices; the wall budget only cuts the repetitions short on a very slow host."""
    import torch

    from oracle import torch_twin as tw

    cores = effective_cores()
    torch.set_num_threads(cores)
    t = tw.to_tensors(sd)
    x = torch.from_numpy(tiles)
    t_begin = time.perf_counter()
    warm = 0
    for _ in range(2):
        tw.forward_tiles(t, x, num_layers=5, activation=activation)
        warm += 1
        if time.perf_counter() - t_begin > budget_s / 2:
            break
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        tw.forward_tiles(t, x, num_layers=5, activation=activation)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > budget_s:
            break
    best = min(times)
    return {
        "value": 320 * 320 / best / 1e6, "unit": "Mpixel/s", "cores": cores, "kind": "port",
        "sample": f"best of {len(times)} after {warm} warm-up(s), each one 320x320 slice = 400 tiles -> 400x24x24 through "
                  f"oracle/torch_twin.py (torch {torch.__version__} CPU, {torch.get_num_threads()} threads); best {best:.3f} s, "
                  f"mean {sum(times) / len(times):.3f} s, {time.perf_counter() - t_begin:.1f} s in all",
    }


def workload_label(args, deep, n_total, world):
    if deep:
        cfg = "BASELINE configs[4] (deep residual 10x512, latent 128; own semantics, parity unpinned)"
    elif args.activation == "morlet":
        cfg = "BASELINE configs[3] (Morlet activation)"
    elif args.total_slices:
        cfg = f"BASELINE configs[2]: batch of {n_total} slices patch-sharded over {world} GPU(s)" if n_total == 64 else \
              f"fixed batch of {n_total} slices sharded over {world} GPU(s) (BASELINE configs[2] shape)"
    elif args.slices == 1:
        cfg = "BASELINE configs[1]: one 320x320 slice"
    else:
        cfg = f"{args.slices} x BASELINE configs[1] (320x320 slices per GPU per step)"
    return cfg


class TorchGroup:
    """torch.distributed plumbing (backends "nccl" = RCCL, "gloo"): same interface as dist.RcclGroup."""

    def __init__(self, backend, rank, world, local_rank):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.backend, self.rank, self.world = torch, dist, backend, rank, world
        self.device = torch.device("cuda", local_rank) if backend == "nccl" else None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    def broadcast_state_dict(self, sd):
        from mri_inr_amd.dist import broadcast_state_dict

        return broadcast_state_dict(sd, src=0, device=self.device) if self.world > 1 else sd

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min(self, v):
        return -self.max(-v)

    def info(self):
        return (self.dist.get_world_size(), self.dist.get_rank()) if self.world > 1 else (1, 0)

    def destroy(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main():
    args = parse()
    from mri_inr_amd import launch

    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if args.gpus > 1 and not launch.under_launcher():
        # no launcher around us: become one.  Nothing in this process has touched HIP (or imported torch).
        rc, _ = launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                   timeout=args.launch_timeout)
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py without a launcher (it starts its "
                         f"own ranks) or make the launcher's --nproc-per-node match")

    deep = args.model == "deep_residual"
    if args.precision is None:
        args.precision = "bf16" if deep else "f16x3"
    H, L, Z = (512, 10, 128) if deep else (256, 5, 256)
    backend = os.environ.get("MSIREN_BENCH_BACKEND", "rccl")
    if backend not in ("rccl", "nccl", "gloo"):
        raise SystemExit(f"MSIREN_BENCH_BACKEND={backend}: expected rccl, nccl or gloo")

    from mri_inr_amd import ModulatedSiren, synthetic as syn
    from mri_inr_amd import _lib
    from mri_inr_amd.dist import RcclGroup, shard_range

    tgroup = None
    if backend != "rccl":
        # torch's bundled HIP runtime has to come up before libmsiren's (mri_inr_amd/_lib.py)
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a gfx950 GPU (the HIP path has no CPU fallback)")
        tgroup = TorchGroup(backend, rank, world, local_rank % torch.cuda.device_count())
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a gfx950 GPU (the HIP path has no CPU fallback)")
    if backend != "gloo" and world > ndev and not os.environ.get("MSIREN_BENCH_ALLOW_SHARED"):  # (test knob: lets RCCL refuse the shared card)
        raise SystemExit(f"{world} ranks but {ndev} GPU(s) visible: one rank per GPU (MSIREN_BENCH_BACKEND=gloo rehearses "
                         f"several ranks on one card)")
    dev = local_rank % ndev

    # ---- model: random-init weights of the named architecture; rank 0's copy is broadcast (RCCL) ----
    # deep residual model: modulations centred on 0.25 keep the 10-layer residual stream in the regime
    # where 16-bit operands are meaningful (with O(1) modulations it is chaotic: even fp32 is only 2e-4)
    kw = dict(modulator_bias_center=0.25, encoder_gain=10.0) if deep else dict(trained_like=True)
    sd = syn.make_state_dict(seed=7, dim_hidden=H, num_layers=L, latent_dim=Z, **kw) if rank == 0 else None
    model = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24,
                           device=f"cuda:{dev}", activation=args.activation, precision=args.precision, residual=deep)
    if tgroup is not None:
        sd = tgroup.broadcast_state_dict(sd)
        model.load_state_dict(sd)
        model.to(f"cuda:{dev}").eval()
        group = tgroup
    else:
        try:
            group = RcclGroup(model)      # communicator through the C ABI (no-op for one rank)
        except Exception as exc:  # noqa: BLE001 -- keep the measurement alive if RCCL cannot come up on this node
            if world == 1:
                raise
            # The bootstrap is collective: it fails on every rank (or rank 0's status blob says it could not make the
            # unique id), so all ranks take this branch together.  Fallback, said out loud in the JSON line
            # ("collective_fallback": true): the weights are synthetic and seeded, so every rank builds its own copy
            # (the broadcast is outside the timed region anyway); barrier and MAX go over gloo on the CPU.  No
            # torch.cuda call is made: the process keeps its one HIP runtime.
            print(f"[rank {rank}] RCCL through the C ABI failed ({exc}); falling back to seeded weights + gloo",
                  file=sys.stderr, flush=True)
            backend = "gloo-fallback"
            group = TorchGroup("gloo", rank, world, dev)
            sd = syn.make_state_dict(seed=7, dim_hidden=H, num_layers=L, latent_dim=Z, **kw)
            model.load_state_dict(sd)
            model.to(f"cuda:{dev}")
        else:
            # load_state_dict on rank 0 only (sd is None elsewhere); ONE ncclBroadcast of the blob.  A failure past the
            # bootstrap is fatal on purpose: a rank that fell back alone would leave the others inside the collective.
            group.broadcast_weights(0, sd)
        model.eval()
    lib, h = model._lib, model._h
    comm_ranks, comm_rank = group.info()
    if world > 1 and (comm_ranks, comm_rank) != (world, rank):
        raise SystemExit(f"[rank {rank}] the communicator reports rank {comm_rank} of {comm_ranks}, the launcher {rank} of {world}")
    # Every rank evaluates the SAME 16 tiles with the weights it ended up with; the bit patterns of the outputs are
    # summed and MAX / MIN-reduced: equal <=> every rank holds the weights rank 0 loaded (a wrong blob on a receiving
    # rank would otherwise be silent -- nothing else in the benchmark compares ranks).
    probe_tiles = np.random.default_rng(4242).random((16, 32, 32), dtype=np.float32)
    probe_sum = float(np.ascontiguousarray(model(probe_tiles)).view(np.uint32).astype(np.float64).sum())
    sum_max, sum_min = group.max(probe_sum), group.min(probe_sum)
    if sum_max != sum_min:
        raise SystemExit(f"[rank {rank}] ranks disagree on the probe output (checksum {probe_sum:.0f}, max {sum_max:.0f}, "
                         f"min {sum_min:.0f}): the weight broadcast did not replicate rank 0's state_dict")

    # ---- synthetic input: slice k = default_rng(1000+k).random((320,320)), tiled 32/16 on the device ----
    if args.total_slices:
        lo, hi = shard_range(args.total_slices, rank, world)
        n_total, scaling = args.total_slices, "strong"
    else:
        lo, hi = rank * args.slices, (rank + 1) * args.slices
        n_total, scaling = world * args.slices, "weak"
    n_sl = hi - lo
    B = n_sl * 400
    imgs = np.stack([syn.make_slice(k, brain_mask=args.brain_mask) for k in range(lo, hi)]) if n_sl else \
        np.zeros((0, 320, 320), np.float32)
    d_img = model.device_array((max(n_sl, 1), 320, 320))
    d_tiles = model.device_array((max(B, 1), 32, 32))
    d_outs = [model.device_array((max(B, 1), 24, 24)) for _ in range(2)]
    d_recons = [model.device_array((max(n_sl, 1), 320, 320)) for _ in range(2)]
    if n_sl:
        _lib.check(lib.msiren_memcpy_h2d(h, d_img.ptr, imgs.ctypes.data, imgs.nbytes))
    _lib.check(lib.msiren_set_streams(h, args.streams))
    _lib.check(lib.msiren_image_to_patches_dev(h, d_img.ptr, n_sl, 320, 320, d_tiles.ptr))
    model.sync()

    nstep = [0]

    def step_forward():
        # consecutive steps are independent batches: alternate the output buffer with the stream
        _lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_outs[nstep[0] & 1].ptr))
        nstep[0] += 1

    def step_reconstruct():
        _lib.check(lib.msiren_reconstruct_slices_dev(h, d_img.ptr, n_sl, 320, 320, d_recons[nstep[0] & 1].ptr))
        nstep[0] += 1

    step = step_reconstruct if args.pipeline == "reconstruct" else step_forward

    def device_sync():
        model.sync()  # hipStreamSynchronize on every stream the handle launches on
        if tgroup is not None:
            tgroup.torch.cuda.synchronize()

    def fence():
        device_sync()
        group.barrier()
        device_sync()

    # ---- warm-up: W steps as asked, then (still untimed) until the card has been under load for MIN_WARMUP_S ----
    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    warm_steps = args.warmup
    device_sync()
    while time.perf_counter() - t_w < MIN_WARMUP_S:
        for _ in range(16):
            step()
        warm_steps += 16
        device_sync()
    fence()

    _lib.check(lib.msiren_profile_enable(h, 1))  # HIP events around every trunk launch, on its stream
    _lib.check(lib.msiren_timer_start(h))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_sync()
    t1 = time.perf_counter()
    dev_ms = C.c_float()
    _lib.check(lib.msiren_timer_stop(h, C.byref(dev_ms)))
    launches, trunk_ms = C.c_int64(), C.c_double()
    _lib.check(lib.msiren_profile_read(h, C.byref(launches), C.byref(trunk_ms)))
    _lib.check(lib.msiren_profile_enable(h, 0))
    elapsed_local = t1 - t0
    group.barrier()
    elapsed = group.max(elapsed_local)
    overlapped_trunk_ms = trunk_ms.value / max(launches.value, 1)
    n_over = int(launches.value)

    if args.streams > 1 and B > 0:
        # With two streams the launches of consecutive steps overlap, so a launch's own duration says
        # little about the kernel.  The roofline figure is therefore taken from a single-stream phase of
        # the same process (kernel alone on the device), after the timed region.
        _lib.check(lib.msiren_set_streams(h, 1))
        for _ in range(8):
            step()
        model.sync()
        _lib.check(lib.msiren_profile_enable(h, 1))
        for _ in range(max(MIN_ROOFLINE_LAUNCHES, args.steps // 4) if n_sl <= 8 else max(20, args.steps // 4)):
            step()
        _lib.check(lib.msiren_profile_read(h, C.byref(launches), C.byref(trunk_ms)))
        _lib.check(lib.msiren_profile_enable(h, 0))
        _lib.check(lib.msiren_set_streams(h, args.streams))

    # behind the timed region and the roofline phase (the card is warm): the power-limited MFMA ceiling of this card
    sustained_tflops = sustained_mhz = None
    if rank == 0 and not args.no_extras:
        t_, m_ = C.c_double(), C.c_double()
        _lib.check(lib.msiren_mfma_sustained_probe(h, C.byref(t_), C.byref(m_)))
        sustained_tflops, sustained_mhz = t_.value, m_.value

    px_per_step = n_total * 320 * 320
    value = px_per_step * args.steps / elapsed / 1e6
    # the slice pipeline skips black tiles (mean < 1e-10, tiling.py:184-198): only evaluated tiles count as work
    evaluated = B
    if args.pipeline == "reconstruct" and B:
        evaluated = int((d_tiles.numpy()[:B].reshape(B, -1).mean(axis=1, dtype=np.float32) >= np.float32(1e-10)).sum())
    flops_launch = model.flops_per_coord() * evaluated * 576
    trunk_avg_s = trunk_ms.value / max(launches.value, 1) / 1e3
    achieved = flops_launch / trunk_avg_s / 1e12 if trunk_avg_s > 0 else 0.0

    if args.precision == "f16x3":
        # 3 fp16 MFMAs per algorithmic multiply-add: the bound for ALGORITHMIC FLOPs is the dense fp16
        # MFMA peak / 3.  The fp32-MFMA peak the north star names is reported next to it.
        dtype, peak = "f16x3", F16_MFMA_PEAK_TFLOPS / 3.0
        dtype_note = "split-fp16: hi/lo fp16 operands, three fp16 MFMAs per product, fp32 accumulate; fp32-equivalent accuracy (1e-4 gate)"
        act_i = 1 if args.activation == "morlet" else 0
        if os.environ.get("MSIREN_F16_TILE") == "32":   # A/B build (make AB32=1) with the 32x32x16 kernel selected
            kernel = "siren_trunk_f16x3_kernel<%d,4>" % act_i
        elif 3 <= L <= 5 and os.environ.get("MSIREN_F16_WS", "1") != "0":
            # single-stream launches (the roofline phase) run the weight-stationary trunk; with two streams the timed region
            # runs the register-resident one beside the next call's encoder / modulator (roofline_timed_mode.kernel)
            kernel = "siren_trunk_f16x3w_kernel<%d,4>" % act_i
        else:                                            # 16x16x32 tiles; the num_layers = 5 straight-line instance
            kernel = "siren_trunk_f16x3n_kernel<%d,4,%d>" % (act_i, 5 if L == 5 else 0)
        kernel_two_streams = "siren_trunk_f16x3n_kernel<%d,3,%d>" % (act_i, 5 if L == 5 else 0)
    elif args.precision in ("bf16", "f16"):
        dtype, peak = args.precision, F16_MFMA_PEAK_TFLOPS
        dtype_note = f"{args.precision} MFMA operands, fp32 accumulate"
        kernel = "siren_trunk_x1_kernel<%d,%d,%d,3>" % (args.precision == "bf16", args.activation == "morlet", deep)
    else:
        dtype, peak = "f32", FP32_MFMA_PEAK_TFLOPS
        dtype_note = "fp32 MFMA (exact fp32 products and accumulation)"
        kernel = "siren_trunk_f32_kernel<%d,%d,%d,0>" % (H, args.activation == "morlet", deep)
    if dtype != "f16x3":
        kernel_two_streams = kernel
    stage = ("slice -> tiles -> black filter -> encoder+modulator+fused trunk -> weighted fold -> slice, device-resident"
             if args.pipeline == "reconstruct" else
             f"ModulatedSiren.forward (encoder+modulator+fused trunk, {args.activation}) on resident tiles -> (B,24,24) in HBM")
    result = {
        "metric": f"Mpixels/sec reconstructed (320x320 slice, hidden={H}, {L} layers)",
        "value": value, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": dtype, "dtype_note": dtype_note, "data": "synthetic",
        # true = RCCL could not come up on this node and the ranks fell back to seeded weights + gloo plumbing
        "collective_fallback": backend == "gloo-fallback",
        "config": {
            "workload": f"{workload_label(args, deep, n_total, world)}; per step {n_total} slice(s) = {n_total * 400} tiles 32x32 "
                        f"over {world} GPU(s); {stage}",
            "slices_per_step_total": n_total, "slices_per_step_rank0": n_sl, "patches_per_step_rank0": B,
            "patches_evaluated_per_step_rank0": evaluated, "coords_per_patch": 576,
            "dim_hidden": H, "num_layers": L, "latent_dim": Z, "residual": deep, "activation": args.activation,
            "precision": args.precision, "streams": args.streams, "pipeline": args.pipeline, "brain_mask": bool(args.brain_mask),
            "parallelism": f"slice-shard x{world} ({scaling}; no data-path collective, one weight broadcast at load)",
            "backend": {"rccl": "RCCL through libmsiren's C ABI (torch-free)", "nccl": "torch.distributed nccl (RCCL)",
                        "gloo": "torch.distributed gloo (rehearsal: ranks may share a card)",
                        "gloo-fallback": "RCCL init failed on this node: seeded weights on every rank, barrier/MAX over gloo"}[backend]
                       if world > 1 else "single process",
            # the communicator as the library reports it (msiren_comm_info / torch.distributed), the collective
            # library behind it, and the cross-rank check of the replicated weights (see above)
            "rccl_ranks": comm_ranks if backend in ("rccl", "nccl") else None,
            "comm_ranks": comm_ranks,
            "rccl_lib": (os.environ.get("MSIREN_RCCL_LIB") or "system librccl") if backend == "rccl" and world > 1 else None,
            "ranks_hold_identical_weights": bool(sum_max == sum_min), "probe_checksum": probe_sum,
            "warmup_steps_run": warm_steps,
        },
        "roofline": {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic_bytes(kernel), "kernel": kernel,
            # achieved HBM/fabric rate of the kernel, to show how far from the 8 TB/s roof it is
            # (only where the committed PMC profile applies: a full 400-tile single-slice launch of the forward pipeline)
            "hbm_gb_s": (traffic_bytes(kernel) / trunk_avg_s / 1e9)
                        if (traffic_bytes(kernel) and trunk_avg_s > 0 and n_sl == 1 and evaluated == 400 and args.pipeline == "forward") else None,
            "hbm_peak_gb_s": 8000.0,
            "flops_per_launch": flops_launch, "avg_launch_ms": trunk_avg_s * 1e3, "launches": int(launches.value),
            "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
            # what this card sustains on nothing but the f16x3 trunk's MFMA stream with operands of the trunk's magnitudes,
            # measured in this run behind the timed region (msiren_mfma_sustained_probe; round 1's standalone probe gave
            # 1476 TFLOP/s): the power-limited ceiling -- context for `frac`, which is against the NOMINAL peak
            "sustained_fp16_mfma_tflops_measured": sustained_tflops,
            "sustained_mfma_clock_mhz_equivalent": sustained_mhz,
            "frac_of_sustained_mfma_rate": (achieved / (sustained_tflops / 3.0)) if (sustained_tflops and dtype == "f16x3") else None,
            "measured": ("HIP event pairs on the kernel's stream; single-stream phase of this run (kernel alone: what a handle with "
                         "one stream launches), rank 0; the two-stream timed region launches " + kernel_two_streams +
                         " (roofline_timed_mode)") if args.streams > 1 else
                        "HIP event pairs on the kernel's stream inside the timed region, rank 0",
            "timed_region_avg_launch_ms": overlapped_trunk_ms, "timed_region_launches": n_over,
            "pipelined_tflops_per_gpu": model.flops_per_coord() * 576 * 400 * n_total * args.steps / elapsed / 1e12 / world,
            "note": "achieved = algorithmic FLOPs (525824 per coordinate at 256x5) / mean kernel time; for f16x3 the "
                    "kernel issues 3x that many fp16 MFMA FLOPs, hence peak = 2500/3",
        },
        "device_ms_per_step": dev_ms.value / args.steps,
    }
    # The same figure for the mode `value` is measured in: with two streams the trunk launches of consecutive steps
    # overlap, so what describes the timed region is the rate at which whole slices leave the pipeline (encoder and
    # modulator included in the time, only the trunk's algorithmic FLOPs counted), not one launch's duration.
    tm_achieved = result["roofline"]["pipelined_tflops_per_gpu"]
    result["roofline_timed_mode"] = {
        "bound": "mfma", "achieved": tm_achieved, "peak": peak, "unit": "TFLOP/s", "frac": tm_achieved / peak,
        "streams": args.streams, "kernel": kernel_two_streams if args.streams > 1 and dtype == "f16x3" else kernel,
        "measured": "trunk FLOPs of the timed region / its wall time (MAX over ranks), per GPU: the figure consistent with `value`",
    }

    if rank == 0:
        if args.check and args.pipeline == "reconstruct":
            from oracle import siren_oracle as orc

            ref = orc.reconstruct_slice(sd, imgs[0], num_layers=L, activation=args.activation, dtype=np.float64)
            got = d_recons[(nstep[0] - 1) & 1].numpy()[0]
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        elif args.check:
            from oracle import siren_oracle as orc

            got = d_outs[(nstep[0] - 1) & 1].numpy()[:64]
            tiles_h = d_tiles.numpy()[:64]
            z = orc.encoder_forward(sd, tiles_h, dtype=np.float64)
            mods = orc.modulator_forward(sd, z, num_layers=L, dtype=np.float64)
            ref = orc.siren_forward(sd, mods, num_layers=L, activation=args.activation, residual=deep,
                                    dtype=np.float64).reshape(-1, 24, 24)
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        if world == 1 and not args.no_extras and not deep and n_sl >= 1:
            result["extra"] = extras(model, lib, h, _lib, d_img, d_tiles, d_recons, args.streams)
        if world == 1 and not args.no_cpu_baseline and not deep:
            result["cpu_baseline"] = cpu_baseline(sd, d_tiles.numpy()[:400], args.activation, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
        info = model.device_info()
        result["device"] = info["name"]
        print(json.dumps(result), flush=True)
    group.barrier()
    group.destroy()


def extras(model, lib, h, _lib, d_img, d_tiles, d_recons, streams, budget_s=0.5):
    """Rates of the two regions either side of the metric's, measured in this process after the timed region on ONE
    slice (never `value`):
      host_to_host_mpixel_s   numpy tiles on the host -> numpy (400,24,24) on the host through msiren_forward_tiles:
                              SURVEY.md §8(d)'s drop-in call, PCIe copies included (error.py:233-258 is the region);
      reconstruct_mpixel_s    slice -> tiles -> black filter -> forward -> weighted fold -> slice, device-resident
                              (msiren_reconstruct_slices_dev): "pixels reconstructed" in the literal sense."""
    tiles = d_tiles.numpy()[:400]
    model(tiles)  # workspaces
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        model(tiles)
        n += 1
    h2h = n * 320 * 320 / (time.perf_counter() - t0) / 1e6
    _lib.check(lib.msiren_set_streams(h, streams))
    k = [0]

    def rstep():
        _lib.check(lib.msiren_reconstruct_slices_dev(h, d_img.ptr, 1, 320, 320, d_recons[k[0] & 1].ptr))
        k[0] += 1

    for _ in range(20):
        rstep()
    model.sync()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(50):
            rstep()
        model.sync()
        n += 50
    rec = n * 320 * 320 / (time.perf_counter() - t0) / 1e6
    return {"host_to_host_mpixel_s": h2h, "reconstruct_mpixel_s": rec,
            "note": "one 320x320 slice per call, after the timed region: host numpy -> host numpy through "
                    "msiren_forward_tiles (PCIe-inclusive), and the device-resident slice -> slice pipeline"}


if __name__ == "__main__":
    main()
