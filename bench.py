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
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2, 3],
                    help="2 (default) = consecutive steps alternate between two HIP streams (independent slices overlap); 3 (default for "
                         "--model deep_residual) = rotate over three: a trunk that owns its CUs packs better when the next call's prologue "
                         "does not queue behind it (config 5: +3.7 %%; the default model: +0.1 %%)")
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
    ap.add_argument("--numa-pin", action="store_true",
                    help="pin each rank to the CPUs of its GPU's NUMA node before it touches HIP (mri_inr_amd/launch.py; off by default, "
                         "reported in config.ranks[]; untested on a multi-socket node)")
    ap.add_argument("--torch-first", action="store_true",
                    help="import torch and bring its bundled HIP runtime up BEFORE libmsiren is loaded: what the reference's own host program "
                         "does (test_mod_siren.py imports torch at the top).  libmsiren then runs on torch's bundled runtime instead of the "
                         "system one (same soname: msiren_runtime_info; config.hip_runtime names the file that is mapped)")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1 without --slices / --total-slices: skip the config-3 strong-scaling region behind the weak one")
    ap.add_argument("--scaling-selftest", action="store_true",
                    help="run the N = 1 measurement twice -- plainly, and through the launcher path (one rank with "
                         "RANK / WORLD_SIZE=1 / MASTER_* set, what a scaling sweep's N = 1 point goes through) -- and fail "
                         "unless the two values agree within 3 %%")
    args = ap.parse_args(argv)
    if args.streams is None:
        args.streams = 3 if args.model == "deep_residual" else 2
    return args


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
    """HBM/fabric bytes per launch of a trunk instance from the committed PMC passes (collected with tools/profile.sh:
    separate --pmc runs, FETCH_SIZE doubled per the gfx950 correction): the newest profile on record that names this
    instance, as {"bytes_per_launch", "source"}; None if there is none."""
    if not kernel:
        return None
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        try:
            d = json.load(open(os.path.join(REPO, "profiles", rnd, "traffic.json")))
        except Exception:
            continue
        for e in d.get("kernels", [d]):
            if e.get("kernel") == kernel:
                return {"bytes_per_launch": e["bytes_per_launch"], "source": e.get("source", f"profiles/{rnd}/traffic.json")}
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

    def max_array(self, values):
        if self.world == 1:
            return [float(x) for x in values]
        t = self.torch.tensor([float(x) for x in values], dtype=self.torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

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
    if args.scaling_selftest:
        sys.exit(scaling_selftest(args))
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

    # optional: CPU affinity by the GPU's NUMA node -- BEFORE anything starts the HIP runtime's threads
    numa = launch.pin_rank(local_rank) if args.numa_pin else None

    if args.torch_first:
        # the reference's host program imports torch first; libmsiren's hip* calls then bind to torch's bundled libamdhip64 (same soname)
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()

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

    # Who sits where: every rank fills its own row of a (world x 9) table -- device index, PCI bus id (domain, bus, device,
    # function), CUs, clock, HBM, what its communicator says its rank is -- and one MAX-reduce hands rank 0 all of it
    # (config.ranks[]): the first run on a real 8-GPU node can be read without a second one.
    info0 = model.device_info()
    try:
        dom, bus, devfn = info0["pci_bus_id"].split(":")
        pci = [int(dom, 16), int(bus, 16), int(devfn.split(".")[0], 16), int(devfn.split(".")[1], 16)]
    except Exception:  # noqa: BLE001
        pci = [-1, -1, -1, -1]
    NF = 11
    table = [-1.0] * (world * NF)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1
    table[rank * NF:(rank + 1) * NF] = [float(dev)] + [float(x) for x in pci] + \
        [float(info0["compute_units"]), float(info0["clock_mhz"]), float(info0["hbm_bytes"] >> 20), float(comm_rank),
         float(numa["numa_node"]) if numa else -1.0, float(ncpu)]
    table = group.max_array(table)
    ranks_info = []
    for r in range(world):
        row = table[r * NF:(r + 1) * NF]
        ranks_info.append({"rank": r, "device": int(row[0]),
                           "pci_bus_id": "%04x:%02x:%02x.%x" % tuple(int(x) for x in row[1:5]) if row[1] >= 0 else None,
                           "compute_units": int(row[5]), "clock_mhz": int(row[6]), "hbm_mib": int(row[7]), "comm_rank": int(row[8]),
                           # --numa-pin: the NUMA node the rank was pinned to (null: not asked for, or unknown) and the CPUs it may run on
                           "numa_node": int(row[9]) if row[9] >= 0 else None, "cpus_allowed": int(row[10])})
    if world > 1 and len({(x["pci_bus_id"]) for x in ranks_info}) < world and backend != "gloo" and not os.environ.get("MSIREN_BENCH_ALLOW_SHARED"):
        raise SystemExit(f"[rank {rank}] two ranks report the same PCI device: {ranks_info}")

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
    d_outs = [model.device_array((max(B, 1), 24, 24)) for _ in range(max(2, args.streams))]
    d_recons = [model.device_array((max(n_sl, 1), 320, 320)) for _ in range(max(2, args.streams))]
    if n_sl:
        _lib.check(lib.msiren_memcpy_h2d(h, d_img.ptr, imgs.ctypes.data, imgs.nbytes))
    _lib.check(lib.msiren_set_streams(h, args.streams))
    _lib.check(lib.msiren_image_to_patches_dev(h, d_img.ptr, n_sl, 320, 320, d_tiles.ptr))
    model.sync()

    nstep = [0]

    def step_forward():
        # consecutive steps are independent batches: alternate the output buffer with the stream
        _lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_outs[nstep[0] % len(d_outs)].ptr))
        nstep[0] += 1

    def step_reconstruct():
        _lib.check(lib.msiren_reconstruct_slices_dev(h, d_img.ptr, n_sl, 320, 320, d_recons[nstep[0] % len(d_recons)].ptr))
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
    timed_kernels = model.profile_kernels()   # what the timed region launched, as the library names it
    _lib.check(lib.msiren_profile_enable(h, 0))
    elapsed_local = t1 - t0
    group.barrier()
    elapsed = group.max(elapsed_local)

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
        alone_kernels = model.profile_kernels()
        _lib.check(lib.msiren_profile_enable(h, 0))
        _lib.check(lib.msiren_set_streams(h, args.streams))
    else:
        alone_kernels = timed_kernels

    # behind the timed region and the roofline phase (the card is warm): the power-limited MFMA ceiling of this card
    sustained_tflops = sustained_mhz = None
    if rank == 0 and not args.no_extras:
        t_, m_ = C.c_double(), C.c_double()
        _lib.check(lib.msiren_mfma_sustained_probe(h, C.byref(t_), C.byref(m_)))
        sustained_tflops, sustained_mhz = t_.value, m_.value

    # N > 1 as the driver starts it (no --slices / --total-slices): the weak one-slice-per-rank region above stays `value` (so that N = 1
    # equals the BENCH line), and BASELINE configs[2] -- the fixed batch of 64 slices sharded over the ranks, the workload the north star's
    # ">= 0.9 linear" is about -- is measured behind it, by all ranks, in the same process group.
    strong = None
    default_sizing = not args.total_slices and args.slices == 1
    if world > 1 and default_sizing and not args.no_strong and not deep and args.pipeline == "forward" and not args.brain_mask:
        strong = strong_config3(model, group, lib, h, _lib, syn, shard_range, rank, world, args, device_sync, fence, backend)

    px_per_step = n_total * 320 * 320
    value = px_per_step * args.steps / elapsed / 1e6
    # the slice pipeline skips black tiles (mean < 1e-10, tiling.py:184-198): only evaluated tiles count as work
    evaluated = B
    if args.pipeline == "reconstruct" and B:
        evaluated = int((d_tiles.numpy()[:B].reshape(B, -1).mean(axis=1, dtype=np.float32) >= np.float32(1e-10)).sum())
    fpc = model.flops_per_coord()
    flops_step = fpc * evaluated * 576          # algorithmic trunk FLOPs of one step on this rank

    if args.precision == "f16x3":
        # 3 fp16 MFMAs per algorithmic multiply-add: the bound for ALGORITHMIC FLOPs is the dense fp16
        # MFMA peak / 3.  The fp32-MFMA peak the north star names is reported next to it.
        dtype, peak = "f16x3", F16_MFMA_PEAK_TFLOPS / 3.0
        dtype_note = "split-fp16: hi/lo fp16 operands, three fp16 MFMAs per product, fp32 accumulate; fp32-equivalent accuracy (1e-4 gate)"
    elif args.precision in ("bf16", "f16"):
        dtype, peak = args.precision, F16_MFMA_PEAK_TFLOPS
        dtype_note = f"{args.precision} MFMA operands, fp32 accumulate"
    else:
        dtype, peak = "f32", FP32_MFMA_PEAK_TFLOPS
        dtype_note = "fp32 MFMA (exact fp32 products and accumulation)"

    def per_kernel(recs, masked_scale=1.0):
        """[{kernel, launches, ms_total, coords}] (msiren_profile_read_kernel) -> the same with rates.  `coords` counts the
        coordinates the launches were handed; in the masked slice pipeline only `evaluated / B` of them are evaluated."""
        out = []
        for r in recs:
            fl = fpc * r["coords"] * masked_scale
            tf = fl / (r["ms_total"] * 1e-3) / 1e12 if r["ms_total"] > 0 else 0.0
            out.append({"kernel": r["kernel"], "launches": r["launches"], "avg_launch_ms": r["ms_total"] / max(r["launches"], 1),
                        "flops_per_launch": fl / max(r["launches"], 1), "achieved": tf, "frac": tf / peak})
        return out

    mscale = (evaluated / B) if B else 1.0
    timed_k, alone_k = per_kernel(timed_kernels, mscale), per_kernel(alone_kernels, mscale)
    dom_timed = max(timed_k, key=lambda r: r["flops_per_launch"] * r["launches"]) if timed_k else None
    dom_alone = max(alone_k, key=lambda r: r["flops_per_launch"] * r["launches"]) if alone_k else None
    pipelined = flops_step * args.steps / elapsed_local / 1e12 if elapsed_local > 0 else 0.0   # this rank's trunk FLOPs / its timed wall

    stage = ("slice -> tiles -> black filter -> encoder+modulator+fused trunk -> weighted fold -> slice, device-resident"
             if args.pipeline == "reconstruct" else
             f"ModulatedSiren.forward (encoder+modulator+fused trunk, {args.activation}) on resident tiles -> (B,24,24) in HBM")
    # The roofline block describes the kernel that RAN IN THE TIMED REGION (named by the library, msiren_profile_read_kernel).
    #   one stream : the launches do not overlap -> algorithmic FLOPs of that instance's launches / their summed HIP-event time;
    #   two streams: the launches of consecutive steps overlap (each event pair spans ~2 steps), so a launch's own duration
    #                stops describing the kernel: achieved = the trunk FLOPs of the timed region / its wall time (encoder +
    #                modulator of the other stream run beside the trunk and are inside that time; only trunk FLOPs counted),
    #                avg_launch_ms = wall / launches; the event mean is kept beside it (it is what rocprofv3 --kernel-trace
    #                shows for the overlapped launches).  roofline_kernel_alone: the single-stream phase behind the region.
    if dom_timed is None:
        roof = {"bound": "mfma", "achieved": 0.0, "peak": peak, "unit": "TFLOP/s", "frac": 0.0, "traffic": None, "kernel": None}
    elif args.streams == 1:
        roof = {"bound": "mfma", "achieved": dom_timed["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dom_timed["frac"],
                "kernel": dom_timed["kernel"], "flops_per_launch": dom_timed["flops_per_launch"],
                "avg_launch_ms": dom_timed["avg_launch_ms"], "launches": dom_timed["launches"],
                "measured": "HIP event pairs on the kernel's stream around every launch of this instance inside the timed region "
                            "(one stream: launches do not overlap), rank 0"}
    else:
        n_l = sum(r["launches"] for r in timed_k)
        roof = {"bound": "mfma", "achieved": pipelined, "peak": peak, "unit": "TFLOP/s", "frac": pipelined / peak,
                "kernel": dom_timed["kernel"], "flops_per_launch": flops_step * args.steps / max(n_l, 1),
                "avg_launch_ms": elapsed_local * 1e3 / max(n_l, 1), "launches": n_l,
                "event_avg_launch_ms": dom_timed["avg_launch_ms"],
                "measured": "two streams: consecutive steps' launches overlap, so achieved = trunk FLOPs of the timed region / its "
                            "wall time and avg_launch_ms = wall / trunk launches (the figure consistent with `value`); "
                            "event_avg_launch_ms = mean HIP-event span of one (overlapped) launch of this instance"}
    tb = traffic_bytes(roof.get("kernel"))
    roof.update({
        "traffic": tb["bytes_per_launch"] if tb else None,
        "traffic_source": (tb["source"] + " (rocprofv3 --pmc passes of this command in separate runs, FETCH_SIZE x2 per the gfx950 "
                           "correction; one 320x320 slice per launch -- not measured in this run)") if tb else None,
        # achieved fabric / HBM rate, to show how far from the 8 TB/s roof the kernel is; only where the committed
        # PMC profile applies: a 400-tile single-slice launch of the forward pipeline
        "hbm_gb_s": (tb["bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9)
                    if (tb and roof.get("avg_launch_ms") and n_sl == 1 and evaluated == 400 and args.pipeline == "forward") else None,
        "hbm_peak_gb_s": 8000.0,
        "frac_of_fp32_mfma_peak": roof["achieved"] / FP32_MFMA_PEAK_TFLOPS,
        "timed_region_kernels": timed_k,
        # what this card sustains on nothing but the f16x3 trunk's MFMA stream with operands of the trunk's magnitudes, measured
        # in this run behind the timed region (msiren_mfma_sustained_probe): the power-limited ceiling -- context for `frac`,
        # which is against the NOMINAL peak
        "sustained_fp16_mfma_tflops_measured": sustained_tflops,
        "sustained_mfma_clock_mhz_equivalent": sustained_mhz,
        "frac_of_sustained_mfma_rate": (roof["achieved"] / (sustained_tflops / 3.0)) if (sustained_tflops and dtype == "f16x3") else None,
        "note": f"achieved = algorithmic FLOPs ({fpc:.0f} per coordinate at {H}x{L}) / time; for f16x3 the kernel issues 3x that "
                "many fp16 MFMA FLOPs, hence peak = 2500/3",
    })
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
            "ranks": ranks_info,
            "warmup_steps_run": warm_steps,
            # which HIP runtime libmsiren's calls ran on in this process (msiren_runtime_info): the system one in a torch-free process,
            # torch's bundled one behind --torch-first / the torch backends
            "hip_runtime": _lib.runtime_info(),
            "torch_first": bool(args.torch_first or tgroup is not None),
            # (flat copies: records that keep only the scalars of `config` still carry them)
            "hip_runtime_version": _lib.runtime_info()["hip_runtime_version"], "libamdhip64": _lib.runtime_info()["libamdhip64"],
        },
        "roofline": roof,
        "device_ms_per_step": dev_ms.value / args.steps,
    }
    if args.streams > 1 and dom_alone is not None:
        # the dominant trunk instance of a ONE-stream handle on the same batch, alone on the device (single-stream phase of this
        # process, behind the timed region): kernel quality without the pipelining
        result["roofline_kernel_alone"] = {
            "bound": "mfma", "achieved": dom_alone["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dom_alone["frac"],
            "kernel": dom_alone["kernel"], "flops_per_launch": dom_alone["flops_per_launch"],
            "avg_launch_ms": dom_alone["avg_launch_ms"], "launches": dom_alone["launches"], "kernels": alone_k,
            "frac_of_sustained_mfma_rate": (dom_alone["achieved"] / (sustained_tflops / 3.0)) if (sustained_tflops and dtype == "f16x3") else None,
            "measured": "HIP event pairs on the kernel's stream, msiren_set_streams(h, 1) phase behind the timed region, rank 0",
        }

    if strong is not None:
        # the scaling claim: config 3 strong.  In `config` as well, which the driver's record keeps whole.
        result.setdefault("extra", {}).setdefault("configs", {})["config3_64_slices_strong"] = strong
        result["config"]["also_measured"] = {"config3_64_slices_strong": {k: strong[k] for k in
                                             ("value", "unit", "ms_per_step", "slices_per_rank", "efficiency_vs_n1", "n1_reference_value", "rccl_ranks")}}
        result["config"].update({"config3_strong_mpixel_s": strong["value"], "config3_strong_ms_per_step": strong["ms_per_step"],
                                 "config3_strong_efficiency_vs_n1": strong["efficiency_vs_n1"], "config3_n1_reference_mpixel_s": strong["n1_reference_value"]})

    if rank == 0:
        if args.check and args.pipeline == "reconstruct":
            from oracle import siren_oracle as orc

            ref = orc.reconstruct_slice(sd, imgs[0], num_layers=L, activation=args.activation, dtype=np.float64)
            got = d_recons[(nstep[0] - 1) % len(d_recons)].numpy()[0]
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        elif args.check:
            from oracle import siren_oracle as orc

            got = d_outs[(nstep[0] - 1) % len(d_outs)].numpy()[:64]
            tiles_h = d_tiles.numpy()[:64]
            z = orc.encoder_forward(sd, tiles_h, dtype=np.float64)
            mods = orc.modulator_forward(sd, z, num_layers=L, dtype=np.float64)
            ref = orc.siren_forward(sd, mods, num_layers=L, activation=args.activation, residual=deep,
                                    dtype=np.float64).reshape(-1, 24, 24)
            result["check_nerr_vs_fp64_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
        tiles400 = d_tiles.numpy()[:400] if (world == 1 and not args.no_cpu_baseline and not deep) else None
        if world == 1 and not args.no_extras and not deep and n_sl >= 1:
            result["extra"] = extras(model, lib, h, _lib, d_img, d_tiles, d_recons, args.streams)
            # SURVEY.md section 8(d)'s PRIMARY timed region -- the drop-in call, tiles on the host -> (B,24,24) on the host, PCIe included
            # (error.py:233-258) -- where the driver's record keeps it: a top-level key, and inside `config` (kept whole).  Never `value`.
            ex = result["extra"]
            result["host_to_host"] = {
                "value": ex["host_to_host_mpixel_s"], "unit": "Mpixel/s", "page_locked_tiles": ex["host_to_host_pinned_mpixel_s"],
                "eight_slices_per_call": ex["host_to_host_8_slices_mpixel_s"], "slice_to_slice": ex["host_slice_to_slice_mpixel_s"],
                "what": "numpy tiles (400,32,32) on the host -> numpy (400,24,24) on the host through msiren_forward_tiles, one 320x320 slice per "
                        "synchronous call, one stream, PCIe copies inside; outputs from the mirror's page-locked pool (its default)"}
            result["config"].setdefault("also_measured", {})["host_to_host_mpixel_s"] = ex["host_to_host_mpixel_s"]
            result["config"]["host_to_host_mpixel_s"] = ex["host_to_host_mpixel_s"]
            if args.pipeline == "forward" and not args.brain_mask and not args.total_slices and args.slices == 1 \
                    and args.activation == "sine" and args.precision == "f16x3":
                # the default (driver-run) line also carries every other BASELINE configuration, measured in this process
                # behind the timed region: never `value`
                for d in (d_img, d_tiles, *d_outs, *d_recons):
                    d.free()
                result["extra"]["configs"] = other_configs(torch_first=args.torch_first)
                f32 = result["extra"]["configs"].get("fp32_trunk", {})
                if "value" in f32:
                    # the north star's strict reading (fp32 MFMA, v_mfma_f32_32x32x2_f32): top level and inside `roofline` (kept whole)
                    result["fp32"] = {"value": f32["value"], "unit": f32["unit"], "kernel": f32["kernel"], "peak_tflops": f32["peak_tflops"],
                                      "kernel_alone_frac": f32["kernel_alone_frac"], "timed_frac": f32["timed_frac"],
                                      "kernel_alone_avg_launch_ms": f32["kernel_alone_avg_launch_ms"], "command": f32["command"]}
                    result["roofline"]["fp32_trunk"] = {k: result["fp32"][k] for k in ("value", "kernel", "kernel_alone_frac", "timed_frac", "peak_tflops")}
                    result["roofline"].update({"fp32_trunk_mpixel_s": f32["value"], "fp32_trunk_kernel_alone_frac": f32["kernel_alone_frac"],
                                               "fp32_trunk_timed_frac": f32["timed_frac"]})
                c3 = result["extra"]["configs"].get("config3_64_slices_n1", {})
                if "value" in c3:
                    result["config"].setdefault("also_measured", {})["config3_64_slices_n1"] = {"value": c3["value"], "ms_per_step": c3["ms_per_step"]}
                    result["config"]["config3_64_slices_n1_mpixel_s"] = c3["value"]
        if world == 1 and not args.no_cpu_baseline and not deep:
            result["cpu_baseline"] = cpu_baseline(sd, tiles400, args.activation, args.cpu_seconds)
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

    def host_rate(x, slices):
        model(x)  # workspaces
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            model(x)
            n += 1
        return n * slices * 320 * 320 / (time.perf_counter() - t0) / 1e6

    _lib.check(lib.msiren_set_streams(h, 1))
    h2h = host_rate(tiles, 1)
    # the same call on page-locked arrays (model.pinned_empty / pin_outputs: what torch users get from pin_memory()), and a call of 8
    # slices, which cuts itself into chunks over the handle's two streams (uploads and downloads beside the other chunk's kernels)
    pin = model.pinned_empty(tiles.shape)
    pin[...] = tiles
    h2h_pinned = host_rate(pin, 1)   # (outputs come from the page-locked pool by default: here the tiles are page-locked as well)
    eight = np.concatenate([tiles] * 8)
    h2h_8 = host_rate(eight, 8)
    del pin, eight
    # slice -> slice on host arrays, one synchronous call per slice: the whole of metrics_error's per-slice work (error.py:231-249)
    img = d_img.numpy()[0]
    model.reconstruct(img)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        model.reconstruct(img)
        n += 1
    h2h_rec = n * 320 * 320 / (time.perf_counter() - t0) / 1e6
    _lib.check(lib.msiren_set_streams(h, streams))
    k = [0]

    def rstep():
        _lib.check(lib.msiren_reconstruct_slices_dev(h, d_img.ptr, 1, 320, 320, d_recons[k[0] % len(d_recons)].ptr))
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
    return {"host_to_host_mpixel_s": h2h, "host_to_host_pinned_mpixel_s": h2h_pinned, "host_to_host_8_slices_mpixel_s": h2h_8,
            "reconstruct_mpixel_s": rec, "host_slice_to_slice_mpixel_s": h2h_rec,
            "note": "after the timed region: host numpy -> host numpy through msiren_forward_tiles (PCIe-inclusive), one 320x320 slice per call "
                    "on pageable tiles (outputs from the mirror's page-locked pool, its default), on page-locked tiles as well (model.pinned_empty), 8 slices per call on pageable tiles (the call "
                    "pipelines itself); the device-resident slice -> slice pipeline, one slice per call; and the same pipeline as one synchronous call "
                    "per slice on host arrays (msiren_reconstruct_slices: host_slice_to_slice)"}


def scaling_selftest(args) -> int:
    """The N = 1 point of a scaling sweep goes through the launcher (RANK / WORLD_SIZE = 1 set by torchrun or spawn_ranks); the
    BENCH line does not.  Run both here, back to back on the same card, and compare: if they differ by more than 3 % the
    scaling curve's base is not the number the single-GPU line reports.  Prints ONE JSON line; exit code 1 on disagreement."""
    import io
    import subprocess

    from mri_inr_amd import launch

    if launch.under_launcher():
        raise SystemExit("--scaling-selftest starts its own processes: run it without a launcher")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--slices", str(args.slices), "--streams", str(args.streams), "--no-cpu-baseline", "--no-extras"]
    if args.total_slices:
        cmd += ["--total-slices", str(args.total_slices)]
    env = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    vals = {}
    for name in ("plain", "launched", "plain_again"):
        if name == "launched":
            out = io.StringIO()
            rc, text = launch.spawn_ranks(cmd, 1, timeout=args.launch_timeout, env=env, stdout=out)
        else:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.launch_timeout)
            rc, text = r.returncode, r.stdout
            sys.stderr.write(r.stderr[-2000:])
        lines = [l for l in text.splitlines() if l.startswith("{")]
        if rc != 0 or len(lines) != 1:
            print(json.dumps({"selftest": "scaling", "ok": False, "error": f"{name} run failed (rc {rc})"}), flush=True)
            return 1
        vals[name] = json.loads(lines[0])["value"]
    base = 0.5 * (vals["plain"] + vals["plain_again"])
    rel = abs(vals["launched"] - base) / base
    noise = abs(vals["plain"] - vals["plain_again"]) / base
    ok = rel <= 0.03
    print(json.dumps({"selftest": "scaling", "ok": ok, "unit": "Mpixel/s", **vals, "launched_vs_plain_rel_diff": rel,
                      "plain_run_to_run_rel_diff": noise, "tolerance": 0.03}), flush=True)
    return 0 if ok else 1


def n1_reference():
    """The stored N = 1 figure of BASELINE configs[2] (64 slices per step on ONE GPU, two streams: `bench.py --total-slices 64`), which a
    multi-rank run divides its strong-scaling value by: profiles/<round>/n1_reference.json, newest round first."""
    import glob

    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "n1_reference.json")), reverse=True):
        try:
            d = json.load(open(path))
            return float(d["config3_64_slices_n1"]["value"]), os.path.relpath(path, REPO) + ": " + d["config3_64_slices_n1"].get("source", "")
        except Exception:  # noqa: BLE001
            continue
    return None, None


def strong_config3(model, group, lib, h, _lib, syn, shard_range, rank, world, args, device_sync, fence, backend, total=64, steps=24, warmup=3):
    """BASELINE configs[2] behind the weak region of a multi-rank run: a fixed batch of 64 slices per step, contiguous slice shards
    (dist.shard_range), no data-path collective; barrier + MAX over ranks around the timed steps like the headline region.
    efficiency_vs_n1 = value / (world x the stored N = 1 value of the same workload)."""
    lo, hi = shard_range(total, rank, world)
    n_sl = hi - lo
    B = n_sl * 400
    imgs = np.stack([syn.make_slice(k) for k in range(lo, hi)]) if n_sl else np.zeros((0, 320, 320), np.float32)
    d_img = model.device_array((max(n_sl, 1), 320, 320))
    d_tiles = model.device_array((max(B, 1), 32, 32))
    d_outs = [model.device_array((max(B, 1), 24, 24)) for _ in range(max(2, args.streams))]
    if n_sl:
        _lib.check(lib.msiren_memcpy_h2d(h, d_img.ptr, imgs.ctypes.data, imgs.nbytes))
    _lib.check(lib.msiren_image_to_patches_dev(h, d_img.ptr, n_sl, 320, 320, d_tiles.ptr))
    model.sync()
    k = [0]

    def step():
        _lib.check(lib.msiren_forward_tiles_dev(h, d_tiles.ptr, B, d_outs[k[0] % len(d_outs)].ptr))
        k[0] += 1

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    local = time.perf_counter() - t0
    group.barrier()
    elapsed = group.max(local)
    per_rank = [0.0] * world
    per_rank[rank] = float(n_sl)
    per_rank = [int(x) for x in group.max_array(per_rank)]
    for d in (d_img, d_tiles, *d_outs):
        d.free()
    value = total * 320 * 320 * steps / elapsed / 1e6
    ref, src = n1_reference()
    comm_ranks, _ = group.info()
    return {"workload": f"BASELINE configs[2]: batch of {total} slices = {total * 400} tiles per step, contiguous slice shards over {world} ranks "
                        f"(strong scaling), tiles and outputs resident in HBM, {args.streams} streams per rank",
            "value": value, "unit": "Mpixel/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup, "scaling": "strong",
            "n_gpus": world, "slices_per_rank": per_rank, "rccl_ranks": comm_ranks if backend in ("rccl", "nccl") else None, "comm_ranks": comm_ranks,
            "n1_reference_value": ref, "n1_reference_source": src,
            "efficiency_vs_n1": (value / (world * ref)) if ref else None,
            "note": "the scaling claim of the north star (>= 0.9 linear at 8 GPUs) is about THIS number; the line's `value` is the weak "
                    "one-slice-per-rank region, kept so that N = 1 equals the single-GPU line"}


def other_configs(launch_timeout=90.0, wall_budget=240.0, torch_first=False):
    """BASELINE.json configs 3, 4, 5 and the exact-fp32 trunk at N = 1, each measured by THIS script in a child process of its
    own behind the headline's timed region (device-resident tiles, random-init weights of the named architecture; the parent
    only waits): the numbers are those of the stand-alone commands, summarised.  (Measured in-process on further handles the
    two-stream figures came out 8-22 % low: a process has a handful of hardware queues, and the streams of a third and fourth
    handle share them.)"""
    import subprocess

    from mri_inr_amd import launch

    env = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    runs = {
        "config3_64_slices_n1": (["--total-slices", "64", "--steps", "24", "--warmup", "3"],
                                 "BASELINE configs[2]: 64 slices = 25600 tiles per call, one GPU, two streams"),
        "config3_64_slices_n1_one_stream": (["--total-slices", "64", "--streams", "1", "--steps", "24", "--warmup", "3"],
                                            "the same on a one-stream handle: one weight-stationary trunk launch behind the call's whole prologue"),
        "config3_8_slices_per_rank": (["--slices", "8", "--steps", "120", "--warmup", "10"],
                                      "8 slices = 3200 tiles per call: one rank's share of configs[2] on 8 GPUs"),
        "config4_morlet": (["--activation", "morlet", "--steps", "600", "--warmup", "30"], "BASELINE configs[3]: Morlet activation, one slice per call"),
        "fp32_trunk": (["--precision", "fp32", "--steps", "300", "--warmup", "20"],
                       "configs[1] on the exact-fp32 trunk (v_mfma_f32_32x32x2_f32), one slice per call"),
        "config5_deep_residual_bf16": (["--model", "deep_residual", "--precision", "bf16", "--steps", "300", "--warmup", "20"],
                                       "BASELINE configs[4]: deep residual 10x512, latent 128, bf16 MFMA (own semantics, parity unpinned); three streams (this model's default)"),
    }
    out = {}
    t_start = time.perf_counter()
    for name, (flags, what) in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-extras"] + flags + (["--torch-first"] if torch_first else [])
        # a side measurement must not delay the headline line without bound: a child gets at most `launch_timeout` seconds, all of them
        # together `wall_budget` (each takes 10-20 s, mostly `import torch` on a fresh box); what does not fit is reported as skipped
        left = wall_budget - (time.perf_counter() - t_start)
        if left < 15.0:
            out[name] = {"error": f"skipped: the side measurements' wall budget of {wall_budget:.0f} s is spent", "command": " ".join(cmd[1:])}
            continue
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(launch_timeout, left))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or len(lines) != 1:
                out[name] = {"error": f"rc {r.returncode}: {r.stderr[-300:]}", "command": " ".join(cmd[1:])}
                continue
            d = json.loads(lines[0])
        except Exception as exc:  # noqa: BLE001 -- a side measurement must not take the headline line down
            out[name] = {"error": repr(exc), "command": " ".join(cmd[1:])}
            continue
        rf, ka = d["roofline"], d.get("roofline_kernel_alone") or d["roofline"]
        out[name] = {"workload": what, "command": "bench.py " + " ".join(flags), "value": d["value"], "unit": d["unit"],
                     "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"], "streams": d["config"]["streams"],
                     "slices_per_step": d["config"]["slices_per_step_total"], "peak_tflops": rf["peak"],
                     "timed_region_kernel": rf["kernel"], "timed_frac": rf["frac"],
                     "timed_region_kernels": [{k: x[k] for k in ("kernel", "launches", "avg_launch_ms", "frac")} for x in rf["timed_region_kernels"]],
                     "kernel": ka["kernel"], "kernel_alone_frac": ka["frac"], "kernel_alone_avg_launch_ms": ka["avg_launch_ms"]}
    out["note"] = ("each entry: a child `bench.py` run behind the headline's timed region (`command`), never the headline's `value`; "
                   "`kernel` / `kernel_alone_frac` = dominant trunk instance of a one-stream phase (or of the timed region when it is "
                   "one-stream) as the library names it; `timed_frac` = the entry's own roofline.frac")
    return out


if __name__ == "__main__":
    main()
