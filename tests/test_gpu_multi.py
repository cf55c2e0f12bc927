"""Multi-GPU plumbing on the one card of the GPU box: RCCL through the C ABI (include/msiren.h, "multi-GPU") with a
communicator of one rank, and bench.py starting its own ranks (gloo lets two ranks share the card).  The 2-rank
logic of partition / broadcast is covered on the CPU in test_dist_gloo.py and test_launch.py."""
import ctypes as C
import io
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from mri_inr_amd import launch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, timeout=600, env=None):
    e = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(script)], cwd=ROOT, env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_rccl_through_the_c_abi_single_rank_torch_free():
    """ncclGetUniqueId -> ncclCommInitRank -> ncclBroadcast of the state_dict blob -> commit -> forward, barrier and
    MAX-reduce, all through libmsiren: no torch in the process, one HIP runtime."""
    r = _run("""
        import ctypes as C, sys, numpy as np
        from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn
        from oracle import siren_oracle as orc
        sd = syn.make_state_dict(seed=3, trained_like=True)
        m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine")
        m._sd.update(sd)
        lib, h = m._lib, m._h
        uid = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(lib.msiren_comm_unique_id(uid, _lib.COMM_ID_BYTES))
        assert any(uid.raw)
        _lib.check(lib.msiren_comm_init_rank(h, uid.raw, _lib.COMM_ID_BYTES, 1, 0))
        assert lib.msiren_comm_init_rank(h, uid.raw, _lib.COMM_ID_BYTES, 1, 0) == _lib.E_STATE   # already a member
        n, r = C.c_int32(), C.c_int32()
        _lib.check(lib.msiren_comm_info(h, C.byref(n), C.byref(r)))
        assert (n.value, r.value) == (1, 0)
        m._push_tensors()
        _lib.check(lib.msiren_broadcast_weights(h, 0))          # collective; commits
        assert lib.msiren_broadcast_weights(h, 1) == _lib.E_INVALID
        m._committed = True
        a = np.empty(sd["net.layers.1.weight"].shape, np.float32)
        _lib.check(lib.msiren_get_tensor(h, b"net.layers.1.weight", a.ctypes.data, a.size))
        assert np.array_equal(a, sd["net.layers.1.weight"])
        tiles = np.random.default_rng(0).random((16, 32, 32), dtype=np.float32)
        out = m(tiles)
        ref = orc.modulated_siren_forward(sd, tiles, num_layers=5, dtype=np.float64)
        err = float(np.abs(out - ref).max() / np.abs(ref).max())
        v = (C.c_double * 2)(err, -1.0)
        _lib.check(lib.msiren_comm_allreduce_max_f64(h, v, 2))
        _lib.check(lib.msiren_comm_barrier(h))
        assert (v[0], v[1]) == (err, -1.0)
        _lib.check(lib.msiren_comm_destroy(h))
        _lib.check(lib.msiren_comm_barrier(h))                  # a communicator of one again
        assert "torch" not in sys.modules
        print("NERR", err)
        sys.exit(0 if err < 1e-4 else 1)
    """)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NERR" in r.stdout


def test_rccl_group_object_and_comm_init_all():
    """dist.RcclGroup (what bench.py uses) with WORLD_SIZE unset, and the one-process form
    msiren_comm_init_all / msiren_broadcast_weights_all on the one device that is here."""
    r = _run("""
        import ctypes as C, sys, numpy as np
        from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn
        from mri_inr_amd.dist import RcclGroup
        sd = syn.make_state_dict(seed=5)
        def model():
            return ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                                  use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                                  outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine")
        tiles = np.random.default_rng(0).random((5, 32, 32), dtype=np.float32)
        ref = model(); ref.load_state_dict(sd); want = ref(tiles)
        m = model(); g = RcclGroup(m); g.broadcast_weights(0, sd); g.barrier()
        assert g.max(3.5) == 3.5 and np.array_equal(m(tiles), want)
        g.destroy()
        m2 = model(); m2._sd.update(sd); m2._push_tensors()
        hs = (C.c_void_p * 1)(m2._h)
        _lib.check(m2._lib.msiren_comm_init_all(hs, 1))
        _lib.check(m2._lib.msiren_broadcast_weights_all(hs, 1, 0))
        m2._committed = True
        assert np.array_equal(m2(tiles), want)
        two = (C.c_void_p * 2)(m._h, m2._h)
        assert m2._lib.msiren_comm_init_all(two, 2) in (_lib.E_INVALID, _lib.E_STATE)   # same device twice / already a member
        assert "torch" not in sys.modules
        print("OK")
    """)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("act", ["sine", "morlet"])
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_commit_without_the_grid_buffer(prec, act):
    """A C-ABI consumer may omit "grid" (the reference registers it as a buffer, modulated_siren.py:427-433): the library rebuilds
    it, for every trunk.  Checked against the REFERENCE run the same way -- tests/golden/reference_grid.npz: "grid" dropped from the
    state_dict, strict=False, so the reference's own linspace / meshgrid buffer produced the fixture -- and, with the reference's buffer
    loaded instead, against the same fixture again; the rebuilt grid gives the bits of the build's own."""
    from conftest import load_golden, nerr
    from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

    g = load_golden("reference_grid.npz")
    sd = syn.make_state_dict(seed=7, trained_like=True)
    kw = dict(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0, use_bias=True,
              dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16,
              siren_patch_size=24, device="cuda:0", activation=act, precision=prec)
    tiles = np.random.default_rng(1).random((7, 32, 32), dtype=np.float32)      # the fixture's input (oracle/gen_fixtures.py)
    mods = syn.make_mods(34, 5, 5, 256)
    a = ModulatedSiren(**kw)
    a.load_state_dict(sd)
    want = a.to("cuda:0")(tiles)
    b = ModulatedSiren(**kw)
    for k, v in sd.items():
        if k != "grid":
            v = np.ascontiguousarray(v, np.float32)
            _lib.check(b._lib.msiren_set_tensor(b._h, k.encode(), v.ctypes.data, v.size))
    _lib.check(b._lib.msiren_commit_weights(b._h))
    b._committed = True
    got = b(tiles)
    assert np.array_equal(got, want)                                            # rebuilt grid == the build's own
    assert nerr(got, g[f"forward_{act}"]) < 1e-4                                # ... and within the gate of the reference on ITS grid
    assert nerr(b.forward_mods(mods), g[f"trunk_{act}"].reshape(5, 24, 24)) < 1e-4
    c = ModulatedSiren(**kw)
    c.load_state_dict(dict(sd, grid=g["grid_24"]))                              # the reference's buffer, as a real checkpoint carries it
    c.to("cuda:0")
    assert np.array_equal(c.state_dict()["grid"], g["grid_24"])
    assert nerr(c(tiles), g[f"forward_{act}"]) < 1e-4
    assert nerr(c.forward_mods(mods), g[f"trunk_{act}"].reshape(5, 24, 24)) < 1e-4


def _model_kwargs(**over):
    kw = dict(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0, use_bias=True,
              dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16,
              siren_patch_size=24, device="cuda:0", activation="sine")
    kw.update(over)
    return kw


def test_weights_blob_export_import_is_the_receive_side_of_the_broadcast():
    """handle A (loaded) -> msiren_weights_export -> handle B (fresh, same configuration) -> msiren_weights_import:
    unpack + commit, the very code a non-root rank of msiren_broadcast_weights runs.  state_dict equal, outputs bit-equal;
    a trunk-only source (no encoder keys) leaves the receiver without an encoder; blobs of another model are refused."""
    from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

    sd = syn.make_state_dict(seed=21, trained_like=True)
    tiles = np.random.default_rng(2).random((9, 32, 32), dtype=np.float32)
    a = ModulatedSiren(**_model_kwargs())
    a.load_state_dict(sd)
    want = a.to("cuda:0")(tiles)
    blob = a.export_weights()
    assert blob.dtype == np.float32 and blob[:1].view(np.uint32)[0] == 0x4257534D   # "MSWB"

    b = ModulatedSiren(**_model_kwargs())          # fresh: holds its own random weights
    assert not np.array_equal(b.to("cuda:0")(tiles), want)
    b.import_weights(blob)
    got_sd = b.state_dict()
    assert set(got_sd) == set(sd)
    for k in sd:
        assert np.array_equal(got_sd[k], sd[k]), k
    assert np.array_equal(b(tiles), want)
    assert np.array_equal(b.export_weights(), blob)  # and it re-exports the same image

    # source without encoder keys: the receiver ends up without them (forward_latent works, forward raises)
    c = ModulatedSiren(**_model_kwargs())
    for k, v in sd.items():
        if not k.startswith("encoder."):
            v = np.ascontiguousarray(v, np.float32)
            _lib.check(c._lib.msiren_set_tensor(c._h, k.encode(), v.ctypes.data, v.size))
    n = C.c_size_t()
    _lib.check(c._lib.msiren_weights_blob_size(c._h, C.byref(n)))
    part = np.empty(n.value, np.float32)
    _lib.check(c._lib.msiren_weights_export(c._h, part.ctypes.data, part.size))
    d = ModulatedSiren(**_model_kwargs())
    d.import_weights(part)
    assert not any(k.startswith("encoder.") for k in d.state_dict())
    z = np.random.default_rng(3).standard_normal((4, 256)).astype(np.float32)
    assert np.array_equal(d.forward_latent(z), a.forward_latent(z))
    with pytest.raises(_lib.MsirenError, match="encoder"):
        d(tiles)

    # refusals leave the receiver as it was
    e = ModulatedSiren(**_model_kwargs(num_layers=4))
    before = e.to("cuda:0")(tiles)
    with pytest.raises(RuntimeError):
        e.import_weights(blob)                      # another depth: MSIREN_E_SHAPE
    bad = blob.copy()
    bad[0] = 1.0
    with pytest.raises(ValueError):
        b.import_weights(bad)                       # not a blob: MSIREN_E_INVALID
    assert np.array_equal(e(tiles), before) and np.array_equal(b(tiles), want)
    assert c._lib.msiren_weights_export(c._h, part.ctypes.data, part.size - 1) == _lib.E_SHAPE


def _build_rccl_stub(tmp_path):
    """tests/stubs/rccl_stub.cpp -> librccl_stub.so (host code only; links the HIP runtime libmsiren links)."""
    out = os.path.join(str(tmp_path), "librccl_stub.so")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp"), "-o", out, "-L/opt/rocm/lib", "-lamdhip64"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return out


TWO_RANK_WORKER = """
    import os, sys, numpy as np
    from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn
    from mri_inr_amd.dist import RcclGroup
    rank = int(os.environ["RANK"])
    no_encoder = os.environ.get("DROP_ENCODER") == "1"
    sd = syn.make_state_dict(seed=33, trained_like=True)          # what rank 0 loads; the others only use it to check
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine")
    g = RcclGroup(m)
    assert g.info() == (2, rank), g.info()
    if rank == 0 and no_encoder:
        for k in list(m._sd):                                     # a trunk + modulator source: nothing pushed yet
            if k.startswith("encoder."):
                del m._sd[k]
            else:
                m._sd[k] = sd[k]
        g.broadcast_weights(0, None)
    else:
        g.broadcast_weights(0, sd if rank == 0 else None)         # rank 1: receive -> unpack -> commit
    have = m.state_dict()
    for k, v in sd.items():
        if no_encoder and k.startswith("encoder."):
            continue
        assert np.array_equal(have[k], v), (rank, k)
    z = np.random.default_rng(5).standard_normal((6, 256)).astype(np.float32)
    out = m.forward_latent(z)
    if not no_encoder:
        tiles = np.random.default_rng(6).random((6, 32, 32), dtype=np.float32)
        out = np.concatenate([out, m(tiles)])
    cs = float(out.view(np.uint32).astype(np.float64).sum())
    assert g.max(cs) == cs and g.min(cs) == cs, (rank, cs)         # both ranks computed the same bits
    g.barrier()
    g.destroy()
    assert "torch" not in sys.modules
    print(f"RANK{rank} OK {cs:.0f}", flush=True)
"""


@pytest.mark.parametrize("drop_encoder", ["0", "1"])
def test_two_ranks_on_one_card_broadcast_receive_path_through_a_stub_rccl(tmp_path, drop_encoder):
    """RCCL proper refuses two ranks on one device, so the 1-GPU box runs the two-rank path of dist.RcclGroup /
    msiren_broadcast_weights against tests/stubs/rccl_stub.cpp (file-carried collectives, same C signatures, device
    pointers): rank 1 executes the receive branch for real -- blob D2H, unpack, commit -- and must end up with rank 0's
    state_dict and bit-identical outputs."""
    stub = _build_rccl_stub(tmp_path)
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(TWO_RANK_WORKER))
    env = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    env.update({"MSIREN_RCCL_LIB": stub, "RCCL_STUB_DIR": str(tmp_path), "PYTHONPATH": ROOT, "DROP_ENCODER": drop_encoder})
    out, err = io.StringIO(), io.StringIO()
    rc, rank0 = launch.spawn_ranks([sys.executable, str(script)], 2, timeout=300, env=env, stdout=out, stderr=err)
    assert rc == 0, out.getvalue()[-2000:] + err.getvalue()[-4000:]
    assert "RANK0 OK" in rank0 and "RANK1 OK" in err.getvalue()


def test_bench_two_ranks_rehearsal_reports_what_the_communicator_saw(tmp_path):
    """bench.py --gpus 2 on the one card through the stub: the line carries the communicator's own rank count, no
    fallback, and the cross-rank check of the replicated weights."""
    stub = _build_rccl_stub(tmp_path)
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
               {"MSIREN_BENCH_ALLOW_SHARED": "1", "MSIREN_RCCL_LIB": stub, "RCCL_STUB_DIR": str(tmp_path)}, timeout=400)
    assert d["n_gpus"] == 2 and d["collective_fallback"] is False
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["rccl_lib"] == stub
    assert d["config"]["ranks_hold_identical_weights"] is True
    # the line carries BOTH regions: the weak one-slice-per-rank `value` (what N = 1 reports) and, behind it in the same process group,
    # BASELINE configs[2] strong -- 64 slices sharded 32 + 32 -- with its efficiency against the stored N = 1 figure (the scaling claim)
    assert d["scaling"] == "weak" and d["config"]["slices_per_step_total"] == 2
    assert abs(d["value"] - 2 * 320 * 320 / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    st = d["extra"]["configs"]["config3_64_slices_strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["slices_per_rank"] == [32, 32] and st["rccl_ranks"] == 2
    assert abs(st["value"] - 64 * 320 * 320 / st["ms_per_step"] / 1e3) < 1e-6 * st["value"]
    assert st["n1_reference_value"] > 0 and "n1_reference.json" in st["n1_reference_source"]
    assert abs(st["efficiency_vs_n1"] - st["value"] / (2 * st["n1_reference_value"])) < 1e-9
    assert 0.3 < st["efficiency_vs_n1"] < 0.75        # two ranks SHARE one card here: about half, never the claim itself
    assert d["config"]["also_measured"]["config3_64_slices_strong"]["value"] == st["value"]
    assert d["config"]["config3_strong_mpixel_s"] == st["value"] and d["config"]["config3_strong_efficiency_vs_n1"] == st["efficiency_vs_n1"]
    assert d["config"]["hip_runtime"]["torch_bundled"] is False and d["config"]["torch_first"] is False


def test_bench_under_torchrun_as_the_driver_starts_it(tmp_path):
    """The scaling sweep's command line, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 --steps K --warmup W` -- two ranks on the one card through the stub RCCL.  Under torchrun the
    unique id travels through the agent's TCP store (TORCHELASTIC_USE_AGENT_STORE; mri_inr_amd/launch.py), i.e. torch is imported in the
    rank AFTER libmsiren has brought the system HIP runtime up: the line must say so (torch-free runtime, rccl backend, no fallback) and
    carry both regions."""
    from mri_inr_amd.launch import free_port

    stub = _build_rccl_stub(tmp_path)
    e = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    e.update({"MSIREN_BENCH_ALLOW_SHARED": "1", "MSIREN_RCCL_LIB": stub, "RCCL_STUB_DIR": str(tmp_path)})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["collective_fallback"] is False
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["ranks_hold_identical_weights"] is True
    assert d["config"]["hip_runtime"]["torch_bundled"] is False            # libmsiren came first: the system runtime
    assert d["extra"]["configs"]["config3_64_slices_strong"]["slices_per_rank"] == [32, 32]
    assert d["cpu_baseline"] is None                                       # (rank 0 at N = 1 only)


def _bench(args, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_single_gpu_line_carries_roofline_cpu_baseline_and_extras():
    d = _bench(["--steps", "20", "--warmup", "5", "--cpu-seconds", "2"])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["metric"] == "Mpixels/sec reconstructed (320x320 slice, hidden=256, 5 layers)"
    assert "configs[1]" in d["config"]["workload"] and d["config"]["warmup_steps_run"] >= 5
    assert abs(d["value"] - 320 * 320 / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    # the roofline block names the kernel of the TIMED region as the library reports it (two streams: the register-resident trunk)
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["kernel"] == "siren_trunk_f16x3n_kernel<0,3,5>" and rf["launches"] == 20 and 0.2 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(rf["achieved"] - rf["flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * rf["achieved"]
    assert rf["flops_per_launch"] == 525824 * 576 * 400 and rf["timed_region_kernels"][0]["kernel"] == rf["kernel"]
    assert abs(rf["avg_launch_ms"] - d["ms_per_step"]) < 0.02 * d["ms_per_step"]   # consistent with `value`
    assert rf["traffic"] is None or "profiles/" in rf["traffic_source"]
    ka = d["roofline_kernel_alone"]
    assert ka["kernel"] == "siren_trunk_f16x3w_kernel<0,4>" and ka["launches"] >= 200 and 0.2 < ka["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert "best of" in d["cpu_baseline"]["sample"]
    assert d["collective_fallback"] is False and d["config"]["comm_ranks"] == 1
    ex = d["extra"]
    assert 0 < ex["host_to_host_mpixel_s"] < d["value"] * 1.05   # PCIe-inclusive: never faster than device-resident
    # SURVEY section 8(d)'s primary region and the strict-fp32 trunk as top-level keys (and inside config / roofline, which the driver keeps)
    assert d["host_to_host"]["value"] == ex["host_to_host_mpixel_s"] and d["host_to_host"]["unit"] == "Mpixel/s"
    assert d["config"]["also_measured"]["host_to_host_mpixel_s"] == ex["host_to_host_mpixel_s"]
    assert d["config"]["host_to_host_mpixel_s"] == ex["host_to_host_mpixel_s"] and d["roofline"]["fp32_trunk_mpixel_s"] == d["fp32"]["value"]
    assert d["config"]["hip_runtime_version"] == d["config"]["hip_runtime"]["hip_runtime_version"] and d["config"]["config3_64_slices_n1_mpixel_s"] > 0
    assert d["fp32"]["kernel"] == "siren_trunk_f32_kernel<256,0,0>" and d["fp32"]["value"] == ex["configs"]["fp32_trunk"]["value"]
    assert 0.5 < d["fp32"]["kernel_alone_frac"] < 1.0 and d["roofline"]["fp32_trunk"]["kernel_alone_frac"] == d["fp32"]["kernel_alone_frac"]
    rt = d["config"]["hip_runtime"]
    assert rt["torch_bundled"] is False and rt["hip_runtime_version"] == rt["built_against_hip"] and d["config"]["torch_first"] is False
    assert ex["reconstruct_mpixel_s"] > 0 and 0 < ex["host_slice_to_slice_mpixel_s"] < ex["reconstruct_mpixel_s"] * 1.05
    # every other BASELINE configuration rides in the driver-run line (child runs behind the timed region, never `value`)
    cfgs = ex["configs"]
    want = {"config3_64_slices_n1": "siren_trunk_f16x3w_kernel<0,4>", "config3_64_slices_n1_one_stream": "siren_trunk_f16x3w_kernel<0,4>",
            "config3_8_slices_per_rank": "siren_trunk_f16x3w_kernel<0,4>", "config4_morlet": "siren_trunk_f16x3w_kernel<1,4>",
            "fp32_trunk": "siren_trunk_f32_kernel<256,0,0>", "config5_deep_residual_bf16": "siren_trunk_x1w_kernel<1,0,1>"}
    for name, kern in want.items():
        c = cfgs[name]
        assert "error" not in c, (name, c)
        assert c["kernel"] == kern and c["value"] > 0 and c["ms_per_step"] > 0 and 0.1 < c["kernel_alone_frac"] < 1.0, (name, c)
        assert abs(c["value"] - c["slices_per_step"] * 320 * 320 / c["ms_per_step"] / 1e3) < 1e-6 * c["value"]
    # on a one-stream handle a large call runs the weight-stationary trunk over the whole batch (round 5: no longer cut in two,
    # profiles/r5/07_*); with two streams consecutive calls overlap on the register-resident one
    assert {k["kernel"] for k in cfgs["config3_64_slices_n1_one_stream"]["timed_region_kernels"]} == {"siren_trunk_f16x3w_kernel<0,4>"}
    assert {k["kernel"] for k in cfgs["config3_64_slices_n1"]["timed_region_kernels"]} == {"siren_trunk_f16x3n_kernel<0,3,5>"}


def test_bench_gpus2_starts_its_own_ranks_gloo_rehearsal_on_one_card():
    """`python bench.py --gpus 2` with no launcher: two ranks, n_gpus 2 in the line (weak: one slice per rank;
    strong: a fixed batch of 6 slices split 3 + 3)."""
    env = {"MSIREN_BENCH_BACKEND": "gloo"}
    d = _bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["slices_per_step_total"] == 2
    assert abs(d["value"] - 2 * 320 * 320 / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]
    assert d["extra"]["configs"]["config3_64_slices_strong"]["slices_per_rank"] == [32, 32]      # the strong region rides along
    assert d["extra"]["configs"]["config3_64_slices_strong"]["rccl_ranks"] is None and d["extra"]["configs"]["config3_64_slices_strong"]["comm_ranks"] == 2
    assert d["config"]["hip_runtime"]["torch_bundled"] is True                                    # (gloo: torch came first)
    s = _bench(["--gpus", "2", "--total-slices", "6", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], env)
    assert s["n_gpus"] == 2 and s["scaling"] == "strong"
    assert s["config"]["slices_per_step_total"] == 6 and s["config"]["slices_per_step_rank0"] == 3
    assert abs(s["value"] - 6 * 320 * 320 / s["ms_per_step"] / 1e3) < 1e-6 * s["value"]


def test_bench_two_rccl_ranks_on_one_card_rendezvous_then_fall_back():
    """Two self-launched ranks with the default (RCCL, torch-free) backend on the ONE card of this box: the unique-id
    rendezvous and ncclCommInitRank run across the two processes for real, RCCL then refuses the shared device
    ("invalid usage"), and bench.py falls back to seeded weights + gloo plumbing instead of losing the measurement --
    the JSON line says so."""
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], {"MSIREN_BENCH_ALLOW_SHARED": "1"},
               timeout=400)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["collective_fallback"] is True and d["config"]["rccl_ranks"] is None and d["config"]["comm_ranks"] == 2
    assert d["config"]["ranks_hold_identical_weights"] is True
    assert "RCCL init failed" in d["config"]["backend"]
    assert abs(d["value"] - 2 * 320 * 320 / d["ms_per_step"] / 1e3) < 1e-6 * d["value"]


def test_bench_torch_first_runs_on_torchs_bundled_hip_runtime():
    """--torch-first = the reference's own host program (torch imported before anything else): libmsiren's HIP calls resolve to torch's
    bundled libamdhip64 (same soname).  The line names the runtime; the numbers of both orders are on file in profiles/r6/02_*."""
    d = _bench(["--torch-first", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"])
    rt = d["config"]["hip_runtime"]
    assert d["config"]["torch_first"] is True and rt["torch_bundled"] is True and "/torch/lib/" in rt["libamdhip64"]
    assert rt["hip_runtime_version"] != rt["built_against_hip"]      # this image: torch bundles ROCm 7.0, the system is 7.2
    assert d["value"] > 0 and d["roofline"]["kernel"] == "siren_trunk_f16x3n_kernel<0,3,5>"


def test_bench_strong_scaling_config3_single_gpu():
    d = _bench(["--total-slices", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["config"]["patches_per_step_rank0"] == 25600
    assert "configs[2]" in d["config"]["workload"]
