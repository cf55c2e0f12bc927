"""Co-residency budget of DESIGN.md §4.3, checked at compile time (hipcc cross-compiles without a GPU).

With two streams the encoder / modulator / tiling kernels of call k+1 must fit on a CU BESIDE the persistent
trunk workgroup of call k: the trunk may take at most 416 of the 512 registers per SIMD lane (arch + accumulator),
everything that runs beside it at most what is left (96).  Register allocation is sensitive to small source changes -- one extra
vector value in the trunk once cost 10 % of the pipelined throughput -- so the numbers are pinned here.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

TU = r"""
#include <hip/hip_runtime.h>
#include "siren_trunk_f16x3n.hip.h"
#include "encoder_modulator.hip.h"
#include "encoder_modulator_f16x3.hip.h"
#include "tiling.hip.h"
template __global__ void msiren::siren_trunk_f16x3n_kernel<0, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3n_kernel<1, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::linear_mfma_tile_kernel<2, 2>(msiren::ModulatorMfmaParams);
template __global__ void msiren::latent_mods_f16x3_kernel<2, 2, 2, 3>(msiren::EmTailParams);
template __global__ void msiren::encoder_conv_f16x3_kernel<1>(msiren::EncoderParams, const float*, msiren::em_u4*, float*);
template __global__ void msiren::siren_trunk_f32_cond_kernel<0>(msiren::TrunkParams);
template __global__ void msiren::siren_trunk_f32_cond_kernel<1>(msiren::TrunkParams);
"""


def _usage(tmp_path):
    src = tmp_path / "budget.hip"
    src.write_text(TU)
    cmd = [HIPCC, "-O3", "-std=c++17", "-c", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"),
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", str(src), "-o", str(tmp_path / "budget.o")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out, name = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            out[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_trunk_and_its_neighbours_fit_on_one_cu(tmp_path):
    usage = _usage(tmp_path)
    trunks = {k: v for k, v in usage.items() if "siren_trunk_f16x3" in k}
    assert sum("siren_trunk_f32_cond" in k for k in usage) == 2   # the conditional exact-fp32 launch runs beside the other stream's trunk
    beside = {k: v for k, v in usage.items() if k not in trunks}
    assert len(trunks) == 2 and len(beside) >= 11 and any("linear_mfma_tile" in k for k in beside), list(usage)  # the register-resident 16x16x32 trunk, sine / Morlet
    # round 5: the one-launch split-fp16 prologue (ring of 2) and its conv kernels run beside the trunk as well -- and without scratch
    tail = [k for k in beside if "latent_mods_f16x3" in k]
    assert len(tail) == 1 and sum("encoder_conv_f16x3" in k for k in beside) == 1, list(beside)
    for k in beside:
        if "f16x3" in k:
            assert beside[k]["ScratchSize"] == 0, (k, beside[k])
    # (the tail's images are dynamic LDS: 33 024 bytes for H = Z = 256, checked where it is computed)

    def alloc(u):  # registers one wave occupies in the unified 512-entry file of a SIMD lane (gfx90a+)
        acc_offset = (u["VGPRs"] + 3) // 4 * 4
        return (acc_offset + u.get("AGPRs", 0) + 7) // 8 * 8

    free = 512
    for k, u in trunks.items():
        assert u["ScratchSize"] == 0, (k, u)  # no spills in the hot kernel
        free = min(free, 512 - alloc(u))
    assert free >= 96, (free, trunks)  # 16x16x32 kernel: 416 (sine) today
    for k, u in beside.items():
        assert alloc(u) <= free, (k, u, free)
        assert u["LDS"] <= 34 * 1024, (k, u)  # what a ring of 3 leaves free


WS_TU = r"""
#include <hip/hip_runtime.h>
#include "siren_trunk_f16x3w.hip.h"
template __global__ void msiren::siren_trunk_f16x3w_kernel<0, 4, 0>(msiren::TrunkWsParams);
template __global__ void msiren::siren_trunk_f16x3w_kernel<1, 4, 0>(msiren::TrunkWsParams);
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_weight_stationary_trunk_owns_its_register_files(tmp_path):
    """siren_trunk_f16x3w.hip.h manages a[0:255] (weight fragments) and v[192:255] (accumulators) BY NAME in asm
    statements; the compiler is kept out of them (amdgpu_num_vgpr(192) + placeholder values).  Should a compiler or source
    change let it back in -- a value parked in an AGPR over the fragments, a spill, an allocation beyond v191 -- results
    would be silently wrong or the kernel would fault: checked in the ISA, at build time."""
    src = tmp_path / "ws.hip"
    src.write_text(WS_TU)
    asm = tmp_path / "ws.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-S", "--cuda-device-only", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"),
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-fno-slp-vectorize", str(src), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    text = asm.read_text()
    kernels = re.split(r"\n(?=_ZN6msiren25siren_trunk_f16x3w_kernel)", text)[1:]
    assert len(kernels) == 2
    for k in kernels:
        body = k.split(".amdhsa_kernel")[0]
        assert "v_accvgpr" not in body and "scratch_" not in body
        assert body.count("v_mfma_f32_16x16x32_f16") == 12 * 192          # twelve slot bodies (variant x parity x flavour)
        for line in body.splitlines():
            code = line.split(";")[0]
            if re.search(r"\ba\[", code):                               # an AGPR operand: only where the kernel put it
                assert code.split()[0] in ("v_mfma_f32_16x16x32_f16", "global_load_dwordx4"), line
            hi = [int(x, 0) for x in re.findall(r"\bv\[(0x[0-9a-f]+|\d+):", code)] + [int(x) for x in re.findall(r"\bv(\d+)\b", code)]
            if hi and max(hi) >= 192:                                     # an accumulator register: MFMA, or the sine / move that reads it
                assert code.split()[0] in ("v_mfma_f32_16x16x32_f16", "v_sin_f32", "v_mov_b32"), line
    for m in re.finditer(r"\.amdhsa_next_free_vgpr (\d+)", text):
        assert int(m.group(1)) == 512
    for m in re.finditer(r"\.amdhsa_accum_offset (\d+)", text):
        assert int(m.group(1)) == 256
    assert len(re.findall(r"\.amdhsa_next_free_vgpr", text)) == 2
