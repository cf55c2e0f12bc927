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
#include "siren_trunk_f16x3.hip.h"
#include "siren_trunk_f16x3n.hip.h"
#include "encoder_modulator.hip.h"
#include "tiling.hip.h"
template __global__ void msiren::siren_trunk_f16x3_kernel<0, 3, 0>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3_kernel<1, 3, 0>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3n_kernel<0, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3n_kernel<1, 3, 5>(msiren::TrunkF16Params);
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
    beside = {k: v for k, v in usage.items() if k not in trunks}
    assert len(trunks) == 4 and len(beside) >= 7, list(usage)  # 32x32x16 (A/B reference) and 16x16x32 (default), sine / Morlet

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
