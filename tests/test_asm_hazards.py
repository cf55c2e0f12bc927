"""Wait states the compiler does not insert, checked in the ISA at build time (hipcc cross-compiles without a GPU).

hipcc pads hazards between instructions it knows; it does not look inside an `asm` statement
(/opt/skills/guides/cdna_hip_programming.md §5.7 item 2).  The split-fp16 trunks issue part of their epilogue -- and the
weight-stationary one all of its MFMAs -- through asm, so the distances that make those sequences legal are properties of the
instruction order the source asks for.  They are measured here on the compiled kernels:

  * an MFMA's result is read by a non-MFMA instruction (v_sin_f32, v_mov_b32, v_fma_mix*) only >= MIN_MFMA_TO_VALU wait
    states later (an 8-pass XDL op needs 12; the kernels are built so that an accumulator is a whole tile / seven MFMAs old);
  * a transcendental's result (v_sin_f32 issued through asm) is not read by the very next instruction;
  * v_fma_mixhi_f16 writes half a register: the instruction that reads the register is not the very next one.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
MIN_MFMA_TO_VALU = 12   # what an 8-pass XDL op needs (cdna_hip_programming.md §5.7 item 2); measured minimum today: 17 (f16x3n), see the printout

TU = r"""
#include <hip/hip_runtime.h>
#include "siren_trunk_f16x3n.hip.h"
#include "siren_trunk_f16x3h.hip.h"
#include "siren_trunk_f16x3w.hip.h"
template __global__ void msiren::siren_trunk_f16x3n_kernel<0, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3n_kernel<1, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3n_kernel<0, 4, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3h_kernel<0, 4, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3h_kernel<1, 3, 5>(msiren::TrunkF16Params);
template __global__ void msiren::siren_trunk_f16x3w_kernel<0, 4, 0>(msiren::TrunkWsParams);
template __global__ void msiren::siren_trunk_f16x3w_kernel<1, 4, 0>(msiren::TrunkWsParams);
"""


def _regs(tok):
    """VGPR numbers named by one operand token ('v12', 'v[8:11]', 'v[0xc0:0xc3]', '-v7', 'v[0xe0]')."""
    tok = tok.strip().lstrip("-|")
    m = re.fullmatch(r"v\[(0x[0-9a-f]+|\d+)(?::(0x[0-9a-f]+|\d+))?\]\|?", tok)
    if m:
        lo = int(m.group(1), 0)
        hi = int(m.group(2), 0) if m.group(2) else lo
        return list(range(lo, hi + 1))
    m = re.fullmatch(r"v(\d+)\|?", tok)
    return [int(m.group(1))] if m else []


def _scan(body, asm_mfma):
    """Walk one kernel's instructions in program order; returns (violations, number of checked accesses).  asm_mfma: the
    kernel issues its MFMAs through asm (every access to their results is checked); otherwise only the instructions issued
    through asm are (the compiler pads between instructions it knows, MFMA builtins included)."""
    mfma_at = {}        # vgpr -> wait-state index of the MFMA that wrote it last
    trans_at = {}       # vgpr -> index of the asm transcendental that wrote it
    half_at = {}        # vgpr -> index of the v_fma_mixhi that wrote (half of) it
    state, in_asm, bad, checked, closest = 0, False, [], 0, 10 ** 9
    for raw in body.splitlines():
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#ASM") else raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.endswith(":") or line.startswith("."):
            if line.endswith(":") and not asm_mfma:   # a label: other paths join here -- forget what this path knew
                mfma_at.clear(), trans_at.clear(), half_at.clear()
            # (weight-stationary kernel: a slot body reads the accumulators the body BEFORE it wrote; every body ends with the
            # same k-step and begins with the same reads, so the distance across the label in file order is the distance on
            # every path -- plus the slot-boundary code, which only lengthens it)
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]   # drop modifiers ("v5 op_sel:...")
        if op == "s_nop":
            state += int(ops[0], 0) + 1
            continue
        state += 1
        if not op.startswith(("v_", "ds_", "global_", "flat_", "buffer_")):
            continue
        is_load = op.startswith(("ds_read", "global_load", "buffer_load", "flat_load", "global_atomic"))
        dst = _regs(ops[0]) if ops and (op.startswith("v_") or is_load) else []
        srcs = [r for o in (ops[1:] if (op.startswith("v_") or is_load) else ops) for r in _regs(o)]
        if op.startswith("v_mfma"):
            for r in dst:
                mfma_at[r] = state
                trans_at.pop(r, None), half_at.pop(r, None)
            continue
        read_modify = op in ("v_fma_mixhi_f16",)   # writes half of dst: reads the other half
        look = asm_mfma or in_asm
        for r in (srcs + (dst if (read_modify or asm_mfma) else [])) if look else []:   # (asm MFMAs: overwriting counts too)
            if r in mfma_at:
                checked += 1
                closest = min(closest, state - mfma_at[r] - 1)
                if state - mfma_at[r] - 1 < MIN_MFMA_TO_VALU:
                    bad.append(f"{op} reads v{r} {state - mfma_at[r] - 1} states after the MFMA that wrote it: {raw.strip()}")
            if r in trans_at and state - trans_at[r] - 1 < 1:
                bad.append(f"{op} reads v{r} right behind the transcendental that wrote it: {raw.strip()}")
        for r in srcs:
            if r in half_at and state - half_at[r] - 1 < 1:
                bad.append(f"{op} reads v{r} right behind the v_fma_mixhi_f16 that wrote half of it: {raw.strip()}")
        for r in dst:
            mfma_at.pop(r, None)
            trans_at.pop(r, None)
            half_at.pop(r, None)
            if in_asm and op in ("v_sin_f32", "v_exp_f32"):
                trans_at[r] = state
            if op == "v_fma_mixhi_f16":
                half_at[r] = state
    return bad, checked, closest


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_wait_states_around_asm_issued_instructions(tmp_path):
    src = tmp_path / "hz.hip"
    src.write_text(TU)
    asm = tmp_path / "hz.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-S", "--cuda-device-only", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"),
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-fno-slp-vectorize", str(src), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    text = asm.read_text()
    kernels = re.split(r"\n(?=_ZN6msiren\d+siren_trunk_f16x3[nhw]_kernel)", text)[1:]
    assert len(kernels) == 7, len(kernels)
    total = 0
    for k in kernels:
        name = k.split(":")[0]
        body = k.split(".amdhsa_kernel")[0]
        bad, checked, closest = _scan(body, "f16x3w" in name)
        print(f"{name}: {checked} accesses to MFMA results checked, closest {closest} wait states behind the MFMA")
        assert not bad, name + "\n" + "\n".join(bad[:20])
        assert checked > 100, (name, checked)   # the scan did see the epilogue reading accumulators
        total += checked
    assert total > 3000
