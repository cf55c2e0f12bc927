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


def _regs(tok, agprs=False):
    """VGPR numbers named by one operand token ('v12', 'v[8:11]', 'v[0xc0:0xc3]', '-v7', 'v[0xe0]'); with agprs=True also the
    accumulation registers ('a7', 'a[16:19]'), numbered from 1000."""
    tok = tok.strip().lstrip("-|")
    m = re.fullmatch(r"([va])\[(0x[0-9a-f]+|\d+)(?::(0x[0-9a-f]+|\d+))?\]\|?", tok)
    if m:
        if m.group(1) == "a" and not agprs:
            return []
        base = 1000 if m.group(1) == "a" else 0
        lo = int(m.group(2), 0)
        hi = int(m.group(3), 0) if m.group(3) else lo
        return list(range(base + lo, base + hi + 1))
    m = re.fullmatch(r"([va])(\d+)\|?", tok)
    if not m or (m.group(1) == "a" and not agprs):
        return []
    return [(1000 if m.group(1) == "a" else 0) + int(m.group(2))]


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


def _compile(tmp, tu, name, extra=()):
    src = tmp / f"{name}.hip"
    src.write_text(tu)
    asm = tmp / f"{name}.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-S", "--cuda-device-only", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"),
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-fno-slp-vectorize", *extra, str(src), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    return asm.read_text()


@pytest.fixture(scope="module")
def trunk_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = _compile(tmp_path_factory.mktemp("hz"), TU, "hz")
    kernels = re.split(r"\n(?=_ZN6msiren\d+siren_trunk_f16x3[nhw]_kernel)", text)[1:]
    assert len(kernels) == 7, len(kernels)
    return [(k.split(":")[0], k.split(".amdhsa_kernel")[0]) for k in kernels]


def test_wait_states_around_asm_issued_instructions(trunk_isa):
    total = 0
    for name, body in trunk_isa:
        bad, checked, closest = _scan(body, "f16x3w" in name)
        print(f"{name}: {checked} accesses to MFMA results checked, closest {closest} wait states behind the MFMA")
        assert not bad, name + "\n" + "\n".join(bad[:20])
        assert checked > 100, (name, checked)   # the scan did see the epilogue reading accumulators
        total += checked
    assert total > 3000


def _scan_loads_in_flight(body, agprs=False):
    """Every vector-memory operation that returns data into VGPRs must be covered by an s_waitcnt vmcnt BEFORE anything
    reads or overwrites its destination.  The compiler guarantees that for loads it issues itself; for loads issued
    through asm (the weight-stationary trunk's table, modulation and queue loads) it only holds while every such asm
    statement is followed by a wait that carries the destination as an operand -- otherwise the registers are dead on
    arrival and get handed on while the data is still in flight, which is how round 3's -DMSIREN_WS_ABL=15 build faulted
    (siren_trunk_f16x3w.hip.h, "Ablation builds").  Program order; vmcnt(N) retires all but the N youngest operations
    (in-order return on gfx9); an unconditional branch ends the path.  agprs=True tracks loads into the accumulation registers
    as well (the config-5 kernel's weight fragments) -- only meaningful where the code is laid out in execution order: the
    walk is linear, and the split-fp16 weight-stationary kernel jumps between slot bodies whose text order is not their
    execution order.  Returns (violations, loads seen)."""
    out, bad, seen = [], [], 0   # outstanding operations, oldest first: [line, text, set of destination VGPRs]
    for i, raw in enumerate(body.splitlines()):
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                out = out[len(out) - n:] if 0 < n < len(out) else ([] if n == 0 else out)
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            out = []
            continue
        is_vm_load = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")) and "_lds_" not in op
        # (the destination of a LATER load may overlap an earlier one's: loads return in order, the later data lands last)
        touched = set(r for o in (ops[1:] if is_vm_load else ops) for r in _regs(o, agprs=agprs))
        for rec in out:
            hit = touched & rec[2]
            if hit:
                bad.append(f"v{sorted(hit)[0]} touched by `{raw.strip()}` (line {i + 1}) while `{rec[1].strip()}` (line {rec[0] + 1}) may be in flight")
                rec[2] -= hit
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):   # (spill code counts on vmcnt like any other vector-memory operation)
            returns = is_vm_load or \
                      (op.startswith("global_atomic") and "sc0" in raw)   # (global_load_lds_*: the VGPR operand is the address)
            dst = set(_regs(ops[0], agprs=agprs)) if (returns and ops and ops[0].startswith(("v", "a") if agprs else "v")) else set()
            seen += bool(dst)
            out.append([i, raw, dst])
    return bad, seen


def test_no_load_destination_is_touched_while_the_load_is_in_flight(trunk_isa, tmp_path):
    for name, body in trunk_isa:
        bad, seen = _scan_loads_in_flight(body)
        print(f"{name}: {seen} VGPR-destination vector-memory operations, {len(bad)} touched before a covering wait")
        assert not bad, name + "\n" + "\n".join(bad[:10])
        assert seen >= 8, (name, seen)
    # the ablation builds of the weight-stationary trunk (timing only, never shipped) must stay safe to RUN as well
    ws = ('#include <hip/hip_runtime.h>\n#include "siren_trunk_f16x3w.hip.h"\n'
          "template __global__ void msiren::siren_trunk_f16x3w_kernel<0, 4, 0>(msiren::TrunkWsParams);\n")
    for abl in (7, 15):
        text = _compile(tmp_path, ws, f"abl{abl}", extra=(f"-DMSIREN_WS_ABL={abl}",))
        bad, seen = _scan_loads_in_flight(text.split(".amdhsa_kernel")[0])
        assert not bad, f"MSIREN_WS_ABL={abl}\n" + "\n".join(bad[:10])


# ---- VALU-written SGPR read by a vector-memory instruction: checked on the BUILT library, every kernel --------------------
def _sregs(tok):
    """SGPR numbers named by one operand token ('s7', 's[26:27]', 'vcc')."""
    tok = tok.strip()
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return [int(m.group(1))]
    return {"vcc": [106, 107], "vcc_lo": [106], "vcc_hi": [107]}.get(tok, [])


def _scan_valu_sgpr_to_vmem(body, need=5):
    """gfx9: an SGPR written by a VALU instruction (v_readlane_b32 / v_readfirstlane_b32: how the compiler restores a
    spilled pointer; VOP3 compares and carries) may be read by a vector-memory instruction only `need` wait states later.
    The compiler pads between instructions it issues itself, but not in front of an asm statement: the weight loads of
    siren_trunk_x1w.hip.h (asm, SGPR base) came directly behind such a restore in the fp16 instances of one build, read a
    stale base and the launch died with an aperture violation (found with rocgdb, `set amdgpu precise-memory on`).
    Program order, s_nop N counts N + 1.  Returns (violations, SGPR reads by vector-memory instructions seen)."""
    wrote, state, bad, seen = {}, 0, [], 0
    for i, raw in enumerate(body.splitlines()):
        line = raw.split("//")[0].split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]
        if op == "s_nop":
            state += int(ops[0], 0) + 1
            continue
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            for o in ops:
                for r in _sregs(o):
                    seen += 1
                    if r in wrote and state - wrote[r] - 1 < need:
                        bad.append(f"`{line}` (line {i + 1}) reads s{r} {state - wrote[r] - 1} wait states after a VALU wrote it")
        # round 5: the two other consumers of a VALU-written SGPR that need software wait states on gfx9 -- the lane select of
        # v_readlane_b32 / v_writelane_b32 (4) and VCC as read by v_div_fmas (4).  (An asm statement that takes a freshly
        # v_readfirstlane'd value as its lane select, or follows a compiler-issued VOP3 compare with v_div_fmas, gets no padding.)
        if op in ("v_readlane_b32", "v_writelane_b32") and len(ops) >= 3:
            for r in _sregs(ops[2]):
                if r in wrote and state - wrote[r] - 1 < 4:
                    bad.append(f"`{line}` (line {i + 1}) takes its lane from s{r} {state - wrote[r] - 1} wait states after a VALU wrote it")
        if op.startswith("v_div_fmas"):
            for r in (106, 107):
                if r in wrote and state - wrote[r] - 1 < 4:
                    bad.append(f"`{line}` (line {i + 1}) reads vcc {state - wrote[r] - 1} wait states after a VALU wrote it")
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            pass
        elif op.startswith("v_") and ops:
            for o in ops[:2] if op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc", "v_subb", "v_div_scale", "v_mad_u64", "v_mad_i64")) else ops[:1]:
                for r in _sregs(o):
                    wrote[r] = state
        elif op.startswith("s_") and ops and not op.startswith(("s_cmp", "s_cbranch", "s_waitcnt", "s_barrier", "s_branch", "s_bitcmp")):
            for r in _sregs(ops[0]):
                wrote.pop(r, None)   # (an SALU result is interlocked)
        state += 1
    return bad, seen


def _scan_lds_reads_in_flight(body):
    """The LDS side of _scan_loads_in_flight (round 5): a ds_read's destination may be read or overwritten only behind an
    s_waitcnt lgkmcnt that covers it.  The counter is shared by LDS operations (in order among themselves) and scalar loads
    (out of order): with a scalar load outstanding only lgkmcnt(0) proves anything -- which is what the compiler emits, for the
    operations it KNOWS about; an LDS read or scalar load inside an asm statement is invisible to it, and a counted wait it
    computed without them can be satisfied early.  Program order, every kernel of the built library.
    Returns (violations, LDS reads seen)."""
    out, bad, seen = [], [], 0   # outstanding lgkm operations, oldest first: [line, text, destination VGPRs, is_smem]
    for i, raw in enumerate(body.splitlines()):
        line = raw.split("//")[0].split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o.split() else o for o in ops]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                if n == 0:
                    out = []
                elif not any(r[3] for r in out) and n < len(out):
                    out = out[len(out) - n:]
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            out = []
            continue
        touched = set(r for o in ops for r in _regs(o))
        for rec in out:
            hit = touched & rec[2]
            if hit:
                bad.append(f"v{sorted(hit)[0]} touched by `{line}` (line {i + 1}) while `{rec[1].strip()}` (line {rec[0] + 1}) may be in flight")
                rec[2] -= hit
        if op.startswith("ds_"):
            returns = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in op
            dst = set(_regs(ops[0])) if returns and ops else set()
            seen += bool(dst)
            out.append([i, raw, dst, False])
        elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_sendmsg", "s_dcache")):
            out.append([i, raw, set(), True])
    return bad, seen


def test_lds_scanner_sees_a_fragment_read_used_behind_a_wait_that_does_not_cover_it():
    ok = "ds_read_b128 v[0:3], v9\nds_read_b128 v[4:7], v9 offset:16\ns_waitcnt lgkmcnt(1)\nv_add_f32 v8, v0, v1\ns_waitcnt lgkmcnt(0)\nv_add_f32 v8, v4, v8\n"
    assert _scan_lds_reads_in_flight(ok) == ([], 2)
    early = ok.replace("lgkmcnt(1)\nv_add_f32 v8, v0, v1", "lgkmcnt(1)\nv_add_f32 v8, v4, v1")
    assert len(_scan_lds_reads_in_flight(early)[0]) == 1
    # a scalar load in between (as an asm statement would issue it, unknown to the compiler): the counted wait proves nothing
    smem = ok.replace("ds_read_b128 v[4:7]", "s_load_dwordx2 s[0:1], s[2:3], 0x0\nds_read_b128 v[4:7]").replace("lgkmcnt(1)", "lgkmcnt(2)")
    assert len(_scan_lds_reads_in_flight(smem)[0]) == 1
    lane = "v_readfirstlane_b32 s4, v0\nv_readlane_b32 s5, v1, s4\n"
    assert len(_scan_valu_sgpr_to_vmem(lane)[0]) == 1 and _scan_valu_sgpr_to_vmem(lane.replace("\nv_readlane", "\ns_nop 3\nv_readlane"))[0] == []


def test_scanner_sees_a_restored_pointer_read_too_early():
    risky = "v_readlane_b32 s0, v232, 17\nv_readlane_b32 s1, v232, 18\nglobal_load_dwordx4 a[16:19], v186, s[0:1]\n"
    bad, seen = _scan_valu_sgpr_to_vmem(risky)
    assert seen == 2 and len(bad) == 2
    padded = risky.replace("global_load", "s_nop 4\nglobal_load")
    assert _scan_valu_sgpr_to_vmem(padded) == ([], 2)
    assert _scan_valu_sgpr_to_vmem("v_readfirstlane_b32 s4, v0\ns_add_u32 s4, s4, 16\nglobal_load_dword v1, v2, s[4:5]\n")[0] == []


def test_no_vector_memory_instruction_of_the_built_library_reads_a_freshly_valu_written_sgpr(tmp_path):
    """Every kernel of mri_inr_amd/libmsiren.so, as built (the gfx950 code object of its offload bundle, disassembled)."""
    import struct
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "mri_inr_amd", "libmsiren.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        pytest.skip("needs the built library and llvm-objdump")
    blob = open(lib, "rb").read()
    # one offload bundle per translation unit (the host units and the k_*.hip kernel units): every gfx950 code object is scanned
    text, at, nobj = "", blob.find(b"__CLANG_OFFLOAD_BUNDLE__"), 0
    assert at >= 0
    while at >= 0:
        (n,), pos = struct.unpack_from("<Q", blob, at + 24), at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tl].decode()
            pos += 24 + tl
            if "gfx950" in triple and size:
                co = tmp_path / f"co{nobj}.elf"
                co.write_bytes(blob[at + off:at + off + size])
                res = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True, timeout=600)
                assert res.returncode == 0, res.stderr[-2000:]
                text += "\n" + res.stdout
                nobj += 1
        at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", at + 24)
    assert nobj >= 1, "no gfx950 code object in the library"
    res = type("R", (), {"stdout": text})
    kernels = re.split(r"\n(?=[0-9a-f]{16} <[^>]+>:\n)", res.stdout)[1:]
    assert len(kernels) >= 50, len(kernels)
    total, names, lds_total, vm_total = 0, [], 0, 0
    for k in kernels:
        name = re.match(r"[0-9a-f]{16} <([^>]+)>:", k).group(1)
        body = "\n".join(re.sub(r"^\s*[0-9a-f]+:\s*", "", l) if False else l for l in k.splitlines()[1:])
        bad, seen = _scan_valu_sgpr_to_vmem(body)
        assert not bad, name + "\n" + "\n".join(bad[:8])
        total += seen
        names.append(name)
        # round 5: every kernel's LDS reads and VGPR-destination vector loads against the waits that cover them
        bad, seen = _scan_lds_reads_in_flight(body)
        assert not bad, name + "\n" + "\n".join(bad[:8])
        lds_total += seen
        if "siren_trunk_f16x3w_kernel" not in name:   # (its slot bodies are not laid out in execution order: checked from its own TU above)
            bad, seen = _scan_loads_in_flight(body)
            assert not bad, name + "\n" + "\n".join(bad[:8])
            vm_total += seen
    assert lds_total > 2000 and vm_total > 2000, (lds_total, vm_total)
    assert sum("latent_mods_f16x3_kernel" in nm for nm in names) >= 8 and sum("encoder_conv_f16x3_kernel" in nm for nm in names) == 1
    assert sum("siren_trunk_x1w_kernel" in nm for nm in names) == 8 and total > 5000, (len(names), total)


def test_config5_weight_stationary_kernel_waits_and_distances(tmp_path):
    """siren_trunk_x1w.hip.h issues its MFMAs and its weight loads (straight into a[0:255]) through asm and waits for the fragments
    with counted vmcnt, k-step by k-step: every load's destination must be covered by a wait before an MFMA reads it, and every
    accumulator must be old enough when the epilogue (plain C++, which the compiler cannot pad against asm MFMAs) reads it."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    tu = ('#include <hip/hip_runtime.h>\n#include "siren_trunk_x1w.hip.h"\n'
          "template __global__ void msiren::siren_trunk_x1w_kernel<1, 0, 1>(msiren::TrunkX1Params);\n"
          "template __global__ void msiren::siren_trunk_x1w_kernel<0, 1, 0>(msiren::TrunkX1Params);\n")
    text = _compile(tmp_path, tu, "x1w")
    kernels = re.split(r"\n(?=_ZN6msiren\d+siren_trunk_x1w_kernel)", text)[1:]
    assert len(kernels) == 2
    for k in kernels:
        name, body = k.split(":")[0], k.split(".amdhsa_kernel")[0]
        bad, seen = _scan_loads_in_flight(body, agprs=True)
        assert not bad, name + "\n" + "\n".join(bad[:10])
        assert seen >= 600, (name, seen)    # 64 fragment loads per (layer, N-pass) code instance, all into AGPRs
        bad, checked, closest = _scan(body, True)
        print(f"{name}: {seen} loads, {checked} reads of MFMA results, closest {closest} wait states")
        assert not bad and checked > 1000 and closest >= MIN_MFMA_TO_VALU, (name, checked, closest, bad[:5])
        bad, reads = _scan_valu_sgpr_to_vmem(body)
        assert not bad and reads > 500, (name, reads, bad[:5])
