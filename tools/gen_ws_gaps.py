#!/usr/bin/env python3
"""Writes mri_inr_amd/csrc/siren_trunk_f16x3w_gaps.hip.h: the epilogue schedule of the weight-stationary trunk as one macro
per MFMA gap (MSIREN_WS_GA_<region>_<mfma>, MSIREN_WS_GB_<region>_<mfma>), most of them empty.  The schedule is a flat list
of 192 gaps per slot (8 k-step regions x 24 MFMAs); written out by hand in macros it expanded to ~100 compile-time
conditionals per gap x 2304 gaps and the compiler ran out of memory.  Re-run after changing the schedule; the output is
committed."""
import os
import sys

PAIRS = "--pairs" in sys.argv   # both halves of a packed register in one statement (5 gaps per half accumulator instead of 7)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mri_inr_amd", "csrc", "siren_trunk_f16x3w_gaps.hip.h")
A, B = {}, {}


def put(tab, j, stmt, r7=False):
    assert 0 <= j < (192 if r7 else 168), j   # region 7 starts with the slot's barrier: no LDS image traffic may be scheduled there
    tab.setdefault(j, []).append(stmt)


def idx(t, hh):
    return f"[{t >> 1}][{2 * (t & 1) + hh}]"


# ---- normal slot: 4 piece pairs x 4 half accumulators x (S0 S1 H0 H1 - L0 L1), store behind each piece pair ----
def half_a(j0, t, g, hh, last=False):
    e = f"[{g}]{idx(t, hh)}"
    put(A, j0, f"MSIREN_WS_S0({t}, {g}, {hh})")
    put(A, j0 + 1, f"MSIREN_WS_S1({t}, {g}, {hh})")
    if PAIRS:
        put(A, j0 + 2, f"MSIREN_WS_MIXH01(ehu{e}, sv0_, em[{t}][{2 * hh}], sv1_, em[{t}][{2 * hh + 1}])")
        put(A, j0 + 4, f"MSIREN_WS_MIXL01{'_LAST' if last else ''}(elu{e}, sv0_, em[{t}][{2 * hh}], sv1_, em[{t}][{2 * hh + 1}], ehu{e})")
        return
    put(A, j0 + 2, f"MSIREN_WS_MIXH0(ehu{e}, sv0_, em[{t}][{2 * hh}])")
    put(A, j0 + 3, f"MSIREN_WS_MIXH1(ehu{e}, sv1_, em[{t}][{2 * hh + 1}])")
    put(A, j0 + 5, f"MSIREN_WS_MIXL0(elu{e}, sv0_, em[{t}][{2 * hh}], ehu{e})")
    put(A, j0 + 6, f"MSIREN_WS_MIXL1{'_LAST' if last else ''}(elu{e}, sv1_, em[{t}][{2 * hh + 1}], ehu{e})")


j = 8   # the first table reads are issued at gap 0
for g, u in ((0, 0), (1, 0), (0, 1), (1, 1)):
    for k, (t, hh) in enumerate(((2 * u, 0), (2 * u, 1), (2 * u + 1, 0), (2 * u + 1, 1))):
        half_a(j + 7 * k, t, g, hh, last=(k == 3))
    put(A, j + 28, f"MSIREN_WS_STORE_PIECE(ehu[{g}][{u}], elu[{g}][{u}], {u}, {g})")
    j += 29

# ---- normal slot, Morlet (ACT == 1): the same 16 half accumulators, 9 gaps each -- the ten instructions of sin(2 pi r) exp2(cg r^2)
# a pair of plain multiplies or ONE transcendental per gap (both free beside an MFMA: tools/mfma_gap_probe.hip) instead of all ten
# in one gap, the hi / lo splits as pairs
A1 = {}


def half_a1(j0, t, g, hh, last=False):
    e = f"[{g}]{idx(t, hh)}"
    put(A1, j0, f"MSIREN_WS_MO_M0({t}, {g}, {hh})")
    put(A1, j0 + 1, f"MSIREN_WS_MO_M1({t}, {g}, {hh})")
    put(A1, j0 + 2, "MSIREN_WS_MO_E0()")
    put(A1, j0 + 3, "MSIREN_WS_MO_E1()")
    put(A1, j0 + 4, f"MSIREN_WS_MO_S0({t}, {g}, {hh})")
    put(A1, j0 + 5, f"MSIREN_WS_MO_S1({t}, {g}, {hh})")
    put(A1, j0 + 6, "MSIREN_WS_MO_P()")
    put(A1, j0 + 7, f"MSIREN_WS_MIXH01(ehu{e}, sv0_, em[{t}][{2 * hh}], sv1_, em[{t}][{2 * hh + 1}])")
    put(A1, j0 + 8, f"MSIREN_WS_MIXL01{'_LAST' if last else ''}(elu{e}, sv0_, em[{t}][{2 * hh}], sv1_, em[{t}][{2 * hh + 1}], ehu{e})")


j = 8
for g, u in ((0, 0), (1, 0), (0, 1), (1, 1)):
    for k, (t, hh) in enumerate(((2 * u, 0), (2 * u, 1), (2 * u + 1, 0), (2 * u + 1, 1))):
        half_a1(j + 9 * k, t, g, hh, last=(k == 3))
    put(A1, j + 36, f"MSIREN_WS_STORE_PIECE(ehu[{g}][{u}], elu[{g}][{u}], {u}, {g})")
    j += 37

# ---- final slot ----
# (mw[t]: the final row of the modulation table = modulation x last_layer.weight, read at the start of region 0)
j = 8
for t in range(4):          # each column group's chain runs tile by tile, element by element: the canonical order
    for g in range(2):
        for hh in range(2):
            put(B, j, f"MSIREN_WS_S0({t}, {g}, {hh})")
            put(B, j + 1, f"MSIREN_WS_S1({t}, {g}, {hh})")
            put(B, j + 2, f"part[{g}] = __builtin_fmaf(sv0_, mw[{t}][{2 * hh}], part[{g}])")
            put(B, j + 3, f"part[{g}] = __builtin_fmaf(sv1_, mw[{t}][{2 * hh + 1}], part[{g}])")
            j += 4
put(B, j, "MSIREN_WS_RED0()"); put(B, j + 1, "MSIREN_WS_RED1()"); put(B, j + 2, "MSIREN_WS_RED2()")
put(B, j + 3, "MSIREN_WS_RED3()"); put(B, j + 4, "MSIREN_WS_RED4()")
j += 6
# layer-0 table loads: half 0 issued at the start of region 0, half 1 at the start of region 2 (4 loads each); the weight loads
# of a loading slot follow MFMAs 1,3,5,7,17,19,21,23 of every region
def younger_weight_loads(j_issue_region, j_use):
    n = 0
    for jj in range(24 * j_issue_region, j_use):
        if (jj % 24) in (1, 3, 5, 7, 17, 19, 21, 23):
            n += 1
    return n


put(B, j - 1, f"MSIREN_WS_RAW0_WAIT(FL, 0, {younger_weight_loads(0, j - 1) + 4}, 4)")   # half 1's 4 loads are younger too
for u, g in ((0, 0), (0, 1), (1, 0), (1, 1)):   # layer 0 of the next pass: hi halves, then lo halves, then the store
    if (u, g) == (1, 0):
        put(B, j - 1, f"MSIREN_WS_RAW0_WAIT(FL, 1, {younger_weight_loads(2, j - 1)}, 0)")
    for k in range(4):
        t, e = 2 * u + (k >> 1), 2 * (k & 1)
        if PAIRS:
            put(B, j + 2 * k, f"MSIREN_WS_MIXH01(l0h_[{k}], raw0[{t}][{g}][{e}], em0_[{t}][{e}], raw0[{t}][{g}][{e + 1}], em0_[{t}][{e + 1}])")
            put(B, j + 8 + 2 * k, f"MSIREN_WS_MIXL01{'_LAST' if k == 3 else ''}(l0l_[{k}], raw0[{t}][{g}][{e}], em0_[{t}][{e}], raw0[{t}][{g}][{e + 1}], em0_[{t}][{e + 1}], l0h_[{k}])")
            continue
        put(B, j + 2 * k, f"MSIREN_WS_MIXH0(l0h_[{k}], raw0[{t}][{g}][{e}], em0_[{t}][{e}])")
        put(B, j + 2 * k + 1, f"MSIREN_WS_MIXH1(l0h_[{k}], raw0[{t}][{g}][{e + 1}], em0_[{t}][{e + 1}])")
        put(B, j + 8 + 2 * k, f"MSIREN_WS_MIXL0(l0l_[{k}], raw0[{t}][{g}][{e}], em0_[{t}][{e}], l0h_[{k}])")
        put(B, j + 8 + 2 * k + 1, f"MSIREN_WS_MIXL1{'_LAST' if k == 3 else ''}(l0l_[{k}], raw0[{t}][{g}][{e + 1}], em0_[{t}][{e + 1}], l0h_[{k}])")
    put(B, j + 16, f"MSIREN_WS_STORE_PIECE(l0h_, l0l_, {u}, {g})")
    j += 17

# ---- B fragments of the next k-step: one ds_read_b128 per gap instead of four in a bunch at the region's start ----
# (four waves issue them at the same moment behind the slot's barrier; bunched, the LDS queue held every wave's MFMA stream
# for ~50 cycles per region.)  Order of need in the next region: hi g0 (MFMA 0), hi g1 (MFMA 1), lo g0 (8), lo g1 (9).
# Gaps with an even MFMA index (odd ones carry a loading slot's weight loads), free of epilogue work where the table has
# such gaps, else beside a plain (4-cycle) statement -- never beside a sine.
SPREAD = "--bunched" not in sys.argv


def place_reads(tab):
    for s in range(8):
        cand = [24 * s + i for i in (2, 4, 6, 8, 10, 12, 14, 16)]
        free = [g for g in cand if g not in tab]
        busy = ("_S0(" , "_S1(", "_MO_E", "RED", "STORE", "WAIT", "FIN")
        plain = [g for g in cand if g in tab and not any(any(b in st for b in busy) for st in tab[g])]
        # (Morlet: where a region has fewer than four such gaps, an odd gap without a transcendental takes a read -- normal
        #  slots that carry weight loads there are the fourth unit's only)
        odd = [24 * s + i for i in (9, 11, 13, 15) if not any(any(b in st for b in busy) for st in tab.get(24 * s + i, []))]
        chosen = sorted((free + plain + odd)[:4])
        assert len(chosen) == 4, (s, free, plain)
        for g, frag in zip(chosen, (0, 2, 1, 3)):
            tab.setdefault(g, []).append(f"MSIREN_WS_BREAD({s}, {frag})")


if SPREAD:
    place_reads(A)
    place_reads(A1)
    place_reads(B)

# region 7 of a final slot (behind the barrier): finish the prev final slot's output
put(B, 168, "MSIREN_WS_FIN0()", True); put(B, 174, "MSIREN_WS_FIN1()", True); put(B, 176, "MSIREN_WS_FIN2()", True); put(B, 178, "MSIREN_WS_FIN3()", True)

with open(OUT, "w") as f:
    f.write("// GENERATED by tools/gen_ws_gaps.py -- the epilogue schedule of siren_trunk_f16x3w.hip.h, one macro per MFMA gap\n"
            "// (region S = k-step, MFMA i of its 24): MSIREN_WS_GA_S_i for a normal slot, MSIREN_WS_GB_S_i for a final-layer slot.\n"
            "#pragma once\n")
    for s in range(8):
        for i in range(24):
            st0, st1 = A.get(24 * s + i, []), A1.get(24 * s + i, [])
            if st0 == st1:
                body = "; ".join(st0)
                f.write(f"#define MSIREN_WS_GA_{s}_{i}(FL) do {{ {body}; }} while (0)\n" if st0 else f"#define MSIREN_WS_GA_{s}_{i}(FL) do {{}} while (0)\n")
            else:
                b0, b1 = "; ".join(st0), "; ".join(st1)
                f.write(f"#define MSIREN_WS_GA_{s}_{i}(FL) do {{ if constexpr (ACT == 0) {{ {b0}; }} else {{ {b1}; }} }} while (0)\n")
    for s in range(8):
        for i in range(24):
            st = B.get(24 * s + i, [])
            body = "; ".join(st)
            f.write(f"#define MSIREN_WS_GB_{s}_{i}(FL) do {{ {body}; }} while (0)\n" if st else f"#define MSIREN_WS_GB_{s}_{i}(FL) do {{}} while (0)\n")
print("wrote", OUT, "gaps used: A", len(A), "A1 (Morlet)", len(A1), "B", len(B), "last A", max(A), "last A1", max(A1), "last B", max(B))
