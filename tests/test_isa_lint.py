"""Static checks on the gfx950 assembly of the two fused kernels (no GPU: hipcc cross-compiles).

Two compiler behaviours cost the kernels 10-15 % before they were designed out (CHANGELOG.md 4.1, 4.4a):
a select between two elements of a small local array turns the array into scratch memory, and a vector
load whose result crosses the timestep loop (a mask, a bound row) makes the compiler put
`s_waitcnt vmcnt(0)` -- a drain of the staging DMAs -- in front of every use.  tools/isa_lint.py finds both."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None,
                                reason="needs hipcc")


@pytest.fixture(scope="module", autouse=True)
def _assemblies():
    """Every compilation the checks below look at, assembled side by side (a cold cache took 3:45 one after the other)."""
    import isa_lint
    isa_lint.prefetch(sorted(set(isa_lint.FAST) | set(SGPR_SPILL_LIMITS) | {"nn_dynamics"}))


def _findings(name):
    import isa_lint
    lines = isa_lint.assembly(name)
    kernels, loops = isa_lint.structure(lines)
    out = {}
    for i, l in enumerate(lines):
        ks = [n for s, n in kernels if s <= i]
        if not ks:
            continue
        f = out.setdefault(ks[-1], {"scratch": 0, "drains": 0})
        if "scratch_" in l and not l.strip().startswith(";"):
            f["scratch"] += 1
        if "s_waitcnt vmcnt(0)" in l and "; counted" not in l:      # hand-written tail waits carry the marker
            # (a wait in a loop that has child loops is paid once per line-search pass, not per timestep)
            if isa_lint.in_timestep_loop(lines, loops, i) and not (name.startswith("lqr_dpp16") and isa_lint.enclosing_loop_has_children(lines, i)):
                f["drains"] += 1
    return out


def _metadata(name):
    """{kernel: {sgpr_spill_count, vgpr_spill_count, private_segment_fixed_size, vgpr_count}} from the code object notes"""
    import re
    import isa_lint
    text = "\n".join(isa_lint.assembly(name))
    out = {}
    for b in text.split("- .agpr_count")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", b)
        if nm:
            out[nm.group(1)] = {k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
                                for k in ("sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "vgpr_count")}
    return out


# Scalar-register spills per kernel, as built for round 3 (+10 % head room): a spilled SGPR is a v_writelane / v_readlane
# pair in the loops (CHANGELOG.md 8.7), cheap one by one, and the count crept from 162 to 486 over round 2 unnoticed.  A
# change that pushes a kernel past its line here has to look at where the new spills execute (tools/isa_lint.py --loops).
SGPR_SPILL_LIMITS = {
    # (round 5, kernelILi2E: 550 -> 690 and 315 -> 430.  The box-constrained line search gained a second way through its multi-trial
    # pass -- the four rows of a wavefront rolling out the remaining trials of its ONE row still searching, lane_as_slot in
    # lqr_dpp16_body.h -- and the scalars that choose between the two live across the inlined pass.  They execute once per
    # pass, not per timestep: tools/isa_lint.py --loops shows the timestep loops unchanged, and the first-iteration step measures
    # 152.4 us against 152.8 before, profiles/r05_ab_row_parallel_trials.log.  Letting two rows take turns through a loop around the
    # pass put the count at 1499 / 1084 for nothing measurable; that form was dropped.)
    "lqr_dpp16": {"kernelILi0E": 200, "kernelILi1E": 440, "kernelILi2E": 690, "kernelILi3E": 160, "kkt_fused": 8},
    "lqr_dpp16_ring2": {"kernelILi0E": 140, "kernelILi1E": 425, "kernelILi2E": 430, "kernelILi3E": 170, "lqr_kkt_dpp16": 0},
    # (round 6) the padded instantiation of the 12/4 kernel: 44 registers of gather maps a lane and eight wave-uniform block bases
    # (... and, with the gathers on shared LDS anchors + immediates, eight buffer descriptors that stay live across a stage: the spill
    # count went up by ~80 and the kernels got 2-10 % faster, profiles/r06_pad12_bench.log)
    "lqr_dpp16_pad": {"kernelILi0E": 430, "kernelILi1E": 715, "kernelILi2E": 945, "kernelILi3E": 655},
    # ... and of its fused KKT backward (the fourth compilation of lqr_dpp16.hip): the sweep's gather maps and block bases in pass 1, the
    # rollout's twelve F gathers in pass 2
    "lqr_dpp16_padkkt": {"kkt_fused": 150},          # (+20 with the gradient blocks leaving through LDS: the run pointers and lane masks of seven 16-byte stores)
    "lqr_mfma40": {"kernelILi0E": 90, "kernelILi1E": 105, "kernelILi2E": 150},
    # (round 4: block addresses as scalar arithmetic -- two more base pointers live in the two-slot build of mode 0)
    "lqr_mfma40_ring2": {"kernelILi0E": 100, "kernelILi1E": 105, "kernelILi2E": 150},
    "lqr_mfma40_kkt": {"kernelILi0E": 55, "kernelILi1E": 115},
    # (round 6) the padded fused backward, dword / 16-byte gathers: the gathers' descriptors and M0 values as in the padded step kernels.
    # (The masked kernel first assembled its pinned set from 8-24 scalar loads at the true n_ctrl, live across the inlined sweep: 364 spills
    # and 1.3x the unmasked kernel's time; from the staged record by one ballot -- kkt_pinned_lds -- it is 160 and 1.07x.)
    "lqr_mfma40_padkkt": {"kernelILi0E": 170, "kernelILi1E": 180},
    "lqr_mfma40_pad16kkt": {"kernelILi0E": 160, "kernelILi1E": 165},
    # the padded instantiation (round 4): every gather instruction wants a 128-bit descriptor and an M0 -- 47 of them a stage in the
    # dword build; the ceilings are what that costs in scalar registers (none of it in vector spills or scratch)
    "lqr_mfma40_pad4": {"kernelILi0E": 380, "kernelILi1E": 700, "kernelILi2E": 930},
    "lqr_mfma40_pad16": {"kernelILi0E": 380, "kernelILi1E": 700, "kernelILi2E": 930},
    "lqr_wave1": {"kernelILi1E": 20, "kernelILi2E": 12, "kernelILi3E": 28, "kernelILi4E": 20, "kernelILi5E": 20, "kernelILi6E": 20},
    # (round 5: the body is compiled for float and for double -- the record's granule map in elements costs the full float32
    # box-constrained instantiation three more scalars; the float64 instantiations carry 64-bit everything)
    "lqr_mfma16": {"mfma16_kernelILb1ELi0E": 35, "mfma16_kernelILb1ELi1E": 45, "mfma16_kernelILb1ELi2E": 92,
                   "mfma16_kernelILb0ELi0E": 215, "mfma16_kernelILb0ELi1E": 205, "mfma16_kernelILb0ELi2E": 355,      # (round 6: the QP's cold start on every timestep's path, +47 in the masked build)
                   "f64_kernelILb1ELi0E": 60, "f64_kernelILb1ELi1E": 80, "f64_kernelILb1ELi2E": 130,
                   "f64_kernelILb0ELi0E": 360, "f64_kernelILb0ELi1E": 365, "f64_kernelILb0ELi2E": 375},
}


@pytest.mark.parametrize("tu", sorted(SGPR_SPILL_LIMITS))
def test_no_vector_spills_no_scratch_and_bounded_scalar_spills(tu):
    """Every fused kernel: nothing in scratch memory, no spilled vector register, scalar spills under their recorded line."""
    md = _metadata(tu)
    seen = set()
    scratch_ops = {k: v["scratch"] for k, v in _findings(tu).items()}
    for k, v in md.items():
        assert v["vgpr_spill_count"] == 0, (k, v)
        # No instruction of the kernel may address scratch memory.  (The descriptor's private segment itself is allowed to be
        # non-zero only where that is provably dead weight: hipcc sometimes leaves the 68-byte frame of scalar spill slots it
        # went on to place in vector-register lanes -- seen on one or the other instantiation of the fused 32/8 backward,
        # flipping with unrelated edits; `-Rpass-analysis=kernel-resource-usage` reports it, the ISA has no scratch_ access.)
        assert v["private_segment_fixed_size"] == 0 or (tu == "lqr_mfma40_kkt" and v["private_segment_fixed_size"] <= 68) or \
            ("mfma16_f64" in k and v["private_segment_fixed_size"] <= 68), (k, v)
        assert scratch_ops[[n for n in scratch_ops if n in k or k in n][0]] == 0, k
        for pat, lim in SGPR_SPILL_LIMITS[tu].items():
            if pat in k:
                seen.add(pat)
                assert v["sgpr_spill_count"] <= lim, (k, v["sgpr_spill_count"], lim)
    assert seen == set(SGPR_SPILL_LIMITS[tu]), (seen, list(md))


@pytest.mark.parametrize("tu,kernels", [("lqr_dpp16", 8), ("lqr_dpp16_ring2", 5), ("lqr_dpp16_pad", 4), ("lqr_dpp16_padkkt", 4)])
def test_dpp16_kernels_keep_their_arrays_in_registers_and_their_dma_queue_full(tu, kernels):
    """Both compilations of lqr_dpp16.hip (csrc/Makefile): the 4-slot ring (step kernel modes 0..3 + the fused KKT
    backward kernels: register-resident gains up to T = 64 and the long-horizon one, each plain and masked) and the 2-slot one (the same four + the three-launch KKT gradient kernel)."""
    f = _findings(tu)
    assert len(f) == kernels
    for k, v in f.items():
        assert v["scratch"] == 0, k
        if "Li0E" in k or "Li3E" in k or "kkt" in k:            # headline kernels and the backward: no drain anywhere
            assert v["drains"] == 0, (k, v)


@pytest.mark.parametrize("tu,kernels", [("lqr_mfma40", 3), ("lqr_mfma40_ring2", 3), ("lqr_mfma40_kkt", 2), ("lqr_mfma40_padkkt", 2), ("lqr_mfma40_pad16kkt", 2)])
def test_mfma40_kernels_keep_their_arrays_in_registers_and_their_dma_queue_full(tu, kernels):
    """The three compilations of lqr_mfma40.hip (csrc/Makefile): the step kernels on the three-slot and on the two-slot sweep
    ring, and the fused KKT backward."""
    f = _findings(tu)
    assert len(f) == kernels
    for k, v in f.items():
        assert v == {"scratch": 0, "drains": 0}, (k, v)


@pytest.mark.parametrize("tu", ["lqr_dpp16", "lqr_dpp16_ring2", "lqr_dpp16_pad", "lqr_dpp16_padkkt"])
def test_register_resident_gains_own_the_accumulation_registers(tu):
    """Mode 0 of the headline kernel parks the gains of the whole horizon in a[0..255] through inline assembly
    (wv::rg_put / rg_get, lqr_dpp16.hip).  That is only sound while the compiler itself never allocates an AccVGPR in
    that kernel: every a-register access must be one of the hand-written v_accvgpr_write / v_accvgpr_read, and no MFMA
    may accumulate there."""
    import re
    import isa_lint
    lines = isa_lint.assembly(tu)          # (the unconstrained step runs on the 2-slot compilation)
    kernels, _ = isa_lint.structure(lines)
    # (the padded fused KKT backward's compilation holds no step kernel: its own register-resident kernel, the unmasked T <= 64 one)
    start = [i for i, n in kernels if ("kkt_fused_dpp16_kernelILb0E" if tu == "lqr_dpp16_padkkt" else "kernelILi0E") in n][0]
    end = min([i for i, n in kernels if i > start] + [len(lines)])
    body = [l for l in lines[start:end] if not l.strip().startswith(";")]
    acc = [l for l in body if re.search(r"\ba\[?\d", l)]
    assert len(acc) > 512                                   # the switches are there
    for l in acc:
        op = l.split()[0]
        assert op in ("v_accvgpr_write_b32", "v_accvgpr_read_b32"), l
    for l in body:
        if l.strip().startswith("v_mfma"):
            assert not re.search(r"\ba\[", l), l


def test_network_kernels_stay_in_registers():
    """csrc/nn_dynamics.hip: no scratch memory and no spilled vector registers in any instantiation; the register-resident
    kernels hold the network (both layers' operands for up to 8 hidden tiles), the kernel with the line search and
    the cost runs at the 512-register limit of one wave per SIMD -- a spill there would put the per-step operands in memory."""
    import re
    import isa_lint
    lines = isa_lint.assembly("nn_dynamics")
    text = "\n".join(lines)
    kernels, _ = isa_lint.structure(lines)
    names = [n for _, n in kernels]
    assert sum("nn_rollout_fast_kernel" in n for n in names) == 36        # 4 widths x (trajectory, + cost, line search) x 3 activations
    assert sum("nn_linearize_fast_kernel" in n for n in names) == 12
    assert sum("nn_rollout_kernel" in n for n in names) == 2 and sum("nn_linearize_kernel" in n for n in names) == 2
    assert not any("scratch_" in l and not l.strip().startswith(";") for l in lines)
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    assert len(spills) >= len(names) and all(v == 0 for v in spills)
    # every layer product is on the matrix core
    assert sum(l.strip().startswith("v_mfma_f32_16x16x4") for l in lines) > 500


def test_no_memory_instruction_sits_in_a_readfirstlane_loop():
    """A buffer instruction wants its descriptor in scalar registers.  Where the compiler has computed a wave-uniform address on the
    vector ALU (next to a loop counter it moved there) it does not complain: it wraps the instruction in a loop -- v_readfirstlane x 4,
    two compares, s_and_saveexec, the instruction, s_cbranch_execnz -- twelve instructions and a branch per gather or store.  The
    padded 32/8 kernels' dword build carried 25 of these a timestep from round 4 to round 6 (0.32 against 0.25 ms at 13/4); the cure
    is wv::uniform_ptr on the address.  No compilation of a fused kernel may contain one around a LOAD or a staging gather.  (Stores are
    not checked: the copy-out of a parked second trial in the padded 32/8 kernels addresses a different row per lane -- a descriptor per
    lane is what that loop is for; forcing it uniform there broke 42 of 2,500 fuzz cases within the hour.)"""
    import isa_lint
    for tu in isa_lint.FAST:
        lines = isa_lint.assembly(tu)
        hits = [i for i, l in enumerate(lines[:-1]) if "s_and_saveexec_b64" in l and
                any(op in lines[i + 1] for op in ("buffer_load", "_load_lds", "global_load")) and
                sum("v_readfirstlane_b32" in x for x in lines[max(i - 10, 0):i]) >= 2]
        assert not hits, (tu, len(hits), lines[hits[0] + 1].strip())

