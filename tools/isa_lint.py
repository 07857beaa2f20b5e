#!/usr/bin/env python3
"""Static checks on the gfx950 ISA of the fused kernels (no GPU needed):

  python tools/isa_lint.py            # compiles csrc/*.hip to assembly under /tmp and reports, per kernel,
                                      #   * scratch (private memory) instructions  -- a dynamically indexed local array
                                      #   * `s_waitcnt vmcnt(0)` inside a loop of < 3000 lines -- a drain of the staging
                                      #     DMAs every trip (the compiler puts one in front of any vector load whose
                                      #     result crosses the loop: masks, bound rows)
                                      #   * VGPR / AGPR / spill counts
  python tools/isa_lint.py --loops lqr_dpp16 'Li0EEE'   # instruction mix of every loop of the kernels matching the regex

Both findings cost config 5 10 % and the masked headline step 12 % before they were removed (CHANGELOG.md 4.4a, 4.5)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc.pytorch_amd", "csrc")
# the two compilations of lqr_dpp16.hip, as csrc/Makefile builds them
FLAGS = {"lqr_dpp16": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_DPP16_NO_KKT"],
         "lqr_mfma40": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "nn_dynamics": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "lqr_mfma16": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         "lqr_dpp16_ring2": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_DPP16_NSTAGE=2", "-DMPC_DPP16_WITH_KKT", "-DMPC_KKT16_NSTAGE=2"],
         "lqr_dpp16_pad": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_DPP16_NSTAGE=2", "-DMPC_DPP16_NO_KKT", "-DMPC_DPP16_PAD"],
         "lqr_dpp16_padkkt": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_DPP16_NSTAGE=2", "-DMPC_DPP16_NO_KKT", "-DMPC_DPP16_PAD", "-DMPC_DPP16_PAD_KKT",
                              "-DMPC_KF_LDS_BYTES=36864"],
         "lqr_mfma40_kkt": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_KKT"],
         "lqr_mfma40_padkkt": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_SWEEP_NSTAGE=2", "-DMPC_MFMA40_PAD=4", "-DMPC_MFMA40_KKT"],
         "lqr_mfma40_pad16kkt": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_SWEEP_NSTAGE=2", "-DMPC_MFMA40_PAD=16", "-DMPC_MFMA40_KKT"],
         "lqr_mfma40_ring2": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_SWEEP_NSTAGE=2"],
         "lqr_mfma40_pad4": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_SWEEP_NSTAGE=2", "-DMPC_MFMA40_PAD=4"],
         "lqr_mfma40_pad16": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DMPC_MFMA40_SWEEP_NSTAGE=2", "-DMPC_MFMA40_PAD=16"]}
SOURCES = {"lqr_dpp16_ring2": "lqr_dpp16", "lqr_dpp16_pad": "lqr_dpp16", "lqr_dpp16_padkkt": "lqr_dpp16", "lqr_mfma40_kkt": "lqr_mfma40", "lqr_mfma40_padkkt": "lqr_mfma40", "lqr_mfma40_pad16kkt": "lqr_mfma40", "lqr_mfma40_ring2": "lqr_mfma40", "lqr_mfma40_pad4": "lqr_mfma40",
           "lqr_mfma40_pad16": "lqr_mfma40"}
FAST = ["lqr_dpp16", "lqr_dpp16_ring2", "lqr_dpp16_pad", "lqr_dpp16_padkkt", "lqr_mfma40", "lqr_mfma40_ring2", "lqr_mfma40_kkt", "lqr_mfma40_padkkt", "lqr_mfma40_pad16kkt", "lqr_mfma40_pad4", "lqr_mfma40_pad16", "lqr_mfma16", "lqr_tiny",
        "kkt_wave"]


def assembly(name, out="/tmp/isa_lint"):
    os.makedirs(out, exist_ok=True)
    import hashlib
    s = os.path.join(out, name + "_" + hashlib.md5(" ".join(FLAGS.get(name, [])).encode()).hexdigest()[:8] + ".s")   # (flags are part of the cache key)
    src = os.path.join(CSRC, SOURCES.get(name, name) + ".hip")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(s) or any(os.path.getmtime(d) > os.path.getmtime(s) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-w",
                               "-S", "--cuda-device-only"] + FLAGS.get(name, []) + ["-o", s, src])
    return open(s).read().split("\n")


def prefetch(names=None, workers=6):
    """Build the assembly of several compilations at once (a cold cache costs 15-25 s a compilation one after the other; the CPU test
    suite wants all of them)."""
    from concurrent.futures import ThreadPoolExecutor
    names = list(names or FAST)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(assembly, names))


def structure(lines):
    kernels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"\b(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < i:
            loops.append((labels[m.group(2)], i))
    return kernels, loops


_DMA_LOOPS = {}


def in_timestep_loop(lines, loops, i):
    """Is line i inside a loop over timesteps?  = the smallest enclosing loop that stages data (has an LDS-DMA in its
    body) and has no other staging loop nested inside it (the loop over line-search passes does)."""
    # (keyed by content, not by id(): a freed list's id is handed to the next one -- seen once the test suite assembled in parallel)
    key = (len(lines), hash(lines[len(lines) // 2]), hash(tuple(lines[:40])), len(loops))
    if key not in _DMA_LOOPS:
        _DMA_LOOPS[key] = [(a, b) for a, b in loops if any("_load_lds_" in x or (x.lstrip().startswith("buffer_load") and x.rstrip().endswith(" lds")) for x in lines[a:b])]
    dma = _DMA_LOOPS[key]
    around = sorted((b - a, a, b) for a, b in dma if a <= i <= b)
    if not around:
        return False
    _, a, b = around[0]
    return not any((c, d) != (a, b) and a <= c and d <= b and (d - c) < (b - a) - 50 for c, d in dma)


def enclosing_loop_has_children(lines, i):
    """LLVM annotates every block with its innermost loop ("in Loop: Header=BBx_y Depth=d") and every loop header with its
    child loops.  True when line i sits in a loop that has child loops: a loop over line-search PASSES around the
    timestep loops (a wait there is paid once per pass, not once per timestep)."""
    hdr = None
    for j in range(i, max(i - 400, 0), -1):
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=\d+", lines[j])
        if m:
            hdr = m.group(1)
            break
        m = re.match(r"^\.L(BB\d+_\d+):", lines[j])
        if m and j + 1 < len(lines) and "Loop Header" in "".join(lines[j:j + 3]):
            hdr = m.group(1)
            break
    if hdr is None:
        return False
    for j, l in enumerate(lines):
        if l.startswith(".L" + hdr + ":"):
            block = "".join(lines[j:j + 40])
            head = block.split("\n")[0]
            # the header's own comment lines follow its label until the first instruction
            k = j + 1
            while k < len(lines) and lines[k].lstrip().startswith(";"):
                if "Child Loop" in lines[k]:
                    return True
                k += 1
            return "Child Loop" in lines[j]
    return False


def short(k):
    return re.sub(r"^_ZN6mpclqr12_GLOBAL__N_1\d+", "", k)[:48]


def klass(op):
    for pre, c in (("v_mfma", "mfma"), ("v_fmac_f32_dpp", "fmac_dpp"), ("v_mov_b32_dpp", "mov_dpp"), ("v_accvgpr", "acc"),
                   ("v_readlane", "lane"), ("v_writelane", "lane"), ("v_", "valu"), ("ds_", "lds"), ("s_nop", "s_nop"),
                   ("s_waitcnt", "waitcnt"), ("global_load", "gload"), ("global_store", "gstore"), ("scratch_", "scratch"),
                   ("s_", "salu")):
        if op.startswith(pre):
            return c
    return "other"


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--loops":
        lines = assembly(sys.argv[2])
        kernels, loops = structure(lines)
        pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else ".")
        for a, b in loops:
            k = [n for s, n in kernels if s <= a][-1]
            if not pat.search(k) or b - a < 100:
                continue
            c = collections.Counter(klass(t.split()[0]) for t in (l.strip() for l in lines[a:b + 1])
                                    if t and t[0] not in ";." and not t.endswith(":"))
            print("%-40s lines %6d-%6d  %5d instr  %s" % (short(k), a, b, sum(c.values()), dict(c)))
        return 0
    bad = 0
    for name in FAST:
        lines = assembly(name)
        kernels, loops = structure(lines)
        per = collections.defaultdict(lambda: [0, 0])
        for i, l in enumerate(lines):
            ks = [n for s, n in kernels if s <= i]
            if not ks:
                continue
            if "scratch_" in l and not l.strip().startswith(";"):
                per[ks[-1]][0] += 1
            if "s_waitcnt vmcnt(0)" in l and "; counted" not in l:      # hand-written tail waits carry the marker
                if in_timestep_loop(lines, loops, i):
                    per[ks[-1]][1] += 1
        meta = dict(re.findall(r"\.name:\s+(\S+)[\s\S]*?\.vgpr_count:\s+(\d+)", "\n".join(lines)))
        for _, k in kernels:
            sc, dr = per[k]
            print("%-12s %-50s vgpr %4s  scratch ops %3d  vmcnt(0) in loops %3d" % (name, short(k), meta.get(k, "?"), sc, dr))
            if name in ("lqr_dpp16", "lqr_mfma40") and sc:
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
