"""CPU: the register budget of the hot kernels, read from the BUILT library (no GPU): the code objects inside libgptq_mi355x.so are unbundled with the
ROCm LLVM tools and their kernel metadata (llvm-readelf --notes) checked -- the decode-copy kernels must not touch scratch (a spill in a 5-microsecond kernel is
what the first strip-major version lost a round of sweeps to), the wide prefill kernel lives in all 512 registers of a lane and may spill a handful, never more.
Skipped where the LLVM tools or the library are missing (the product needs neither)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "autogptq_amd", "libgptq_mi355x.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels():
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(SO) or not all(os.path.exists(t) for t in tools):
        pytest.skip("built library or ROCm LLVM tools not present")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", SO, os.path.join(d, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in .hip_fatbin"
        for i, a in enumerate(starts):                      # one bundle per translation unit
            part = os.path.join(d, f"b{i}.bin")
            open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"co{i}.o")
            r = subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True).stdout
            for ent in re.split(r"\n  - ", notes):
                nm = re.search(r"\.name:\s+(\S+)", ent)
                if not nm:
                    continue
                get = lambda key: int(m.group(1)) if (m := re.search(rf"\.{key}:\s+(\d+)", ent)) else None
                out[nm.group(1)] = {"vgpr": get("vgpr_count"), "agpr": get("agpr_count"), "spill": get("vgpr_spill_count"), "scratch": get("private_segment_fixed_size"),
                                    "lds": get("group_segment_fixed_size")}
    return out


def test_hot_kernels_stay_inside_their_register_budget():
    ks = _kernels()
    tiled = {n: v for n, v in ks.items() if "gemv_tiled_kernel" in n}
    # plain (XM = 0), act-order (1) and tensor-parallel (3) forms x 3 packings x 3 row counts (+ the 5..8-row form of the plain and act-order kernels) x chunk
    # depths x 2 dtypes -- ONE compilation each since round 6 (the 8- and the 16-wave compilations of a form differed by a register or two: tiled_maxw)
    assert 200 <= len(tiled) <= 260, len(tiled)
    bad = {n: v for n, v in tiled.items() if v["spill"] or v["scratch"]}
    assert not bad, f"decode-copy kernels touching scratch: {list(bad.items())[:4]}"
    assert all(v["vgpr"] <= 128 for v in tiled.values())             # 16-wave workgroups: 4 waves per SIMD
    wide = {n: v for n, v in ks.items() if "gemm_wide_kernel" in n}
    assert len(wide) >= 7
    for n, v in wide.items():
        assert v["vgpr"] == 512 or (v["vgpr"] or 0) + (v["agpr"] or 0) == 512, (n, v)      # accumulators in the AGPR half
        assert (v["spill"] or 0) <= 4 and (v["scratch"] or 0) <= 32, (n, v)
    mid = {n: v for n, v in ks.items() if "gemm_mid_kernel" in n}
    assert mid and all((v["spill"] or 0) == 0 for v in mid.values())
    # round 5: the stream-K prefill kernel (every packing: 512 registers, no scratch) ...
    sk = {n: v for n, v in ks.items() if "gemm_wide_sk" in n}
    assert len(sk) >= 4 + 14, len(sk)                               # <T, G128> x 4 and <T, BITS, GM> x 14
    for n, v in sk.items():
        assert (v["vgpr"] or 0) <= 512 and (v["agpr"] or 0) == 256 and (v["spill"] or 0) == 0 and (v["scratch"] or 0) == 0, (n, v)      # (vgpr_count = both halves)
    # ... and the exchange-free batched-decode kernels: their weight / constant loads are inline asm whose results sit in registers across a hand-counted
    # s_waitcnt -- a spilled one would be stored before it has landed.  Every form the planner (or the lab knob) can launch is spill-free; the one geometry
    # that is not (8 bits, one row block, four strips) is refused by plan_rows.
    rows = {n: v for n, v in ks.items() if "gemm_rows_kernel" in n or "gemm_rows64_kernel" in n}
    assert len(rows) >= 144 + 24, len(rows)
    spilled = {n for n, v in rows.items() if (v["spill"] or 0) or (v["scratch"] or 0)}
    assert not spilled, sorted(spilled)[:4]      # (the one geometry that spilled -- <T, 8, 1, 4, GM> -- is no longer instantiated: round 6)


def test_no_kernel_of_the_library_touches_scratch():
    """Round 6 (review item: 175 spilling instantiations among 1,714, among them the defaults for fp32 layers, 2-bit and layers without a decode copy): the
    checkpoint-layout families were cut to what a plan can ask for, the per-k (raw act-order) form of gemv_generic_kernel walks a unit's values in a rolled loop
    instead of parking 32 dependent load chains in scratch, the streamed GEMVs keep their accumulators as one vector per row of x, and the spilling geometries of
    the skinny / stream64 / rows / matrix-core GEMV families are gone.  So the rule is no longer per family: NO kernel in the library spills or has a private
    segment -- whatever plan (default or forced) selects it -- except the 128 x 512 whole-round prefill kernel, which lives in all 512 registers of a lane by
    design and may spill a handful (<= 4 registers, <= 32 bytes).  Also guards the instantiation count and the library size against growing back."""
    ks = _kernels()
    bad = {n: (v["spill"], v["scratch"]) for n, v in ks.items() if ((v["spill"] or 0) or (v["scratch"] or 0)) and "gemm_wide_kernel" not in n}
    assert not bad, list(bad.items())[:6]
    assert len(ks) <= 1160, len(ks)                                  # 1,714 at the end of round 5; 1,153 once every decode-copy form is compiled once
    assert os.path.getsize(SO) <= 17 * 2 ** 20, os.path.getsize(SO)  # 19.9 MB at the end of round 5
    for fam in ("gemv_q4_f16_kernel", "gemv_q4_f16_direct_kernel"):      # round 1's comparison GEMVs (tuning.path = 2 / 4): retired
        assert not any(fam + "I" in n for n in ks), fam


def test_panel_kernels_hold_their_in_flight_loads_in_registers():
    """Round 6 (csrc/gemm_panel.hip): 8 waves per workgroup = two per SIMD = 256 registers per lane; the weight / constant loads are inline asm whose results sit in
    registers across a hand-counted s_waitcnt, so a spilled one would be stored before it has landed: every instantiated form is spill-free and scratch-free."""
    ks = _kernels()
    panel = {n: v for n, v in ks.items() if "gemm_panel_kernel" in n}
    assert len(panel) == 2 * (4 + 4 + 3) * 2, sorted(panel)          # <T, BITS, NT, G32>: fp16 / bf16 x (3 / 4 bits: NT = 1..4, 8 bits: NT = 1..3) x (groups of 64+, 32-wide groups)
    for n, v in panel.items():
        assert (v["vgpr"] or 0) + 0 <= 256 and (v["spill"] or 0) == 0 and (v["scratch"] or 0) == 0, (n, v)


def test_no_instruction_touches_a_register_whose_asm_load_is_in_flight():
    """The panel kernel's loads are inline asm behind hand-counted s_waitcnt: the compiler does not know they are loads and may copy a destination register before the
    data has landed (round 6: a v_mov at a control-flow join gave the 8-bit form outputs that differed from run to run).  The built code objects are disassembled and
    linted (tools/isa_inflight_lint.py): between a global_load and the next s_waitcnt vmcnt nothing reads or writes the load's destination."""
    import importlib.util
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(SO) or not all(os.path.exists(t) for t in tools):
        pytest.skip("built library or ROCm LLVM tools not present")
    spec = importlib.util.spec_from_file_location("isa_inflight_lint", os.path.join(ROOT, "tools", "isa_inflight_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    seen = rows_seen = 0
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", SO, os.path.join(d, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, a in enumerate(starts):
            chunk = blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)]
            if b"gemm_panel_kernel" not in chunk and b"gemm_rows" not in chunk:      # the two kernel families built on this technique
                continue
            part, co = os.path.join(d, f"b{i}.bin"), os.path.join(d, f"co{i}.o")
            open(part, "wb").write(chunk)
            r = subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            asm = subprocess.run([tools[2], "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            seen += len(re.findall(r"gemm_panel_kernel\w*>?:", asm))
            rows_seen += len(re.findall(r"gemm_rows(?:64)?_kernel\w*>?:", asm))
            for fam in ("gemm_panel_kernel", "gemm_rows"):
                bad = lint.lint(asm, fam)
                assert not bad, bad[:5]
    assert seen >= 44 and rows_seen >= 150, (seen, rows_seen)
