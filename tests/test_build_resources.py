"""Build-time guard: the hot kernels must not spill.  vidi_amd/build.py keeps hipcc's per-kernel resource report
(-Rpass-analysis=kernel-resource-usage) next to each object; a spill has no functional symptom (a 100-VGPR spill in the GEMM body
cost 11 % of the prefill until a two-build A/B caught it), so it is checked here, on the CPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "vidi_amd", "csrc", "build")
HOT = {
    "gemm.resources.txt": ["gemm_kernel", "gemm_f32_kernel"],
    "gemv.resources.txt": ["gemv_kernel", "gemv_glu_kernel", "gemv_norm2_kernel"],
    "gemv_mfma.resources.txt": ["gemvm_kernel"],
    "gemm_w4_bf16.resources.txt": ["gemm_w4_kernel"],
    "gemm_w4_f16.resources.txt": ["gemm_w4_kernel"],
    "gemm_w4_modes.resources.txt": ["gemm_w4_kernel"],
    "gemm_w4_lnf.resources.txt": ["gemm_w4_kernel"],
    "gemm_w4_patch.resources.txt": ["gemm_w4_kernel"],
    "attn_self.resources.txt": ["attn_self_kernel"],
    "attn_self_rm.resources.txt": ["attn_self_rm_kernel"],
    "attn_cross.resources.txt": ["attn_cross_kernel", "attn_cross2_kernel", "attn_merge"],
    "attn_cross_rows.resources.txt": ["attn_cross_rows_kernel"],
    "attn_text.resources.txt": ["attn_text_kernel", "attn_text_decode_kernel", "rope_cache_kernel"],
    "rowops.resources.txt": ["norm_kernel", "resid_norm2_kernel", "resid_norm2_rows_kernel", "row_stats_kernel"],
    "elementwise.resources.txt": ["softcap_argmax_kernel"],
    "preproc.resources.txt": ["resize_h_u8_kernel", "resize_v_u8_norm_kernel"],
}


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.parametrize("report", sorted(HOT))
def test_hot_kernels_do_not_spill(report):
    path = os.path.join(BUILD, report)
    if not os.path.exists(path):
        import shutil
        if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
            pytest.skip("no resource report and no hipcc to produce one")
        from vidi_amd.build import build
        build(force=True, verbose=False)
    kernels = parse(path)
    assert kernels, f"no resource report in {path}"
    checked = 0
    for name, res in kernels.items():
        if not any(h in name for h in HOT[report]):
            continue
        # tile shapes that are never dispatched by default (tile_cfg 1, 2, 11) may spill; the persistent 4-wave kernel (every
        # instantiation), the 8-wave 256x256 16x16x32-MFMA kernel (tile_cfg 4) and the small-problem 128x128 one (tile_cfg 0) may not
        if "gemm_kernel" in name and not (re.search(r"Li256ELi256ELi2ELi4ELi2ELi\dELb[01]ELi6ELi16E7LabNoneEv10GemmParams$", name)
                                          or re.search(r"Li128ELi128ELi2ELi2ELi2ELi\dELb[01]ELi0ELi32E7LabNoneEv10GemmParams$", name)):
            continue
        checked += 1
        assert res.get("ScratchSize", 0) == 0 and res.get("VGPRs Spill", 0) == 0, (name, res)
        # SGPR spills go to VGPR lanes (v_writelane / v_readlane), not to memory.  The persistent GEMM keeps 17-65 scalars there (four
        # buffer descriptors, the next tile's coordinates, the epilogue's bases); in the ISA they are written in the per-tile scalar
        # section (8-40 v_writelane per tile switch) and the 128-MFMA K-loop bodies read back 0-7 of them per iteration (0 in the plain
        # epilogue instantiations) — i.e. < 1 % of a K = 1152 tile.  The ceiling keeps that from growing unnoticed; other kernels: none.
        limit = 72 if "gemm_w4_kernel" in name else (8 if ("gemm_kernel" in name or "attn_cross2_kernel" in name) else 0)    # (cross2: two parameter blocks)
        if "attn_cross_rows_kernel" in name and not name.endswith("ELi0EEv15AttnCrossParamsS1_i"):
            # the running-reference forms (MODE 1 / 2) park up to 9 scalars in VGPR lanes around their COLD re-reference path; the main loops
            # hold none (test_asm_matrix_instructions_... reads that back from the ISA).  The fixed-reference form (MODE 0, the BASELINE dtype): 0.
            limit = 12
        assert res.get("SGPRs Spill", 0) <= limit, (name, res.get("SGPRs Spill"), limit)
    assert checked > 0


def test_encoder_attention_occupancy():
    """the d = 72 and d = 64 instantiations run two query sets per wave in <= 256 registers (two waves per SIMD); the other head dims one
    set in <= 168 (three waves per SIMD) — a register more on either side halves / cuts the resident waves with no functional symptom"""
    path = os.path.join(BUILD, "attn_self_rm.resources.txt")
    if not os.path.exists(path):
        pytest.skip("no resource report")
    seen = 0
    for name, res in parse(path).items():
        m = re.search(r"attn_self_rm_kernelI\w+?Li(\d+)ELi(\d)E", name)
        if not m:
            continue
        d, qs = int(m.group(1)), int(m.group(2))
        assert qs == (2 if d in (72, 64) else 1), name
        assert res["Occupancy"] >= (2 if qs == 2 else 3), (name, res)
        assert res["LDS Size"] <= 65536, (name, res)
        seen += 1
    assert seen == 10                                    # 4 head dims x 2 dtypes + the prescaled-Q form of d = 72 x 2 dtypes


def test_transpose_read_destinations_untouched_until_the_wait():
    """vidi_amd/csrc/attn_self_rm.hip issues its V transpose reads (`ds_read_b64_tr_b16`) as inline asm and waits for them with an explicit
    `s_waitcnt lgkmcnt(0)` (tr_wait): the compiler believes the destinations are written at the asm statement.  The built object is
    disassembled and every instruction between a transpose read and the next lgkmcnt(0) wait is checked: none may name a pending
    destination register (a copy or spill there would move stale data without any functional symptom in most runs)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from vidi_amd.build import OBJ, build
    build(verbose=False)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "attn_self_rm.o")
        shutil.copy(os.path.join(OBJ, "attn_self_rm.o"), o)
        subprocess.run([objdump, "--offloading", o], check=True, capture_output=True, cwd=d)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        assert co, os.listdir(d)
        dis = subprocess.run([objdump, "-d", os.path.join(d, co[0])], check=True, capture_output=True, text=True).stdout
    reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")

    def regs(tok):
        out = set()
        for m in reg.finditer(tok):
            out.update([int(m.group(1))] if m.group(1) is not None else range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    pending, kern, reads, windows, viol = set(), None, 0, 0, []
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            assert not pending, f"{kern}: ends with transpose reads in flight"
            kern = m.group(1)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        if op == "ds_read_b64_tr_b16":
            parts = args.split(",")
            if regs(",".join(parts[1:])) & pending:
                viol.append((kern, line.strip()))
            windows += not pending
            pending |= regs(parts[0])
            reads += 1
        elif pending:
            if op == "s_waitcnt" and "lgkmcnt(0)" in args:
                pending = set()
            elif regs(args) & pending:
                viol.append((kern, line.strip()))
    assert reads >= 100 and windows >= 10, (reads, windows)            # the asm form is the one that was built
    assert not viol, viol[:5]


def test_asm_matrix_instructions_of_the_many_row_cross_attention_keep_their_distances():
    """vidi_amd/csrc/attn_cross_rows.hip issues its MFMAs from inline asm with pinned register files (the accumulators in AGPRs): the compiler
    sees opaque statements and inserts none of the wait states gfx940+ needs in software around matrix instructions, nor does it know that a
    copy of an accumulator into a VGPR reads a matrix result.  The kernel keeps those distances by construction; this test reads them back
    from the compiled ISA: the main loop must not touch the accumulator file with anything but the MFMAs (no v_accvgpr_* — a copy there means
    the compiler carries accumulators in VGPRs again: 300 extra VALU instructions per sub-tile and reads of in-flight results), no spills,
    and no non-matrix instruction may read an MFMA's result within the scanned window behind it."""
    import subprocess
    import sys
    import tempfile
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "vidi_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", csrc, os.path.join(csrc, "attn_cross_rows.hip"), "-o", out],
                       check=True, capture_output=True)
        src = open(out).read()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mfma_hazards as H
    found = 0
    for m in re.finditer(r"^(_Z22attn_cross_rows_kernel\w+):[^\n]*\n", src, re.M):
        body = src[m.end(): src.find(".Lfunc_end", m.end())]
        lines = body.split("\n")
        heads = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
        assert heads, "no loop found"
        for h in heads:                                        # (the two modalities' bodies are inlined: two loops per kernel)
            label = lines[h].split(":")[0].strip()
            ends = [i for i, l in enumerate(lines) if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\s*$", l) and i > h]
            assert ends, f"no back edge to {label}"
            loop = [l for l in lines[h: ends[-1] + 1] if l.strip() and not l.strip().startswith(";")]
            assert not [l for l in loop if "v_accvgpr" in l], f"{m.group(1)}: the main loop copies accumulator registers: {[l.strip() for l in loop if 'v_accvgpr' in l][:3]}"
            assert not [l for l in loop if "scratch_" in l], f"{m.group(1)}: spills inside the main loop"
            assert not [l for l in loop if "v_readlane" in l or "v_writelane" in l], f"{m.group(1)}: scalar spills (VGPR lanes) inside the main loop"
            assert sum("v_mfma" in l for l in loop) in (32, 16), "QK^T + PV of one sub-tile"
            # the key-padding mask rides in SGPRs (one s_load per sub-tile, a step ahead): an ordinary global load in the loop makes the compiler
            # drain the hand-counted K / V DMA ring with `s_waitcnt vmcnt(0)` in front of its use (the form up to round 5, on every product launch)
            plain_loads = [l.strip() for l in loop if re.search(r"\b(global|buffer|flat)_load_", l) and "lds" not in l]
            assert not plain_loads, f"{m.group(1)}: ordinary vector loads inside the main loop: {plain_loads[:3]}"
            assert not [l for l in loop if "vmcnt(0)" in l and "lgkmcnt" not in l] or sum("vmcnt(" in l for l in loop) > 1, "a lone full drain of the DMA ring inside the loop"
            assert sum("s_load_dwordx8" in l for l in loop) == 1, "one scalar mask load per sub-tile"
        n, _, _, min_reader, worst = H.scan(lines)
        assert n and min_reader >= 10, f"{m.group(1)}: an instruction reads a matrix result {min_reader} instructions behind its MFMA: {worst}"
        found += 1
    assert found == 10, "both head dims x (bf16: fixed reference, running reference with / without a cap; fp16: running reference with / without a cap)"
