"""
Static guard on the hot GEMM kernel's code generation (no GPU: hipcc cross-compiles gfx950 to assembly).

Parity tests cannot see a register spill, but a spill inside the k-loop of ``gi::gemm_img_kernel`` costs more than
most optimisations gain (a scratch reload is a vector-memory operation: ~500 cycles with every wave of the workgroup
idle, and behind an epilogue's stores it waits for all of them).  This pins what the round-2 tree achieves
(profiles/r02_gemm_img_isa.txt) so that a later change that makes hipcc spill again fails loudly.
"""
import os
import re
import subprocess

import pytest

from foldingdiff_amd import build as fbuild

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPI = {0: "GELU", 1: "LN", 2: "QK", 3: "VT", 4: "BIAS", 5: "QKV"}
# scratch bytes per work-item the production (non-PROF) instantiations may use: none since round 3 (four fragment buffers,
# row info through LDS, half-block residual reads through buffer descriptors)
MAX_SCRATCH = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0, 5: 0}


@pytest.fixture(scope="module")
def gemm_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "gemm_img.s"
    try:
        fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    cmd = [fbuild.find_hipcc(), "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include"),
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "gemm_img.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _kernels(asm):
    """{(epi, prof): body text} of the gemm_img_kernel instantiations, and their .amdhsa metadata."""
    bodies, meta = {}, {}
    # template arguments: <epilogue, swapped form, PROF (stamps), TAIL (slice-capable tile list)>; keys (epilogue, prof, tail)
    for m in re.finditer(r"^(_ZN4fdmi2gi15gemm_img_kernelILi(\d)ELb[01]ELb([01])ELi([01])EEEvNS_11GemmImgArgsE):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm,
                         re.S | re.M):
        bodies[(int(m.group(2)), int(m.group(3)), int(m.group(4)))] = m.group(5)
    for m in re.finditer(r"\.name:\s+_ZN4fdmi2gi15gemm_img_kernelILi(\d)ELb[01]ELb([01])ELi([01])EEEvNS_11GemmImgArgsE\n(.*?)\.wavefront_size",
                         asm, re.S):
        fields = dict(re.findall(r"\.(\w+):\s+(\d+)", m.group(4)))
        meta[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = {k: int(v) for k, v in fields.items()}
    return bodies, meta


def test_production_gemm_kernels_stay_within_their_scratch_budget(gemm_asm):
    bodies, meta = _kernels(gemm_asm)
    assert sorted(k[0] for k in meta if k[1] == 0 and k[2] == 0) == sorted(EPI), sorted(meta)
    assert sorted(k[0] for k in meta if k[1] == 0 and k[2] == 1) == sorted(EPI), sorted(meta)
    for (epi, prof, tail), md in sorted(meta.items()):
        if prof:
            continue  # the stamp-recording instantiations are debug builds
        assert md["vgpr_count"] <= 168, (EPI[epi], md)   # 10 waves per workgroup: three waves on a SIMD share 512 registers
        assert md["private_segment_fixed_size"] <= MAX_SCRATCH[epi], (EPI[epi], md)


def test_k_loops_hold_no_scratch_and_no_vector_memory_wait(gemm_asm):
    """The steady-state k-loop body (30 MFMAs, barrier, 6 MFMAs, backward branch) of every production instantiation:
    no scratch instruction, no s_waitcnt vmcnt (the compute waves have no vector-memory operation in flight there)."""
    bodies, _ = _kernels(gemm_asm)
    assert sorted(k[0] for k in bodies if k[1] == 0 and k[2] == 0) == sorted(EPI), sorted(bodies)
    for (epi, prof, tail), text in sorted(bodies.items()):
        if prof:
            continue
        lines = text.splitlines()
        loops = 0
        label_at = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        for i, ln in enumerate(lines):
            if "Inner Loop Header" not in ln:
                continue
            # the loop = the header block down to the first BACKWARD branch behind it, plus the latch block that branch
            # targets (hipcc rotates the k-loop: the latch sits in front of the header and falls through into it)
            j = next((k for k in range(i + 1, len(lines))
                      for m in [re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", lines[k])] if m and label_at.get(m.group(1), 1 << 30) <= i), None)
            if j is None:
                continue
            tgt = label_at[re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", lines[j]).group(1)]
            body = lines[tgt:j + 1]
            if sum("v_mfma" in b for b in body) < 36:
                continue   # (parameter / loader loops)
            loops += 1
            assert not any("scratch_" in b for b in body), (EPI[epi], [b for b in body if "scratch_" in b])
            waits = [b.strip() for b in body if re.search(r"s_waitcnt vmcnt\(\d+\)", b)]
            assert len(waits) == 0, (EPI[epi], waits)   # (round 2 tolerated one in the q|k / v^T kernels: the row info now lands in LDS)
        assert loops >= 1, EPI[epi]


def _vgprs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def wide_store_hazards(asm):
    """12- / 16-byte stores whose NEXT instruction is a VALU write of one of their data registers.

    Round 3 met this once (profiles/r03_gemm_notes.log): `v_xor_b32 v82, 64, v121` right behind
    `buffer_store_dwordx4 v[82:85], v121, s[16:19], s28 offen` stored wrong dwords on gfx950 -- hipcc pads the
    hazard only for stores without an SGPR offset.  The product code now computes addresses before its store
    groups and ends them with store_guard(); this is the net under it."""
    lines = [ln.strip() for ln in asm.splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    found = []
    for i, ln in enumerate(lines[:-1]):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)", ln)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        data = _vgprs(ops[0]) if m.group(1) == "buffer" else _vgprs(ops[1]) if len(ops) > 1 else set()
        nxt = re.match(r"(v_\w+)\s+([^,\s]+)", lines[i + 1])
        if nxt and not nxt.group(1).startswith("v_cmp") and (_vgprs(nxt.group(2)) & data):
            found.append((ln, lines[i + 1]))
    return found


def test_the_hazard_scanner_sees_the_pattern_it_is_for():
    bad = "buffer_store_dwordx4 v[82:85], v121, s[16:19], s28 offen\n\tv_xor_b32_e32 v82, 64, v121\n"
    good = "buffer_store_dwordx4 v[82:85], v121, s[16:19], s28 offen\n\ts_nop 0\n\tv_xor_b32_e32 v82, 64, v121\n"
    assert len(wide_store_hazards(bad)) == 1 and not wide_store_hazards(good)


@pytest.mark.parametrize("source", ["gemm_img", "gemm_ws", "gemm_ln_rows", "rowwise_img", "attention_img"])
def test_no_wide_store_is_followed_by_a_write_of_its_data_registers(source, tmp_path):
    flags = fbuild.PER_SOURCE_FLAGS.get(source, [])  # the flags the product build uses for this source
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / f"{source}.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include")] + flags + [
        "-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, f"{source}.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert wide_store_hazards(out.read_text()) == []


def test_the_benchmark_attention_kernel_is_spill_free(tmp_path):
    """attn_img_kernel<4, relative_key, table in LDS, 2 groups> -- every released configuration at L in 97..128 -- keeps its
    state in registers (round 4: 199 VGPRs after the band rework; a spill in the item loop is a vector-memory round trip
    in front of counted s_waitcnt vmcnt bookkeeping) and its 160 KiB of LDS."""
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / "attention_img.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include")] \
        + fbuild.PER_SOURCE_FLAGS.get("attention_img", []) \
        + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "attention_img.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    name = "_ZN4fdmi2ai15attn_img_kernelILi4ELb1ELb1ELi2ELb0ELb0ELb0EEEvNS_11AttnImgArgsE"
    m = re.search(r"\.name:\s+" + name + r"\n(.*?)\.wavefront_size", asm, re.S)
    assert m, "kernel not found"
    md = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(1))}
    assert md["private_segment_fixed_size"] == 0, md
    assert md["vgpr_count"] <= 256, md
    body = re.search(r"^" + name + r":[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M).group(1)
    assert "scratch_" not in body
    assert "v_pk_fma_f32" not in body and "v_pk_mul_f32" not in body and "v_pk_add_f32" not in body  # packed fp32 serializes with the matrix pipe


def test_the_weight_stationary_gemm_keeps_its_weights_in_registers(tmp_path):
    """gemm_ws_kernel holds a 32-column slice of W as 192 VGPRs for the whole launch, two waves per SIMD (256 registers each):
    a spill would put weights in scratch and a scratch reload (s_waitcnt vmcnt(0)) inside the counted copy ring."""
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / "gemm_ws.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include"),
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "gemm_ws.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    metas = re.findall(r"\.name:\s+(_ZN4fdmi2ws14gemm_ws_kernelILi\d+EEEvNS_11GemmImgArgsE)\n(.*?)\.wavefront_size", asm, re.S)
    assert len(metas) == 3, [m[0] for m in metas]   # GELU, q | k | v and plain bias
    for name, body in metas:
        md = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", body)}
        assert md["vgpr_count"] <= 256 and md["vgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, (name, md)
        text = re.search(rf"^{name}:[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M).group(1)
        n_mfma = text.count("v_mfma_f32_32x32x16_f16")
        assert n_mfma >= 72 and n_mfma % 72 == 0, (name, n_mfma)           # whole hand-placed MFMA phases (hipcc may clone the loop)
        assert "scratch_" not in text, name


def test_the_few_rows_layernorm_gemm_is_spill_free(tmp_path):
    """gemm_ln_rows_kernel<12 / 24>: three rotating weight buffers (144 VGPRs) beside 48 accumulator registers, one wave per SIMD."""
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / "gemm_ln_rows.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include"),
           "-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "gemm_ln_rows.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    metas = re.findall(r"\.name:\s+(_ZN4fdmi2lr19gemm_ln_rows_kernelILi\d+EEEvNS_11GemmImgArgsE)\n(.*?)\.wavefront_size", out.read_text(), re.S)
    assert len(metas) == 2, [m[0] for m in metas]   # K = 384 and K = 768
    for name, body in metas:
        md = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", body)}
        assert md["vgpr_count"] <= 512 and md["vgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, (name, md)


# ------------------------------------------------------------ seq_attn.hip (round 5)
@pytest.fixture(scope="module")
def seq_attn_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "seq_attn.s"
    try:
        fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    cmd = [fbuild.find_hipcc(), "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include"),
           "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "seq_attn.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def test_fused_attention_kernel_has_no_scratch_and_never_drains_the_weight_stream(seq_attn_asm):
    """The production instantiations of sa::seq_attn_kernel (d_model 384 and 192): one wave per SIMD holds the hidden state in
    registers, so a spill is a vector-memory load in the middle of the counted LDS-DMA stream (its vmcnt(0) drains the ring: the
    first versions lost 5-12 k cycles per head to sixteen spilled address registers, profiles/r05_seq_attn_notes.log).  Pinned:
    no scratch access and no s_waitcnt vmcnt(0) between the loop header and the loop's end, 294 MFMAs per (sequence, head)
    -- 216 projection + 78 attention -- in the d_model-384 loop, none of the attention's accumulators copied out of AGPRs."""
    asm = seq_attn_asm
    found = 0
    for m in re.finditer(r"^(_ZN4fdmi2sa15seq_attn_kernelILi(\d+)ELb0EEEvNS_11SeqAttnArgsE):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        nkt, body = int(m.group(2)), m.group(3)
        found += 1
        lines = body.splitlines()
        # the item loop = the LAST loop of the kernel (the first one uploads the bias to LDS)
        hdr = max(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l)
        label = re.match(r"(\.LBB\d+_\d+):", lines[hdr]).group(1)
        end = max(i for i, l in enumerate(lines) if re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", l))
        loop = lines[hdr:end + 1]
        # (a few values that only the code before and behind the loop uses -- the first item's projection, the last item's
        # attention -- are parked in scratch across it: once per launch, never inside the loop)
        assert not any("scratch_" in l for l in loop), nkt
        assert sum("scratch_" in l for l in lines) <= 32, nkt
        assert not any(re.search(r"s_waitcnt\s+vmcnt\(0\)", l) for l in loop), nkt
        n_mfma = sum("v_mfma_f32_32x32x16_f16" in l for l in loop)
        assert n_mfma == (18 * nkt + 78), (nkt, n_mfma)
        # 48 = the projected head's three accumulators; the rest are loop-invariant values hipcc parks in AGPRs (20 and 42 today)
        assert sum("v_accvgpr_read" in l for l in loop) <= 96, nkt
        assert not any("v_accvgpr_write" in l for l in loop), nkt
    assert found == 2
    for m in re.finditer(r"\.name:\s+_ZN4fdmi2sa15seq_attn_kernelILi(\d+)ELb0EEEvNS_11SeqAttnArgsE\n(.*?)\.wavefront_size", asm, re.S):
        fields = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(2))}
        assert fields["private_segment_fixed_size"] <= 64 and fields["vgpr_spill_count"] <= 12, fields


# ------------------------------------------------------------ seq_attn16.hip (round 6)
def test_sixteen_row_fused_attention_kernel_is_spill_free_and_keeps_its_weight_stream_counted(tmp_path):
    """s16::seq_attn16_kernel (d_model 384 and 192), the default fused projection + attention kernel since round 6: two waves per SIMD,
    256 registers each, 96 of them the hidden state.  A spill is a vector-memory load behind s_waitcnt vmcnt(0) in the middle of the
    counted LDS-DMA weight stream (every structure tried on the way that spilled lost the stream: profiles/r06_seq_attn16_notes.log).
    Pinned for the production instantiations: no scratch at all, <= 256 VGPRs, exactly the hand-written vector-memory waits inside
    the item loop (one vmcnt(0) per item in front of the K / V barrier, counted waits everywhere else), no packed fp32 arithmetic
    (it serializes with the partner wave's MFMAs), every contraction on v_mfma_f32_16x16x32_f16, the wide-store hazard scan clean."""
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / "seq_attn16.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include")] \
        + fbuild.PER_SOURCE_FLAGS.get("seq_attn16", []) \
        + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "seq_attn16.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert wide_store_hazards(asm) == []
    found = 0
    for m in re.finditer(r"^(_ZN4fdmi3s1617seq_attn16_kernelILi(\d+)ELb0EEEvNS_11SeqAttnArgsE):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        nkt, body = int(m.group(2)), m.group(3)
        found += 1
        assert "scratch_" not in body, nkt
        assert "v_accvgpr" not in body, nkt   # accumulators live in VGPRs: no copies out of the accumulator file
        assert not re.search(r"v_pk_(fma|mul|add)_f32", body), nkt
        assert "v_mfma_f32_32x32" not in body and body.count("v_mfma_f32_16x16x32_f16") >= 18 * nkt + 75, nkt
        # every vector-memory wait of the kernel is hand written: counted waits (2: the ctx stores, 3: the pieces of the next stage) and
        # vmcnt(0) only where nothing may be in flight -- the two table-fill loops, the prologue, the K / V barrier, the kernel's end
        # (d_model 192: three stages per head, the newest request IS the stage a top waits for: its tops are vmcnt(0) by design)
        waits = [int(x) for x in re.findall(r"s_waitcnt\s+vmcnt\((\d+)\)", body)]
        assert set(waits) <= {0, 2, 3}, (nkt, sorted(set(waits)))
        if nkt == 12:
            assert waits.count(0) <= 6, (nkt, waits)
        assert body.count("s_barrier") >= (nkt // 2 - 2) + 2, nkt
    assert found == 2
    for m in re.finditer(r"\.name:\s+_ZN4fdmi3s1617seq_attn16_kernelILi(\d+)ELb0EEEvNS_11SeqAttnArgsE\n(.*?)\.wavefront_size", asm, re.S):
        fields = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(2))}
        assert fields["private_segment_fixed_size"] == 0 and fields["vgpr_spill_count"] == 0 and fields["vgpr_count"] <= 256, fields


# ------------------------------------------------------------ ffn16.hip (round 6)
def test_fused_layer_tail_kernel_is_spill_free_and_keeps_its_weight_stream_counted(tmp_path):
    """ffn::ffn16_kernel (d_model 384 and 192; with and without BertSelfOutput in front): two waves per SIMD, <= 256 registers each
    (96 output accumulators + 48 of the stationary operand).  With both planes of the stationary operand in registers hipcc spilled
    the image and reloaded it behind vmcnt(0) in every group (profiles/r06_ffn16_notes.log): pinned -- no scratch, only the
    hand-written vector-memory waits (counted: 2 = the pieces of the next stage, 4 / 6 in the prologue; vmcnt(0) in the parameter
    fill, at a pass's end and the kernel's), every contraction on v_mfma_f32_16x16x32_f16 and as many of them as the arithmetic needs
    (3 per 16 x 16 x 32 tile product), no packed fp32 arithmetic beside the partner wave's matrix instructions, no LDS-DMA into the
    rows' lo plane (it arrived stale: the plane goes through registers)."""
    try:
        hipcc = fbuild.find_hipcc()
    except RuntimeError as e:
        pytest.skip(str(e))
    out = tmp_path / "ffn16.s"
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={fbuild.ARCH}", "-I", os.path.join(REPO, "include")] \
        + fbuild.PER_SOURCE_FLAGS.get("ffn16", []) + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(fbuild.CSRC, "ffn16.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert wide_store_hazards(asm) == []
    found = 0
    for m in re.finditer(r"^(_ZN4fdmi3ffn12ffn16_kernelILi(\d+)ELb([01])ELb0EEEvNS_7FfnArgsE):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        nkt, tail, body = int(m.group(2)), int(m.group(3)), m.group(4)
        found += 1
        assert "scratch_" not in body, (nkt, tail)
        assert "v_accvgpr" not in body, (nkt, tail)
        assert not re.search(r"v_pk_(fma|mul|add)_f32", body), (nkt, tail)
        # per pass: the first group's first dense once, the loop body (next group's first dense + this group's second) once, the last
        # group's second dense once: 3 x NKT steps of 12 MFMAs; TAIL: + NKT * NKT / 2 steps
        steps = 3 * nkt + (nkt * nkt // 2 if tail else 0)
        assert "v_mfma_f32_32x32" not in body and body.count("v_mfma_f32_16x16x32_f16") == 12 * steps, (nkt, tail)
        waits = [int(x) for x in re.findall(r"s_waitcnt\s+vmcnt\((\d+)\)", body)]
        # (everything but the stage tops' vmcnt(2): the parameter fill in front of the first barrier, the prologue, a pass's end, the kernel's)
        assert set(waits) <= {0, 1, 2, 3, 4, 5, 6}, (nkt, tail, sorted(set(waits)))
        assert waits.count(0) <= 5 and sum(1 for w in waits if w != 2) <= 12, (nkt, tail, waits)
        assert body.count("s_barrier") == 2 * (steps // 2) + 1, (nkt, tail)   # a stage top per two steps, at one of two places per wave group
        assert body.count("offen lds") == 3 * 2 + 2 * 2 * (steps // 2), (nkt, tail)   # the weight stream's pieces only
    assert found == 4
    for m in re.finditer(r"\.name:\s+_ZN4fdmi3ffn12ffn16_kernelILi(\d+)ELb[01]ELb0EEEvNS_7FfnArgsE\n(.*?)\.wavefront_size", asm, re.S):
        fields = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)", m.group(2))}
        assert fields["private_segment_fixed_size"] == 0 and fields["vgpr_spill_count"] == 0 and fields["vgpr_count"] <= 256, fields
