"""CPU: an ISA-level check of the hand-placed scalar mask loads (csrc/common.hpp TileMasks).  TileMasks::load issues four
s_load_dwordx16 from inline asm and hands the 64 SGPRs back as ordinary values; the data only exists after the matching
s_waitcnt lgkmcnt(0) (TileMasks::wait).  The hardware does not interlock SMEM results, so any instruction that reads (or
copies, or spills) those registers in between would see stale bits.  This test compiles the kernels that use the masks to
gfx950 assembly and asserts that no instruction between an s_load_dwordx16 of TileMasks::load and the next lgkmcnt(0)
wait -- within the load's basic block: where the compiler would put a copy or a spill of the asm's outputs -- touches its
destination registers.  (The kernels also wait once more behind the key loop, so no load outlives it.)"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visiontransformer_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _sgprs(text):
    """SGPR numbers an operand string mentions."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(n) for n in re.findall(r"\bs(\d+)\b", text))
    return regs


@pytest.mark.parametrize("src,flags", [("attention_bf16.hip", ["-fno-slp-vectorize"]), ("attention_bwd_bf16.hip", [])])
def test_mask_word_sgprs_are_not_touched_before_their_wait(src, flags, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags,
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    loads = 0
    pending = set()          # SGPRs with a hand-issued SMEM load in flight
    lines = [l.split(";")[0].strip() for l in out.read_text().splitlines()]
    lines = [t for t in lines if t]
    for i, t in enumerate(lines):
        if t.startswith(".") or t.endswith(":"):
            if t.endswith(":"):
                pending = set()                      # a label: the check is per basic block (no control-flow analysis here)
            continue
        op, _, rest = t.partition(" ")
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            pending = set()
            continue
        if op == "s_load_dwordx16":
            # TileMasks::load = four s_load_dwordx16 in a row off one base (offsets 0x0, 0x40, 0x80, 0xc0); the compiler's
            # own scalar loads (kernel arguments) carry its own waits and are not this test's business
            block = [l for l in lines[max(0, i - 3):i + 4] if l.startswith("s_load_dwordx16")]
            if len(block) >= 4:
                pending |= _sgprs(rest.split(",")[0])
                loads += 1
            continue
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in rest:
                pending = set()
            continue
        hit = pending & _sgprs(rest)
        assert not hit, f"{src}: `{t}` touches s{sorted(hit)} while their s_load_dwordx16 is in flight"
    assert loads >= 8, loads     # the mask-word variants exist and were checked


def _asm(src, flags, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags,
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


@pytest.mark.parametrize("src,flags", [("gemm_h16p.hip", []), ("gemm_f32p.hip", ["-fno-slp-vectorize"]), ("gemm_p8.hip", []),
                                       ("gemm_f32s.hip", [])])
def test_persistent_gemms_do_not_spill(src, flags, tmp_path):
    """The persistent GEMMs order their LDS-DMA ring with hand-counted s_waitcnt vmcnt(N).  A register spill is a
    vector-memory instruction the counts do not know about, and hipcc's own wait for a reload ignores the DMA pieces in
    flight behind it: measured on gfx950, 80 scratch accesses per tile doubled a kernel's time.  So
    no kernel of these files may use scratch at all."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    text = _asm(src, flags, tmp_path)
    sizes = [int(m) for m in re.findall(r"; ScratchSize: (\d+)", text)]
    assert sizes and max(sizes) == 0, sizes
    assert "scratch_load" not in text and "scratch_store" not in text


@pytest.mark.parametrize("src,flags", [("gemm_h16p.hip", []), ("gemm_f32p.hip", ["-fno-slp-vectorize"]), ("gemm_p8.hip", []),
                                       ("attention_bf16.hip", ["-fno-slp-vectorize"]), ("attention_bwd_bf16.hip", []),
                                       ("preprocess.hip", [])])
def test_no_wide_buffer_store_with_scalar_offset(src, flags, tmp_path):
    """gfx950, seen on hardware (gemm_h16p.hip): a buffer_store_dwordx3/x4 whose soffset is an SGPR, followed directly by a
    VALU write of its data registers, stored the NEW register contents -- hipcc's hazard table exempts that form from the
    wait state it inserts for the other addressing forms.  The kernels therefore keep the whole offset in the VGPR operand;
    this test keeps the form out of the library."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    bad = [l.strip() for l in _asm(src, flags, tmp_path).splitlines()
           if re.match(r"\s*buffer_store_dwordx[34]\b", l) and re.search(r",\s*s\d+\s+offen", l.split(";")[0])]
    assert not bad, bad[:4]


def test_layernorm_backward_keeps_three_blocks_per_cu(tmp_path):
    """backward.hip sizes the LayerNorm-backward grid as 3 resident 256-thread blocks per CU (launch_layernorm_bwd); that
    holds while the <3, ...> instantiations stay within 168 registers (3 waves per SIMD) and use no scratch -- nothing at
    run time would report a kernel that grew past it, it would only run a third of its blocks in a second round."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    text = _asm("backward.hip", [], tmp_path)
    found = 0
    for name, body in re.findall(r"\.amdhsa_kernel (\S*layernorm_bwd_kernelILi3\S*)(.*?)\.end_amdhsa_kernel", text, re.S):
        nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        assert nv <= 168 and scratch == 0, (name, nv, scratch)
        found += 1
    assert found >= 1, "no layernorm_bwd_kernel<3, ...> instantiation found"


def test_small_attention_backward_fits_two_blocks_per_cu(tmp_path):
    """attention_bwd_small.hip: one launch, two co-resident 256-thread blocks per CU (__launch_bounds__(256, 2): 256 registers
    per lane, 66 KiB of LDS).  Its dk / dv half keeps 128 accumulator + operand registers across the loop; the loop-invariant
    LDS addresses are formed where they are used (opaque lane coordinates) so that they are not kept in ~50 more -- which
    spilled 26 registers to scratch.  Nothing at run time reports a spill; this does."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    text = _asm("attention_bwd_small.hip", [], tmp_path)
    body = re.search(r"\.amdhsa_kernel (\S*attn_bwd_small_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S).group(2)
    nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
    assert nv <= 256 and scratch == 0 and lds <= 80 * 1024, (nv, scratch, lds)
    assert "scratch_load" not in text and "scratch_store" not in text


def _vgprs(text):
    """VGPR numbers an operand string mentions (v7, v[4:7]; not a[..] / s[..])."""
    regs = set()
    for a, b in re.findall(r"(?<![\w\]])v\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(n) for n in re.findall(r"(?<![\w\]])v(\d+)\b", text))
    return regs


_VMEM = re.compile(r"^(buffer|global|flat|scratch)_")


def _hand_load_violations(text):
    """For every global/buffer load with a VGPR destination issued from INLINE ASM (hipcc neither counts it nor protects its
    destination): walk every control-flow path from the load to the first `s_waitcnt vmcnt(N)` that covers it (vector-memory
    instructions retire in order and vmcnt(N) leaves at most the N youngest outstanding, so the load has landed once N <= the
    number issued after it) and report any instruction on the way that reads, writes or copies the destination registers.  Returns (number of hand-issued loads checked, list of violations)."""
    lines, in_asm = [], []
    asm = False
    for raw in text.splitlines():
        t = raw.split(";;#")[0] if ";;#" not in raw else raw.strip()
        if ";;#ASMSTART" in raw:
            asm = True
            continue
        if ";;#ASMEND" in raw:
            asm = False
            continue
        t = raw.split(";")[0].strip()
        if not t or (t.startswith(".") and not t.endswith(":")):
            continue
        lines.append(t)
        in_asm.append(asm)
    label_at = {t[:-1]: i for i, t in enumerate(lines) if t.endswith(":")}
    checked, bad = 0, []
    for i, t in enumerate(lines):
        op, _, rest = t.partition(" ")
        if not (in_asm[i] and re.match(r"(global|buffer)_load_dword", op)) or " lds" in " " + rest:
            continue
        dest = _vgprs(rest.split(",")[0])
        if not dest:
            continue
        checked += 1
        seen = {}
        stack = [(i + 1, 0)]
        while stack:
            j, out = stack.pop()
            while j < len(lines):
                if seen.get(j, -1) >= out:
                    break
                seen[j] = out
                u = lines[j]
                if u.endswith(":"):
                    j += 1
                    continue
                uop, _, urest = u.partition(" ")
                if uop == "s_endpgm":
                    bad.append(f"`{t}` (line {i}): a path reaches s_endpgm without a covering wait")
                    break
                if uop == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", urest)
                    if m and int(m.group(1)) <= out:
                        break                       # data landed on this path
                    j += 1
                    continue
                hit = dest & _vgprs(urest)
                if hit:
                    bad.append(f"`{u}` touches v{sorted(hit)} while `{t}` (hand-issued, line {i}) is in flight")
                    break
                if _VMEM.match(uop):
                    out = min(out + 1, 64)
                if uop == "s_branch" or uop.startswith("s_cbranch"):
                    tgt = label_at.get(urest.strip())
                    if tgt is not None:
                        stack.append((tgt, out))
                    if uop == "s_branch":
                        break
                j += 1
    return checked, bad


@pytest.mark.parametrize("src,flags,least", [("gemm_h16p.hip", [], 8), ("gemm_f32p.hip", ["-fno-slp-vectorize"], 8)])
def test_hand_issued_loads_are_not_touched_before_their_wait(src, flags, least, tmp_path):
    """The GPU memory fault of round 3 (gpurun_out/h16p_where.txt): an inline-asm global_load whose destination was a
    temporary -- the compiler reused the registers before the data landed and the late write clobbered a pointer.  The
    kernels now load into the long-lived destination and name it "+v" in the wait statement, but nothing in the language
    stops a register-allocation change from putting a copy between load and wait again.  This is the guard: at ISA level,
    on every path, nothing touches a hand-issued load's destination before the vmcnt wait that covers it."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    checked, bad = _hand_load_violations(_asm(src, flags, tmp_path))
    assert checked >= least, checked
    assert not bad, bad[:6]


def test_hand_load_checker_catches_a_copy_and_a_missing_wait():
    """The checker itself, on hand-made streams: a copy of the destination before the wait, a wait that is too shallow
    (one younger load: vmcnt(1) covers, vmcnt(2) does not), a loop that carries the load across its back edge."""
    ok = """
    ;;#ASMSTART
    global_load_dwordx4 v[4:7], v[2:3], off
    ;;#ASMEND
    v_add_u32_e32 v8, v9, v10
    global_load_dword v11, v[2:3], off
    s_waitcnt vmcnt(1)
    v_mov_b32_e32 v12, v4
    s_endpgm
    """
    assert _hand_load_violations(ok) == (1, [])
    copy = ok.replace("v_add_u32_e32 v8, v9, v10", "v_mov_b32_e32 v8, v5")
    assert len(_hand_load_violations(copy)[1]) == 1
    assert len(_hand_load_violations(ok.replace("vmcnt(1)", "vmcnt(2)"))[1]) == 1
    loop = """
    .LBB0_1:
    s_waitcnt vmcnt(0)
    v_mov_b32_e32 v20, v6
    ;;#ASMSTART
    global_load_dwordx4 v[4:7], v[2:3], off
    ;;#ASMEND
    s_cbranch_scc1 .LBB0_1
    s_waitcnt vmcnt(0)
    s_endpgm
    """
    assert _hand_load_violations(loop) == (1, [])
    assert len(_hand_load_violations(loop.replace("vmcnt(0)\n    v_mov", "vmcnt(1)\n    v_mov"))[1]) == 1
    assert len(_hand_load_violations(loop.replace("s_waitcnt vmcnt(0)\n    s_endpgm", "s_endpgm"))[1]) == 1
