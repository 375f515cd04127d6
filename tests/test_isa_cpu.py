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


@pytest.mark.parametrize("src,flags", [("gemm_h16p.hip", []), ("gemm_f32p.hip", ["-fno-slp-vectorize"]), ("gemm_p8.hip", [])])
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
