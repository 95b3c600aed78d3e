"""CPU: ISA lint of the shipped gfx950 code object (tools/isa_lint.py).

The dominant kernels feed their MFMAs from inline-asm `ds_read_b32` reads whose validity rests on counted `s_waitcnt lgkmcnt`
waits the compiler cannot see, and issue LDS-DMA through a hand-managed M0 (csrc/conv1d_mfma.hpp).  The GPU parity tests prove
today's binary; this test proves the PROPERTY on whatever binary was just built, so a hipcc bump or a register-pressure change
that slips a copy / spill of an in-flight register between a read and its wait fails here, on the CPU, before anything runs."""
from pathlib import Path
import sys

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import isa_lint  # noqa: E402


def _prog(lines):
    """[(mnemonic, operands)] -> instruction list with 4-byte addresses"""
    return [(0x100 + 4 * i, op, args) for i, (op, args) in enumerate(lines)]


CLEAN = [
    ("ds_read_b32", "v10, v1 offset:128"),           # group 0
    ("ds_read_b32", "v11, v2 offset:256"),
    ("ds_read_b32", "v12, v1 offset:384"),           # group 1
    ("ds_read_b32", "v13, v2 offset:512"),
    ("s_waitcnt", "lgkmcnt(2)"),                     # group 0 landed
    ("v_mfma_f32_32x32x2_f32", "v[20:35], v10, v11, v[20:35]"),
    ("s_waitcnt", "lgkmcnt(0)"),
    ("v_mfma_f32_32x32x2_f32", "v[20:35], v12, v13, v[20:35]"),
    ("s_endpgm", ""),
]


def test_lint_accepts_the_counted_wait_pattern():
    assert isa_lint.lint_function("clean", _prog(CLEAN)) == []


def test_lint_rejects_a_consumer_before_the_covering_wait():
    prog = list(CLEAN)
    prog[4] = ("s_waitcnt", "lgkmcnt(3)")            # one read of group 0 may still be in flight
    bad = isa_lint.lint_function("early", _prog(prog))
    assert len(bad) == 1 and "v11" in bad[0] and "(R1)" in bad[0]


def test_lint_rejects_a_register_copy_of_an_in_flight_destination():
    prog = list(CLEAN)
    prog.insert(2, ("v_mov_b32_e32", "v40, v10"))    # what a careless register allocator could insert
    bad = isa_lint.lint_function("copy", _prog(prog))
    assert bad and "v_mov_b32_e32" in bad[0]


def test_lint_follows_the_control_flow_graph():
    # the reads are issued in a block placed BEHIND their consumer (as hipcc lays out `if (issue_next) dma_chunk()`):
    # block A (0x100): jump to C;  block B (0x104..): wait + MFMA + end;  block C: reads, branch back to B
    prog = [
        ("s_branch", "3"),                                   # 0x100 -> 0x110
        ("s_waitcnt", "lgkmcnt(1)"),                         # 0x104  (B)
        ("v_mfma_f32_32x32x2_f32", "v[20:35], v10, v11, v[20:35]"),
        ("s_endpgm", ""),
        ("ds_read_b32", "v10, v1 offset:128"),               # 0x110  (C)
        ("ds_read_b32", "v11, v2 offset:256"),
        ("s_branch", "65530"),                               # -> 0x104
    ]
    bad = isa_lint.lint_function("cfg", _prog(prog))
    assert len(bad) == 1 and "v11" in bad[0]                 # lgkmcnt(1) leaves the second read in flight
    prog[1] = ("s_waitcnt", "lgkmcnt(0)")
    assert isa_lint.lint_function("cfg", _prog(prog)) == []


def test_lint_checks_the_m0_sequence_of_the_lds_dma():
    good = [
        ("s_mov_b32", "s9, m0"), ("s_mov_b32", "m0, s8"), ("s_nop", "0"),
        ("global_load_lds_dwordx4", "v[66:67], off"), ("s_mov_b32", "m0, s9"), ("s_endpgm", ""),
    ]
    assert isa_lint.lint_function("dma", _prog(good)) == []
    no_nop = [g for g in good if g[0] != "s_nop"]
    assert any("(R4)" in b for b in isa_lint.lint_function("dma", _prog(no_nop)))
    no_restore = list(good)
    no_restore[4] = ("s_mov_b32", "m0, s10")
    assert any("(R4)" in b for b in isa_lint.lint_function("dma", _prog(no_restore)))
    stray = good[:5] + [("s_mov_b32", "m0, s3")] + good[5:]
    assert any("outside the LDS-DMA sequence" in b for b in isa_lint.lint_function("dma", _prog(stray)))
    burst = [
        ("s_mov_b32", "s9, m0"), ("s_mov_b32", "m0, s8"),
        ("s_nop", "0"), ("global_load_lds_dwordx4", "v[66:67], off"), ("s_add_u32", "m0, m0, 0x1000"),
        ("s_nop", "0"), ("global_load_lds_dwordx4", "v[68:69], off"), ("s_add_u32", "m0, m0, 0x1000"),
        ("s_nop", "0"), ("global_load_lds_dwordx4", "v[70:71], off"),
        ("s_mov_b32", "m0, s9"), ("s_endpgm", ""),
    ]
    assert isa_lint.lint_function("burst", _prog(burst)) == []
    no_nop2 = [b for k, b in enumerate(burst) if k != 5]                     # second piece without its wait state
    assert any("(R4)" in b for b in isa_lint.lint_function("burst", _prog(no_nop2)))
    unrestored = burst[:10] + burst[11:]
    assert any("(R4)" in b for b in isa_lint.lint_function("burst", _prog(unrestored)))


def test_shipped_library_passes_the_isa_lint():
    from multimodal_vqvae_compression_audio_tactile_amd import _lib
    if not _lib.SO_PATH.exists():
        pytest.fail(f"{_lib.SO_PATH} has not been built")
    bad, summary = isa_lint.run(_lib.SO_PATH)
    assert not bad, "\n".join(bad[:20])
    # not vacuous: the LDS-DMA kernels with hand-placed operand reads are in the binary and were all inspected
    assert summary["kernels"] >= 100 and summary["asm_read_kernels"] >= 40, summary
    assert summary.get("metadata_checked", 0) >= summary["kernels"] - 5, summary          # R3 looked at the metadata of every kernel
