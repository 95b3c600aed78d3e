#!/usr/bin/env python3
"""ISA lint of the shipped gfx950 code object (libmvq_hip.so) -- what keeps the hand-written pieces of the MFMA conv kernels
(csrc/conv1d_mfma.hpp: inline-asm `ds_read_b32` operand reads with counted `s_waitcnt lgkmcnt`, the LDS-DMA issue with its M0
save / write / restore) safe across a hipcc bump.  LLVM's waitcnt pass does not model loads issued from inline asm, so nothing
in the compiler would object if a future register allocator put a copy or a spill of an in-flight register between a read and
its wait; this script disassembles every kernel and proves, instruction by instruction, that it did not.

Checks (every kernel of the library unless noted):
  R1  no instruction reads or writes the destination of an LDS read (ds_read*) while that read may still be in flight.  Model
      (hardware-true for one wave): LDS operations complete in order; a counted `s_waitcnt lgkmcnt(N)` guarantees all but the
      N youngest LDS operations are done (an outstanding scalar load only makes the wait longer).  The in-flight queue is
      propagated over the kernel's control-flow graph (every distinct queue that can reach a basic block is simulated), so loops
      and the compiler's block placement are handled exactly; compiler-issued reads are judged by the same rule.
  R3  NO kernel of the library uses scratch (round 5: every kernel, not only the MFMA conv kernels): zero `scratch_` instructions,
      and in the code-object metadata a zero private segment and no VGPR spills.  A register array indexed at run time, or a
      register budget the allocator cannot meet, would otherwise turn into private-memory traffic silently.
  R4  every `global_load_lds_dwordx4` sits inside a burst of the exact shape
          s_mov_b32 sK, m0 / s_mov_b32 m0, sD / { s_nop 0 / global_load_lds_dwordx4 ... / [s_add_u32 m0, m0, imm] }+ / s_mov_b32 m0, sK
      and no other instruction of such a kernel writes M0.

Usage:  python tools/isa_lint.py [path/to/libmvq_hip.so] [--stats] [--kernel SUBSTR] [--dump DIR]
Exit status 0 = clean.  tests/test_isa_lint.py runs it in the CPU suite (`-m "not gpu"`).
"""
from __future__ import annotations

import argparse
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so: Path, arch: str = "gfx950") -> list[bytes]:
    """The device ELFs of every translation unit: .hip_fatbin is a concatenation of clang offload bundles."""
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(so)], check=True, capture_output=True)
        data = fat.read_bytes()
    out = []
    pos = data.find(BUNDLE_MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(BUNDLE_MAGIC))[0]
        p = pos + len(BUNDLE_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if triple.startswith("hip") and arch in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(BUNDLE_MAGIC, pos + 1)
    return out


def demangle(names: list[str]) -> list[str]:
    """objdump -C already demangles the function labels; the metadata names go through c++filt when it exists."""
    import shutil
    tool = shutil.which("c++filt") or (str(LLVM / "llvm-cxxfilt") if (LLVM / "llvm-cxxfilt").exists() else None)
    if not tool:
        return names
    out = subprocess.run([tool], input="\n".join(names) + "\n", check=True, capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")


def regs(text: str, rx, tag: str) -> set[str]:
    out = set()
    for m in rx.finditer(text):
        if m.group(1) is not None:
            out.add(f"{tag}{int(m.group(1))}")
        else:
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add(f"{tag}{i}")
    return out


def all_regs(text: str) -> set[str]:
    return regs(text, VREG, "v") | regs(text, SREG, "s") | regs(text, AREG, "a")


def disassemble(elf: bytes) -> tuple[dict[str, list[tuple[int, str, str]]], dict[str, dict]]:
    with tempfile.TemporaryDirectory() as td:
        f = Path(td) / "co.elf"
        f.write_bytes(elf)
        txt = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "-C", str(f)], check=True, capture_output=True, text=True).stdout
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
    funcs: dict[str, list[tuple[int, str, str]]] = {}
    cur = None
    for line in txt.splitlines():
        m = FUNC.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    # kernel metadata (msgpack rendered as YAML by llvm-readelf): name -> private segment size, spill counts
    meta: dict[str, dict] = {}
    blocks = re.split(r"\n\s+- \.agpr_count:", notes)
    for b in blocks[1:]:
        name = re.search(r"\.name:\s+(\S+)", b)
        if not name:
            continue
        meta[name.group(1)] = {
            "private": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)),
            "vgpr": int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)),
            "vgpr_spill": int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)),
            "sgpr_spill": int(re.search(r"\.sgpr_spill_count:\s+(\d+)", b).group(1)),
        }
    return funcs, meta


LGKM_WAIT = re.compile(r"lgkmcnt\((\d+)\)")


def branch_target(addr: int, args: str) -> int | None:
    """Branch target of an s_branch / s_cbranch_*: next address + simm16 * 4 (objdump prints the raw 16-bit immediate)."""
    m = re.search(r"(-?\d+|0x[0-9a-fA-F]+)\s*$", args)
    if not m:
        return None
    v = int(m.group(1), 0)
    if 0x8000 <= v <= 0xFFFF:
        v -= 0x10000
    return addr + 4 + 4 * v


def lint_function(name: str, insns: list[tuple[int, str, str]]) -> list[str]:
    """R1 (data flow over the control-flow graph) and R4 on one function; returns violations."""
    bad: list[str] = []
    index = {addr: i for i, (addr, _, _) in enumerate(insns)}
    # ---- basic blocks: a block ends behind every branch / s_endpgm and starts at every branch target
    leaders = {0}
    for i, (addr, op, args) in enumerate(insns):
        if op.startswith(("s_branch", "s_cbranch")):
            t = branch_target(addr, args)
            if t in index:
                leaders.add(index[t])
            if i + 1 < len(insns):
                leaders.add(i + 1)
        elif op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64") and i + 1 < len(insns):
            leaders.add(i + 1)
    order = sorted(leaders)
    block_end = {b: (order[k + 1] if k + 1 < len(order) else len(insns)) for k, b in enumerate(order)}
    parsed = [all_regs(args) for _, _, args in insns]

    def run_block(b: int, q: tuple) -> tuple[tuple, list[int]]:
        """simulate one block from the in-flight queue q (oldest first: (destination registers, address)); returns the queue
        at its end and the successor blocks"""
        q = list(q)
        end = block_end[b]
        for i in range(b, end):
            addr, op, args = insns[i]
            if op == "s_waitcnt":
                m = LGKM_WAIT.search(args)
                n = None
                if m:
                    n = int(m.group(1))
                elif not re.search(r"vmcnt|expcnt", args):     # raw immediate form: lgkmcnt = bits 11:8
                    try:
                        n = (int(args, 0) >> 8) & 0xF
                    except ValueError:
                        n = None
                if n is not None:
                    q = q[max(0, len(q) - n):] if n > 0 else []
                continue
            if op == "s_waitcnt_lgkmcnt":
                q = []
                continue
            if q:
                touched = parsed[i]
                for dst, a0 in q:
                    hit = touched & dst
                    if hit:
                        msg = (f"{name}: {addr:#x} `{op} {args}` touches {sorted(hit)} while the LDS read issued at {a0:#x} may "
                               "still be in flight (R1)")
                        if msg not in seen_msgs:
                            seen_msgs.add(msg)
                            bad.append(msg)
            if op.startswith("ds_"):
                dst = frozenset()
                if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in op:
                    first = args.split(",")[0]
                    dst = frozenset(regs(first, VREG, "v") | regs(first, AREG, "a"))
                q.append((dst, addr if dst else 0))            # stores only count; collapsing them keeps the state space small
                q = q[-15:]                                     # lgkmcnt is a 4-bit counter: at most 15 operations are outstanding, the
                                                                # 16th cannot issue before the oldest has returned (LLVM's SIInsertWaitcnts
                                                                # relies on the same bound: it never waits for an operation 15 issues back)
        addr, op, args = insns[end - 1]
        succ = []
        if op == "s_branch":
            t = branch_target(addr, args)
            if t in index:
                succ.append(index[t])
        elif op.startswith("s_cbranch"):
            t = branch_target(addr, args)
            if t in index:
                succ.append(index[t])
            if end < len(insns):
                succ.append(end)
        elif op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            pass
        elif end < len(insns):
            succ.append(end)
        return tuple(q), succ

    seen_msgs: set[str] = set()
    seen_states: dict[int, set] = {}
    work = [(0, ())]
    steps = 0
    while work:
        b, q = work.pop()
        if q in seen_states.setdefault(b, set()):
            continue
        seen_states[b].add(q)
        steps += 1
        if steps > 200000:
            bad.append(f"{name}: data-flow analysis did not converge (R1 unverified)")
            break
        q2, succ = run_block(b, q)
        for sb in succ:
            work.append((sb, q2))
    # ---- R4: the LDS-DMA bursts.  A burst is
    #     s_mov_b32 sK, m0 / s_mov_b32 m0, sD / { s_nop / global_load_lds_dwordx4 / [s_add_u32 m0, m0, imm] }+ / s_mov_b32 m0, sK
    # every LDS-DMA instruction and every write of M0 of such a kernel must lie inside one.
    is_dma = lambda op, args: op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in args)
    writes_m0 = lambda op, args: op.startswith("s_") and args.replace(" ", "").startswith("m0,")
    covered: set[int] = set()
    i = 0
    while i < len(insns):
        addr, op, args = insns[i]
        m = re.fullmatch(r"(s\d+), m0", args.strip()) if op == "s_mov_b32" else None
        if m and i + 1 < len(insns) and insns[i + 1][1] == "s_mov_b32" and insns[i + 1][2].replace(" ", "").startswith("m0,"):
            keep = m.group(1)
            j = i + 2
            ok, n = True, 0
            while ok:
                if j + 1 < len(insns) and insns[j][1] == "s_nop" and is_dma(insns[j + 1][1], insns[j + 1][2]):
                    n += 1
                    j += 2
                    if j < len(insns) and insns[j][1] == "s_add_u32" and insns[j][2].replace(" ", "").startswith("m0,m0,"):
                        j += 1
                        continue
                    break
                ok = False
            if ok and n > 0 and j < len(insns) and insns[j][1] == "s_mov_b32" and insns[j][2].replace(" ", "") == f"m0,{keep}":
                covered.update(range(i, j + 1))
                i = j + 1
                continue
        i += 1
    has_dma = any(is_dma(op, args) for _, op, args in insns)
    for i, (addr, op, args) in enumerate(insns):
        if i in covered:
            continue
        if is_dma(op, args):
            ctx = "; ".join(f"{o} {a}" for _, o, a in insns[max(0, i - 3):i + 2])
            bad.append(f"{name}: LDS-DMA at {addr:#x} is not inside a save-M0 / (write-M0, s_nop, DMA)+ / restore-M0 burst: [{ctx}] (R4)")
        elif has_dma and writes_m0(op, args):
            bad.append(f"{name}: {addr:#x} `{op} {args}` writes M0 outside the LDS-DMA sequence (R4)")
    return bad


VALU_PREFIX = ("v_",)


def stats(insns: list[tuple[int, str, str]]) -> dict:
    s = {"total": len(insns), "mfma": 0, "valu": 0, "ds_read": 0, "ds_write": 0, "vmem": 0, "dma": 0, "scratch": 0, "salu": 0, "waitcnt": 0, "barrier": 0}
    for _, op, args in insns:
        if op.startswith("v_mfma"):
            s["mfma"] += 1
        elif op.startswith("v_"):
            s["valu"] += 1
        elif op.startswith("ds_read"):
            s["ds_read"] += 1
        elif op.startswith("ds_"):
            s["ds_write"] += 1
        elif op.startswith("global_load_lds"):
            s["dma"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            s["vmem"] += 1
        elif op.startswith("scratch_"):
            s["scratch"] += 1
        elif op == "s_waitcnt":
            s["waitcnt"] += 1
        elif op == "s_barrier":
            s["barrier"] += 1
        elif op.startswith("s_"):
            s["salu"] += 1
    # instructions behind the last MFMA = the epilogue
    last = max((i for i, (_, op, _) in enumerate(insns) if op.startswith("v_mfma")), default=-1)
    if last >= 0:
        tail = insns[last + 1:]
        s["epilogue_valu"] = sum(1 for _, op, _ in tail if op.startswith("v_") and not op.startswith("v_mfma"))
        s["epilogue_vmem_store"] = sum(1 for _, op, _ in tail if op.startswith(("global_store", "buffer_store")))
        s["epilogue_vmem_load"] = sum(1 for _, op, _ in tail if op.startswith(("global_load", "buffer_load")))
    return s


def run(so: Path, kernel_filter: str | None = None, want_stats: bool = False, dump: Path | None = None) -> tuple[list[str], dict]:
    violations: list[str] = []
    summary = {"code_objects": 0, "kernels": 0, "dma_kernels": 0, "asm_read_kernels": 0, "stats": {}}
    for k, elf in enumerate(code_objects(so)):
        summary["code_objects"] += 1
        funcs, meta = disassemble(elf)
        if dump:
            dump.mkdir(parents=True, exist_ok=True)
            (dump / f"co{k}.elf").write_bytes(elf)
        mangled_private = {}
        for mname, md in meta.items():
            mangled_private[mname] = md
        for name, insns in funcs.items():
            if kernel_filter and kernel_filter not in name:
                continue
            if not insns:
                continue
            summary["kernels"] += 1
            has_dma = any(op.startswith("global_load_lds") for _, op, _ in insns)
            has_asm = any(op == "ds_read_b32" and "offset:" in a for _, op, a in insns) and has_dma
            summary["dma_kernels"] += has_dma
            summary["asm_read_kernels"] += has_asm
            violations += lint_function(name, insns)
            n_scr = sum(1 for _, op, _ in insns if op.startswith("scratch_"))
            if n_scr:
                violations.append(f"{name}: {n_scr} scratch_ instructions (R3)")
            if want_stats:
                summary["stats"][name] = stats(insns)
        # R3 (metadata): private segment / VGPR spills of EVERY kernel (SGPR spills go to VGPR lanes, not to memory)
        names = list(meta)
        if names:
            for mname, dname in zip(names, demangle(names)):
                if kernel_filter and kernel_filter not in dname:
                    continue
                md = meta[mname]
                summary["metadata_checked"] = summary.get("metadata_checked", 0) + 1
                if md["private"] or md["vgpr_spill"]:
                    violations.append(f"{dname}: private segment {md['private']} B, spills v{md['vgpr_spill']} s{md['sgpr_spill']} (R3)")
    return violations, summary


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("so", nargs="?", default=str(ROOT / "multimodal_vqvae_compression_audio_tactile_amd" / "libmvq_hip.so"))
    ap.add_argument("--stats", action="store_true", help="print per-kernel instruction counts (JSON lines)")
    ap.add_argument("--kernel", default=None, help="only kernels whose demangled name contains this")
    ap.add_argument("--dump", default=None, help="write the extracted code objects here")
    a = ap.parse_args()
    bad, summary = run(Path(a.so), a.kernel, a.stats, Path(a.dump) if a.dump else None)
    if a.stats:
        import json
        for name, s in sorted(summary["stats"].items()):
            print(json.dumps({"kernel": name.split("(")[0], **s}))
    for b in bad:
        print("VIOLATION:", b)
    print(f"isa_lint: {summary['code_objects']} code objects, {summary['kernels']} kernels ({summary['dma_kernels']} LDS-DMA, "
          f"{summary['asm_read_kernels']} with hand-placed operand reads), {len(bad)} violations", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
