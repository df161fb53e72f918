"""Scan gfx950 code for VALU-write -> MFMA-operand-read pairs closer than the 2 wait states the hardware needs
(cdna_hip_programming.md 5.7 item 2): ANY v_mfma - builtin or inline asm - whose A / B / C source VGPRs were written by a
non-MFMA VALU instruction in the previous two issue slots (s_nop N counts N + 1).  hipcc pads such a pair itself when it
knows the writer is a VALU instruction; it does NOT when the writer sits inside an `asm` statement - and it is free to
schedule a non-volatile asm statement right in front of the MFMA that consumes its output.  Round 4 found exactly that in
the round-3 two-tile experiment's prologue forward (tools/dev/k4_pipe_experiment.patch): `v_pk_max_i16 v38` one slot ahead
of the v_mfma reading v[36:39] - the "first tile of every wave is wrong" failure - and 23 more such triples in the shipped
K3 / K4, masked only by LDS waits (csrc/mlp.hip, "Packed 16-bit helpers").

usage: python tools/dev/mfma_operand_hazard_scan.py <file.s | lib.so | file.o> [kernel-name-substring]
       (tests/test_lib_abi.py calls scan_library() on the built libhbr_hip.so)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_lines(lines, pat=""):
    """lines: hipcc -S output or llvm-objdump -d output.  Returns (mfmas scanned, [findings])."""
    kernel, inside, in_asm = "", True, False
    hist, total, found = [], 0, []
    for ln, line in enumerate(lines, 1):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line) or re.match(r"^([A-Za-z_$][\w.$]*):", line)
        if m:  # a function symbol (objdump / .s)
            kernel, inside, hist = m.group(1), pat in m.group(1), []
            continue
        if re.match(r"^\.L\w+:", line):
            hist = []  # a branch target in a .s file: the predecessor's tail is unknown - start over
            continue
        if not inside:
            continue
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s[0] in ";./":
            continue
        s = re.split(r"\s//|;", s)[0].strip()  # drop objdump's encoding comment / .s comments
        parts = s.split(None, 1)
        if not parts:
            continue
        op, args = parts[0], ([a.strip() for a in parts[1].split(",")] if len(parts) > 1 else [])
        if op.startswith("v_mfma"):
            total += 1
            src = set()
            for a in args[1:4]:
                src |= _regs(a)
            slots = 0
            for pop, pdst, pasm, pln in reversed(hist):
                if slots >= 2:
                    break
                if pop.startswith("v_") and not pop.startswith(("v_mfma", "v_accvgpr_read", "v_nop")) and isinstance(pdst, set) and (pdst & src):
                    found.append(f"{kernel[:80]} line {ln}: {pop} (line {pln}{', inline asm' if pasm else ''}) writes v{sorted(pdst & src)} "
                                 f"{slots} slot(s) ahead of {s[:90]}")
                    break
                slots += (pdst + 1) if pop == "s_nop" else 1
        if op == "s_nop":
            hist.append((op, int(args[0], 0), in_asm, ln))
        else:
            hist.append((op, _regs(args[0]) if args else set(), in_asm, ln))
        hist = hist[-8:]
    return total, found


def disassemble(path):
    """Every gfx950 code object bundled in an ELF's .hip_fatbin section (a .o holds one bundle, the .so one per
    translation unit), disassembled."""
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
    out = []
    for i in range(len(offs) - 1):
        part, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"b{i}.co")
        open(part, "wb").write(data[offs[i]:offs[i + 1]])
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={part}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        out += subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout.splitlines()
    return out


def scan_library(path, pat=""):
    return scan_lines(disassemble(path), pat)


if __name__ == "__main__":
    path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    total, found = scan_lines(open(path), pat) if path.endswith(".s") else scan_library(path, pat)
    for f in found:
        print("HAZARD", f)
    print(f"{total} MFMAs scanned, {len(found)} suspect pairs")
