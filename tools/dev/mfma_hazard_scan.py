"""Scan a gfx950 code object's disassembly for VALU-write -> MFMA-read hazards that an inline-asm MFMA does not get padded
for: any v_mfma whose A/B source VGPRs were written by a non-memory VALU instruction in the previous 2 issue slots
(s_nop N counts N+1).  usage: python tools/dev/mfma_hazard_scan.py <file.o> <kernel-name-substring>"""
import re, subprocess, sys, tempfile, os
lib, pat = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
# <lib> is the OBJECT file of one translation unit (csrc/build/mlp.o or mlp_<variant>.o): its .hip_fatbin holds one bundle
co = f"{tmp}/k.co"
subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{tmp}/fat.bin"], check=True)
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={tmp}/fat.bin", f"--output={co}"], check=True)
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
inside, hist, bad, nm = False, [], 0, 0
for line in dis.splitlines():
    if re.match(r"^[0-9a-f]+ <", line):
        inside = pat in line; hist = []; continue
    if not inside: continue
    m = re.match(r"\s+(\S+)\s*(.*?)\s*//", line)
    if not m: continue
    op, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if op.startswith("v_mfma") and args[0].startswith("a["):  # the inline-asm MFMAs: accumulators pinned in AGPRs
        nm += 1
        src = regs(args[1]) | regs(args[2])
        slots = 0
        for pop, pdst in reversed(hist):
            if slots >= 2: break
            if pop.startswith("v_") and not pop.startswith("v_mfma") and (pdst & src):
                bad += 1; print("HAZARD:", pop, sorted(pdst & src), "->", line.strip()[:90]); break
            slots += (int(pdst) + 1) if pop == "s_nop" else 1
    if op == "s_nop": hist.append((op, int(args[0])))
    else: hist.append((op, regs(args[0]) if args else set()))
    hist = hist[-6:]
print(f"{nm} MFMAs scanned, {bad} suspect")
