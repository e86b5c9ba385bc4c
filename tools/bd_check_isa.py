"""gemm_bd.hip keeps its accumulators (and, for 160-row tiles, the B operand ring) in AGPRs behind the compiler's back (inline asm
only), and its VGPR B ring (256-row tiles) is the destination of loads the compiler does not know to be in flight.  This check
compiles the file to assembly and asserts, inside every gemm_bf16_bd_kernel instantiation: (1) no instruction OUTSIDE an
inline-asm block names an AGPR, (2) no scratch, (3) between an inline-asm global_load into VGPRs and the next inline-asm
s_waitcnt that names... follows it, no compiler-generated instruction reads or writes the load's destination registers.
Usage: python tools/bd_check_isa.py  (needs hipcc; no GPU)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "afft_amd", "csrc", "gemm_bd.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "bd.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           src, "-o", out] + sys.argv[1:])
    text = open(out).read()


def regs(tok):
    """v[12:15] -> {12..15}; v7 -> {7}"""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = 0
for m in re.finditer(r"^(_ZN\S*gemm_bf16_bd_kernel\S*):[^\n]*\n(.*?)\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    in_asm = False
    n_out = n_touch = 0
    queue = []                # vector-memory operations issued by inline asm, oldest first: destination VGPRs (empty for LDS-DMA)
    for line in body.splitlines():
        if "#ASMSTART" in line: in_asm = True; continue
        if "#ASMEND" in line: in_asm = False; continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":"):
            if code.endswith(":"):
                queue = []            # per basic block: the loop body is one block, which is where it matters
            continue
        if in_asm:
            lm = re.match(r"global_load_dwordx4\s+(\S+?),", code)
            if lm:
                queue.append(regs(lm.group(1)))          # an AGPR destination gives the empty set
            elif code.startswith("global_load_lds"):
                queue.append(set())
            wm = re.search(r"vmcnt\((\d+)\)", code)
            if code.startswith("s_waitcnt") and wm:      # all but the N youngest operations are complete
                n = int(wm.group(1))
                queue = queue[-n:] if n else []
            continue
        if re.search(r"\ba(\d+|\[)", code):
            n_out += 1
            if n_out <= 5: print("AGPR outside asm:", line.strip())
        inflight = set().union(*queue) if queue else set()
        if inflight:
            used = set()
            for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", code):
                used |= regs(tok)
            if used & inflight:
                n_touch += 1
                if n_touch <= 5: print("in-flight VGPR touched:", line.strip())
    stats = {k: int(v) for k, v in re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", text[m.end():m.end() + 8000])}
    print(name[:100], stats, "agpr-outside-asm:", n_out, "inflight-vgpr-touched:", n_touch)
    bad += n_out + n_touch + stats.get("ScratchSize", 0)
sys.exit(1 if bad else 0)
