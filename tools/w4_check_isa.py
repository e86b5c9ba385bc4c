"""gemm_w4.hip keeps its accumulators in AGPRs behind the compiler's back (inline asm only).  This check compiles the
file to assembly and asserts that, inside the w4 kernels, no instruction OUTSIDE an inline-asm block names an AGPR and
that there is no scratch.  Usage: python tools/w4_check_isa.py  (needs hipcc; no GPU)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "afft_amd", "csrc", "gemm_w4.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "w4.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           src, "-o", out] + sys.argv[1:])
    text = open(out).read()
bad = 0
for m in re.finditer(r"^(_ZN\S*gemm_bf16_w4r?_kernel\S*):[^\n]*\n(.*?)\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    in_asm = False
    n_out = 0
    for line in body.splitlines():
        if "#ASMSTART" in line: in_asm = True; continue
        if "#ASMEND" in line: in_asm = False; continue
        code = line.split(";")[0]
        if not in_asm and re.search(r"\ba(\d+|\[)", code):
            n_out += 1
            if n_out <= 5: print("AGPR outside asm:", line.strip())
    stats = {k: int(v) for k, v in re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", text[m.end():m.end() + 8000])}
    print(name[:90], stats, "agpr-outside-asm:", n_out)
    bad += n_out + stats.get("ScratchSize", 0)
sys.exit(1 if bad else 0)
