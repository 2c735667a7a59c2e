"""Static audit of the compiled prefill_w64_kernel (no GPU needed: hipcc cross-compiles gfx950).

The kernel keeps its output accumulator and Q fragments in accumulator registers a[64:255] that
only inline-asm statements name; nothing reserves them, so hipcc may home its own values in
accumulator registers too. It is safe only while every accumulator register hipcc itself uses
(anything outside ASMSTART/ASMEND) lies in a[0:63], and while nothing goes to scratch memory.
"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "light_vllm_amd", "csrc")
OWNED_FROM = 64


def _kernel_bodies(asm: str):
    for m in re.finditer(r"^(_ZN2lv18prefill_w64_kernelILb[01]E\w*):[^\n]*\n(.*?)\n\ts_endpgm", asm, re.S | re.M):
        yield m.group(1), m.group(2)


def test_w64_kernel_leaves_the_asm_owned_registers_alone():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "prefill.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                        f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
                        os.path.join(CSRC, "prefill_attn.hip"), "-o", out],
                       check=True, capture_output=True, timeout=600)
        asm = open(out).read()
    bodies = list(_kernel_bodies(asm))
    assert len(bodies) == 2, "expected the packed and the paged instantiation"
    for name, body in bodies:
        outside, inside_asm = [], False
        for line in body.splitlines():
            if "ASMSTART" in line:
                inside_asm = True
            elif "ASMEND" in line:
                inside_asm = False
            elif not inside_asm and not line.lstrip().startswith(";"):
                outside.append(line)
        text = "\n".join(outside)
        used = [int(m.group(2) or m.group(1)) for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text)]
        used += [int(m.group(1)) for m in re.finditer(r"\ba(\d+)\b", text)]
        assert max(used, default=-1) < OWNED_FROM, \
            f"{name}: hipcc uses accumulator register a{max(used)} (asm-owned range starts at a{OWNED_FROM})"
        assert "scratch_" not in text, f"{name}: spills to scratch memory"
    meta = re.findall(r"\.name:\s+_ZN2lv18prefill_w64_kernel.*?\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", asm)
    assert all(int(x) == 0 for x in meta)
