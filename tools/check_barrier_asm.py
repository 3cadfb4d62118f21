#!/usr/bin/env python3
"""Build-time check of the one hand-counted wait in the product: the step barrier of gru_mfma2_kernel is the inline asm
`s_waitcnt lgkmcnt(1); s_barrier`.  It is correct only if, when a wave reaches it, exactly ONE LDS/SMEM operation
younger than its h-exchange write can still be outstanding (the y-partial `ds_write_b32`): LDS ops retire in order, so
lgkmcnt <= 1 then proves that the h write has completed.  An extra ds op, an LDS spill or a re-materialised s_load
between the two would turn the exchange into a silent race that only the parity tests could catch -- so the
disassembly of every non-diagnostic instantiation is checked here: walking back from each such barrier to the nearest
h-exchange write (ds_write_b128, or the two ds_write_b64 of the f16x3 engine) there must be exactly one lgkm-counted
instruction and it must be a ds_write_b32.

usage: check_barrier_asm.py [gru_mfma2.hip]     (exit status 0 = ok)"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "neural-tape-modeling_amd", "csrc", "gru_mfma2.hip")
LGKM = re.compile(r"^\s*(ds_|s_load_|s_buffer_load|s_memtime|s_memrealtime|s_sendmsg|s_scratch_load)")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(HERE, "..", "include"),
                        "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", asm, SRC],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    kernels = re.findall(r"^(_ZN3ntm16gru_mfma2_kernel\w+):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel", text, flags=re.S | re.M)
    checked = 0
    for name, body in kernels:
        m = re.match(r"_ZN3ntm16gru_mfma2_kernelILb1ELb(\d)ELi(\d+)ELi(\d)ELi(\d+)E(?:Lb\dE)?EE", name)
        if not m or m.group(1) == "1" or m.group(2) != "0":
            continue                                      # STAMP / ablation builds are diagnostics (they use lgkmcnt(0))
        # the step loop: from its "Inner Loop Header" label to the last branch back to it, taken as a CYCLIC sequence
        # (the h write that a barrier waits for is at the end of the previous trip through the body)
        raw = body.splitlines()
        lines = None
        for h in [i for i, ln in enumerate(raw) if "Inner Loop Header" in ln]:     # the step loop is the one with barriers
            tag = "Header=" + raw[h].split(":")[0].strip().lstrip(".L")            # blocks of the loop carry this in their comment
            cand, inside = [], True
            for ln in raw[h + 1:]:
                if re.match(r"\s*(\.LBB\w+:|; %bb\.\d+:)", ln):                       # a basic-block boundary
                    inside = tag in ln
                    continue
                if inside and ln.split(";")[0].strip():
                    cand.append(ln.split(";")[0].strip())
            if "s_barrier" in cand:
                assert lines is None, f"{name}: two loops with barriers"
                lines = cand
        assert lines, f"{name}: step loop not found"
        n = len(lines)
        step_barriers = [i for i, ln in enumerate(lines) if ln == "s_barrier" and "lgkmcnt(1)" in lines[i - 1]]
        assert len(step_barriers) == 2, f"{name}: expected the two step barriers of the 2x unrolled loop, found {len(step_barriers)}"
        for b in step_barriers:
            between, k = [], 2
            while k <= n:
                ln = lines[(b - k) % n]
                if re.match(r"ds_write(2)?_b(128|64)\b", ln):
                    break
                if LGKM.match(ln):
                    between.append(ln)
                k += 1
            assert k <= n, f"{name}: no h-exchange write ahead of a step barrier"
            ok = len(between) == 1 and between[0].startswith("ds_write_b32")
            assert ok, f"{name}: between the h write and the step barrier: {between} (expected exactly one ds_write_b32)"
            checked += 1
    assert checked >= 4, f"only {checked} step barriers checked"
    print(f"check_barrier_asm: {checked} step barriers in {len(kernels)} instantiations: ok")


if __name__ == "__main__":
    main()
