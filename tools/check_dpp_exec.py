#!/usr/bin/env python3
"""Build-time check for a hipcc behaviour that cost a silent wrong result in round 5 (docs/DESIGN_measurement_log_r1_r5.md 13): a DPP move written under a
select (`lane ? dpp(x) : y`) was SUNK into the divergent branch, where lanes read from EXEC-disabled neighbours and get nothing.
Cross-lane DPP operations (row_shr / row_bcast / row_newbcast / quad_perm / wave_shr ...) in the product kernels are all meant to run
with the full wavefront active; this script disassembles the kernels that use them and proves, by a forward data-flow over the
control-flow graph of each kernel, that no DPP instruction can execute while a divergent region is open:
    s_and_saveexec_b64 sX, m                                                        opens a region saved in sX
    s_xor_b64 sY, exec, sX ; s_andn2_saveexec / s_or_saveexec sZ, sY                the else-flip: the region moves to sY, then sZ
    s_or_b64 exec, exec, sX                                                         closes it
    s_andn2_b64 exec, exec, .. / s_and_b64 exec, exec, .. / v_cmpx_*                narrow EXEC until the next full restore
States merge by union (conservative).  usage: check_dpp_exec.py [file.hip ...]    (exit status 0 = ok)"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "neural-tape-modeling_amd", "csrc")
DEFAULT = [os.path.join(CSRC, f) for f in ("gru_mfma2.hip", "gru_lat.hip", "gru_small.hip")]
DPP = re.compile(r"\b(row_shr|row_shl|row_ror|row_bcast|row_newbcast|row_mirror|row_half_mirror|quad_perm|wave_shr|wave_shl|wave_ror|wave_rol)\b")
SAVE = re.compile(r"^s_and_saveexec_b64\s+(s\[\d+:\d+\])")
FLIP = re.compile(r"^s_(?:andn2|or|xor)_saveexec_b64\s+(s\[\d+:\d+\]),\s*(s\[\d+:\d+\])")      # the else-flip of an if / else
MOVE = re.compile(r"^s_xor_b64\s+(s\[\d+:\d+\]),\s*exec,\s*(s\[\d+:\d+\])")              # else lanes = saved ^ exec
CLOSE = re.compile(r"^s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")
NARROW = re.compile(r"^(?:s_(?:and|andn2)_b64\s+exec,\s*exec|v_cmpx_)")
RESTORE = re.compile(r"^s_mov_b64\s+exec,")
DEST = re.compile(r"^[sv]_\w+\s+(s\[\d+:\d+\])\s*,")
BRANCH = re.compile(r"^s_(c?branch\w*)\s+(\.LBB\w+)")


def kernels(src):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(HERE, "..", "include"), "-Wno-unused-function",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    return re.findall(r"^(\w+):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel", text, flags=re.S | re.M)


def check_kernel(name, body):
    """-> list of offending DPP instructions (with the regions open at them)."""
    ins, labels = [], {}
    for ln in body.splitlines():
        m = re.match(r"\s*(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = ln.split(";")[0].strip()
        if t and not t.startswith("."):
            ins.append(t)
    n = len(ins)
    state = [None] * (n + 1)                 # set of open regions on entry to instruction i (None = not reached yet)
    state[0] = frozenset()
    work = [0]
    while work:
        i = work.pop()
        if i >= n:
            continue
        st = set(state[i])
        t = ins[i]
        m = SAVE.match(t)
        if m:
            st.add(m.group(1))
        m = MOVE.match(t)
        if m and m.group(2) in st:           # the lanes to come back to now live in another register pair
            st.discard(m.group(2))
            st.add(m.group(1))
        m = FLIP.match(t)
        if m:                                # then-lanes saved in the destination, EXEC = the else lanes: still one open region
            st.discard(m.group(2))
            st.add(m.group(1))
        m = CLOSE.match(t)
        if m:
            st.discard(m.group(1))
            st.discard("narrowed")           # a full `exec |= saved` also ends a lane-retirement narrowing
        else:
            # a register pair that is overwritten by anything else no longer holds a mask to come back to (the compiler
            # re-uses these pairs as ordinary booleans once a region is closed on every path)
            d = DEST.match(t)
            if d and not SAVE.match(t) and not FLIP.match(t) and not MOVE.match(t):
                st.discard(d.group(1))
        if NARROW.match(t):
            st.add("narrowed")
        if RESTORE.match(t):
            st = set()
        succ = []
        b = BRANCH.match(t)
        if b:
            succ.append(labels[b.group(2)])
            if b.group(1) != "branch":
                succ.append(i + 1)
        elif t != "s_endpgm":
            succ.append(i + 1)
        for j in succ:
            new = frozenset(st) if state[j] is None else frozenset(st) | state[j]
            if state[j] is None or new != state[j]:
                state[j] = new
                work.append(j)
    return [(i, ins[i], sorted(state[i])) for i in range(n) if DPP.search(ins[i]) and state[i]]


def main(argv):
    files = argv or DEFAULT
    total, bad = 0, []
    for f in files:
        for name, body in kernels(f):
            nd = sum(1 for ln in body.splitlines() if DPP.search(ln.split(";")[0]))
            if not nd:
                continue
            total += nd
            for i, t, st in check_kernel(name, body):
                bad.append(f"{os.path.basename(f)} {name}: `{t}` may run with a divergent region open {st}")
    for b in bad:
        print("FAIL:", b, file=sys.stderr)
    if bad:
        return 1
    print(f"check_dpp_exec: {total} DPP instructions in {len(files)} files, none under a partial EXEC mask: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
