#!/usr/bin/env python3
"""Register / LDS / scratch numbers of every kernel in a built library, read from the gfx950 code objects it embeds
(llvm-objdump --offloading unbundles them, llvm-readelf --notes prints the AMDGPU metadata), plus the compiler that
produced them.  __graft_entry__.build() writes the result to neural-tape-modeling_amd/build_info.json; bench.py attaches
it to its line as `build`, so a compiler change or a box-to-box difference shows in the record.

    python3 tools/kernel_resources.py [libntm.so] [-o build_info.json] [--check]

--check (also run by tests/test_host.py): no PRODUCT kernel may use scratch or spill a register, and the recurrent
kernels' VGPR counts may not exceed the binary the committed profiles were measured on (PINNED_VGPRS below) -- the kernel
leans on hipcc-specific workarounds (docs/DESIGN_measurement_log_r1_r5.md 4), so a silent register-allocation change is a performance event."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LLVM = os.environ.get("NTM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = os.path.join(ROOT, "neural-tape-modeling_amd", "libntm.so")

# gru_mfma2_kernel<PRESCALE, STAMP, ABL, ENGINE, YPN, FUSE, ESR[, DCPRE]> of libntm.so: VGPRs of the binary behind profiles/r05_*
# (hipcc 7.2); keyed by (ENGINE, YPN, FUSE, ESR, DCPRE).  A count ABOVE the pin fails --check; below is reported.
PINNED_VGPRS = {
    (0, 16, 0, 0, 0): 194, (0, 4, 0, 0, 0): 184,
    (0, 16, 0, 1, 0): 254, (0, 4, 0, 1, 0): 206,
    (1, 16, 0, 0, 0): 194, (1, 4, 0, 0, 0): 190,
    (0, 16, 1, 0, 0): 222, (0, 4, 1, 0, 0): 216,
    (0, 16, 1, 1, 0): 234, (0, 4, 1, 1, 0): 228,
    (0, 16, 0, 1, 1): 264, (0, 4, 0, 1, 1): 220,          # ESR + DCP: 256 VGPRs + 8 AGPRs (AGPR spills, no scratch)
    (0, 16, 1, 1, 1): 251, (0, 4, 1, 1, 1): 243,          # FUSE + ESR + DCP
    (2, 16, 0, 0, 0): 250, (2, 4, 0, 0, 0): 250,          # bf16x3 engine (round 6; opt-in): <= 256, so that two groups share a CU from 8192 streams up
}
MFMA2_DYNAMIC_LDS = {16: 151680, 4: 53376}      # csrc/gru_mfma2.hip m2::smem_floats(YPN) * 4
MFMA2_DYNAMIC_LDS_BF16X3 = {16: 153728, 4: 55424}      # ... m2::smem_floats(YPN, 2) * 4: the exchange buffer holds three pieces


def hipcc_version():
    try:
        out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.TimeoutExpired):
        return None
    hip = re.search(r"HIP version:\s*(\S+)", out)
    clang = re.search(r"clang version\s*(\S+)", out)
    return {"hip": hip.group(1) if hip else None, "clang": clang.group(1) if clang else None}


def demangle(names):
    try:
        out = subprocess.run([shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")] + names, capture_output=True, text=True,
                             timeout=60).stdout
        d = out.splitlines()
        return d if len(d) == len(names) else names
    except (OSError, subprocess.TimeoutExpired):
        return names


def kernels_of(lib_path):
    """-> list of dicts, one per kernel of every gfx950 code object bundled in `lib_path`."""
    out = []
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        local = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, local)                      # the unbundler writes next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp, timeout=300)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True,
                                   text=True, timeout=300).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda key: re.search(r"\." + key + r":\s*(\S+)", blk)       # noqa: E731
                if not g("name"):
                    continue
                out.append({"symbol": g("name").group(1), "vgprs": int(g("vgpr_count").group(1)), "agprs": int(g("agpr_count").group(1)),
                            "sgprs": int(g("sgpr_count").group(1)), "lds_static_bytes": int(g("group_segment_fixed_size").group(1)),
                            "scratch_bytes": int(g("private_segment_fixed_size").group(1)),
                            "vgpr_spills": int(g("vgpr_spill_count").group(1)) if g("vgpr_spill_count") else 0,
                            "sgpr_spills": int(g("sgpr_spill_count").group(1)) if g("sgpr_spill_count") else 0})
    for k, d in zip(out, demangle([k["symbol"] for k in out])):
        k["kernel"] = re.sub(r"^void\s+", "", d).replace("(ntm::GruArgs)", "")
    return out


def mfma2_key(symbol):
    m = re.match(r"_ZN3ntm16gru_mfma2_kernelILb1ELb(\d)ELi(\d+)ELi(\d)ELi(\d+)E((?:Lb\dE)*)EE", symbol)
    if not m or m.group(1) == "1" or m.group(2) != "0":
        return None                                       # not the product kernel, or a diagnostic instantiation
    fl = [int(v) for v in re.findall(r"Lb(\d)E", m.group(5))] + [0, 0, 0]
    return (int(m.group(3)), int(m.group(4)), fl[0], fl[1], fl[2])


def build_info(lib_path=DEFAULT_LIB):
    ks = kernels_of(lib_path)
    for k in ks:
        key = mfma2_key(k["symbol"])
        if key:
            k["lds_dynamic_bytes"] = (MFMA2_DYNAMIC_LDS_BF16X3 if key[0] == 2 else MFMA2_DYNAMIC_LDS).get(key[1])
            k["pinned_vgprs"] = PINNED_VGPRS.get(key)
    try:
        src = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=30).stdout.strip() or None
    except (OSError, subprocess.TimeoutExpired):
        src = None
    import hashlib
    with open(lib_path, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    return {"library": os.path.relpath(lib_path, ROOT), "library_sha256": digest, "compiler": hipcc_version(), "arch": "gfx950", "git_head_at_build": src,
            "kernels": sorted(ks, key=lambda k: k["kernel"])}


def check(info):
    """Product rules: no scratch, no VGPR spills anywhere in the library; pinned VGPR ceilings for the recurrent kernel."""
    bad = []
    for k in info["kernels"]:
        if k["scratch_bytes"] or k["vgpr_spills"]:        # (SGPR spills go to VGPR lanes, not to memory: reported, not failed)
            bad.append(f"{k['kernel']}: scratch {k['scratch_bytes']} B, {k['vgpr_spills']} VGPR spills")
        key = mfma2_key(k["symbol"])
        if key is None:
            continue
        pin = PINNED_VGPRS.get(key)
        if pin is None:
            bad.append(f"{k['kernel']}: product instantiation {key} has no pinned VGPR count (tools/kernel_resources.py)")
        elif k["vgprs"] > pin:
            bad.append(f"{k['kernel']}: {k['vgprs']} VGPRs, the measured binary has {pin}: register allocation changed -- re-measure")
        elif k["vgprs"] < pin:
            print(f"note: {k['kernel']}: {k['vgprs']} VGPRs (pinned {pin})")
    return bad


def main(argv):
    lib = next((a for a in argv if a.endswith(".so")), DEFAULT_LIB)
    info = build_info(lib)
    if "-o" in argv:
        with open(argv[argv.index("-o") + 1], "w") as f:
            json.dump(info, f, indent=1)
    else:
        print(json.dumps(info, indent=1))
    if "--check" in argv:
        bad = check(info)
        for b in bad:
            print("FAIL:", b, file=sys.stderr)
        if bad:
            return 1
        print(f"kernel_resources: {len(info['kernels'])} kernels, no scratch, no spills, VGPR pins hold: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
