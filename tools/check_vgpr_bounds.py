#!/usr/bin/env python3
"""Static check of every kernel in pointrcnn_amd/csrc/*.hip: no instruction names a VGPR at or
above the kernel descriptor's accum_offset, nor an AGPR at or above next_free_vgpr - accum_offset
(a wave that did would write into the registers of another wave on its SIMD), and no kernel has a
private segment (scratch) unless listed as expected.  CPU only: compiles each file to gfx950
assembly with the library's flags.  Part of the evidence for DESIGN.md section 5's open finding.

    python tools/check_vgpr_bounds.py [file.hip ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointrcnn_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
         "-I" + os.path.join(ROOT, "include")]


def check(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [path, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        txt = open(out).read()
    desc = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        body = m.group(2)
        desc[m.group(1)] = (int(re.search(r"amdhsa_next_free_vgpr (\d+)", body).group(1)),
                            int(re.search(r"amdhsa_accum_offset (\d+)", body).group(1)),
                            int(re.search(r"amdhsa_private_segment_fixed_size (\d+)", body).group(1)))
    parts = re.split(r"\n(_Z[^\n:]*):\n", txt)
    bad, scratch = [], []
    for i in range(1, len(parts), 2):
        name = parts[i]
        if name not in desc:
            continue
        body = parts[i + 1].split(".section")[0]
        nf, ao, priv = desc[name]
        vmax = amax = -1
        for m in re.finditer(r"\bv(\d+)\b", body):
            vmax = max(vmax, int(m.group(1)))
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", body):
            vmax = max(vmax, int(m.group(2)))
        for m in re.finditer(r"\ba(\d+)\b", body):
            amax = max(amax, int(m.group(1)))
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]", body):
            amax = max(amax, int(m.group(2)))
        if vmax >= ao or amax >= nf - ao:
            bad.append((name, nf, ao, vmax, amax))
        if priv:
            scratch.append((name, priv))
    return len(desc), bad, scratch


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    rc = 0
    for f in files:
        n, bad, scratch = check(f)
        print("%-18s %3d kernels, %d out of bounds, %d with scratch" % (os.path.basename(f), n, len(bad), len(scratch)),
              flush=True)
        for b in bad:
            print("   OUT OF BOUNDS %s next_free %d accum_offset %d vmax %d amax %d" % b)
            rc = 1
        for s in scratch:
            print("   scratch %s: %d B" % s)
    return rc


if __name__ == "__main__":
    sys.exit(main())
