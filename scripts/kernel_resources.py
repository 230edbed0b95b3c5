#!/usr/bin/env python3
"""Compiles every kernel source of multi_modal_amd/csrc with the in-tree flags plus -Rpass-analysis=kernel-resource-usage and prints
one line per kernel instantiation: registers, spills, scratch, the compiler's occupancy and the static LDS size.

    python scripts/kernel_resources.py > profiles/<tag>_kernel_resources.txt

The objects go to build/res/ (not part of the product).  The flags are read from the Makefile's rules: fbank512.hip and resample.hip
are built without the IR load/store vectoriser, like the in-tree build.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("fbank512", "rfft400", "stft32", "dftgen", "resample", "frame_stats", "patch_embed")
NO_VECTORISER = ("fbank512", "resample")
BASE = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-signed-zeros -fno-slp-vectorize -Wall -Wno-unused-function".split()
FIELDS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch_B"),
          ("Occupancy [waves/SIMD]", "occ"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("LDS Size [bytes/block]", "lds_static_B"))


def remarks(src):
    out = os.path.join(ROOT, "build", "res")
    os.makedirs(out, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc"] + BASE + (["-mllvm", "-amdgpu-load-store-vectorizer=0"] if src in NO_VECTORISER else [])
    cmd += ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "multi_modal_amd", "csrc"),
            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "multi_modal_amd", "csrc", src + ".hip"),
            "-o", os.path.join(out, src + ".o")]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return r.stderr


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.split("\n")


def main():
    print("# compiler resource usage of every kernel instantiation in libmmfe.so (hipcc " + " ".join(BASE) + ")")
    print("# occ = the compiler's waves/SIMD from registers alone; the LDS a launch asks for dynamically can lower it")
    print("# %-6s %5s %5s %5s %10s %10s %10s %4s %12s  kernel" % ("source", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_B", "occ", "lds_static_B"))
    for src in SOURCES:
        text = remarks(src)
        rows, cur = [], None
        for line in text.split("\n"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            for label, key in FIELDS:
                m = re.search(r"remark:\s+" + re.escape(label) + r": (\d+)", line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"mmfe::\(anonymous namespace\)::", "", n)
            n = re.sub(r"^void ", "", n)
            n = re.sub(r"\(.*\)$", "", n)
            print("%-8s %5d %5d %5d %10d %10d %10d %4d %12d  %s" % (src, r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"],
                                                                  r["scratch_B"], r["occ"], r["lds_static_B"], n))


if __name__ == "__main__":
    main()
