#!/usr/bin/env python
"""Compare the kernels of two builds by what decides their occupancy and launch cost: VGPRs (-> wavefronts per SIMD), LDS bytes,
scratch bytes -- read from the code objects' metadata on the CPU (no GPU needed).

    python scripts/kernel_meta_diff.py OLD [OLD ...] -- NEW [NEW ...]        (each: a .so or .o holding gfx950 code objects)

Prints the kernels whose occupancy step, LDS or scratch differ.  This is how round 2's two give-backs were found after GPU
access closed (scratch in the tall cost tiles, 64 KiB of opt-in LDS counters in every tk_front workgroup)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def meta(paths):
    out = {}
    for p in paths:
        d = tempfile.mkdtemp()
        try:
            shutil.copy(p, os.path.join(d, "x.o"))          # (llvm-objdump --offloading writes next to its input)
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.o"], cwd=d, capture_output=True)
            for o in glob.glob(os.path.join(d, "x.o.*gfx950")):
                txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], capture_output=True, text=True).stdout
                for block in txt.split(".agpr_count:")[1:]:
                    f = dict(re.findall(r"\.(name|private_segment_fixed_size|vgpr_count|group_segment_fixed_size):\s+(\S+)", block))
                    if "name" in f:
                        out[f["name"]] = (int(f["vgpr_count"]), int(f["group_segment_fixed_size"]), int(f["private_segment_fixed_size"]))
        finally:
            shutil.rmtree(d)
    return out


def waves(v):
    return min(8, 512 // max(v, 1))


def main():
    args = sys.argv[1:]
    if "--" not in args:
        sys.exit(__doc__)
    k = args.index("--")
    old, new = meta(args[:k]), meta(args[k + 1:])
    print("kernels: %d old, %d new, %d in both" % (len(old), len(new), len(set(old) & set(new))))
    for name in sorted(set(old) & set(new)):
        o, n = old[name], new[name]
        if waves(o[0]) != waves(n[0]) or o[1] != n[1] or o[2] != n[2]:
            nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
            print("%-58s vgpr %3d -> %3d (waves/SIMD %d -> %d)  lds %6d -> %6d  scratch %3d -> %3d"
                  % (nice[:58], o[0], n[0], waves(o[0]), waves(n[0]), o[1], n[1], o[2], n[2]))
    for name in sorted(set(new) - set(old)):
        n = new[name]
        if n[2]:
            print("new kernel with scratch:", name, n)


if __name__ == "__main__":
    main()
