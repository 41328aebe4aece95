#!/usr/bin/env python
"""One table for "what has run on hardware": every kernel of the library, the hash of its machine code, and the newest build
that ran on an MI355X holding bit-identical code for it.

    python scripts/kernel_state_table.py [LIB.so] > profiles/r04_kernel_state.md

Reference builds (ab/, built from git worktrees of those commits with the same toolchain -- scripts/build_variant.sh does
the same for HEAD):
    ab/r1_0b7050d.so   round 1, the last commit a DRIVER run saw green (GPUTEST_r01.json: 90 passed)
    ab/r2_8cbc337.so   round 2, the last commit the BUILDER ran on MI355X boxes (113 GPU tests, profiles/r02_a_*)
Rounds 3 and 4 had no GPU access: a kernel that matches neither build has only run on the CPU emulation of its source
(tests/hip_emu).  The comparison is scripts/isa_diff.py's: instruction streams with addresses dropped."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_hashes as kh  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFS = (("driver, round 1 (0b7050d)", os.path.join(REPO, "ab", "r1_0b7050d.so")),
        ("builder, round 2 (8cbc337)", os.path.join(REPO, "ab", "r2_8cbc337.so")))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "bot-sort-onnx-tensorrt_amd", "libbotsort_hip.so")
    head = kh.library_hashes(lib)
    refs = [(name, set(kh.library_hashes(p).values())) for name, p in REFS if os.path.exists(p)]
    rows = []
    n_ref = [0] * len(refs)
    never = 0
    for k in sorted(head):
        marks = []
        for j, (name, hs) in enumerate(refs):
            hit = head[k] in hs
            marks.append("identical" if hit else "-")
            n_ref[j] += int(hit)
        never += int(all(m == "-" for m in marks))
        rows.append((k, head[k], marks))
    print("| kernel | machine-code hash | " + " | ".join(name for name, _ in refs) + " |")
    print("|---|---|" + "---|" * len(refs))
    for k, h, marks in rows:
        print("| `%s` | `%s` | %s |" % (k, h, " | ".join(marks)))
    print()
    print("%d kernels; bit-identical machine code in: %s; in neither (CPU emulation of the source only): %d"
          % (len(rows), ", ".join("%s %d" % (name, n) for (name, _), n in zip(refs, n_ref)), never))


if __name__ == "__main__":
    main()
