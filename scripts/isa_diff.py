#!/usr/bin/env python
"""Which kernels of a build have EXACTLY the machine code of another build?  Disassembles the gfx950 code objects of two
libraries on the CPU (llvm-objdump of /opt/rocm) and compares every kernel's instruction stream (addresses and encodings
dropped, branch targets kept as offsets inside the function).  Used in round 3 -- no GPU access -- to show which kernels are
still the code that was green on hardware at commit 8cbc337 (the last GPU run of round 2) and which are new / changed and
therefore wait for their first GPU run.

    python scripts/isa_diff.py OLD.so NEW.so
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    d = tempfile.mkdtemp(prefix="isa")
    try:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
        out = {}
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
            names = set(re.findall(r"\.name:\s+(\S+)", notes))
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(d, f)], capture_output=True, text=True, check=True).stdout
            cur, base = None, 0
            for ln in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:", ln)
                if m:
                    cur = m.group(2) if m.group(2) in names else None
                    base = int(m.group(1), 16)
                    if cur:
                        out[cur] = []
                    continue
                if cur and "\t" in ln:
                    ins = ln.split("//")[0].strip()
                    # branch targets: absolute address -> offset inside the function
                    ins = re.sub(r"<[^>]+>", "", ins)
                    out[cur].append(ins)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)


def main():
    old, new = kernels(sys.argv[1]), kernels(sys.argv[2])
    dm = demangle(sorted(set(old) | set(new)))
    same = sorted(short(dm[k]) for k in new if k in old and old[k] == new[k])
    changed = sorted((short(dm[k]), len(old[k]), len(new[k])) for k in new if k in old and old[k] != new[k])
    gone_keys = [k for k in old if k not in new]
    renamed, added = [], []
    for k in new:
        if k in old:
            continue
        twin = [g for g in gone_keys if old[g] == new[k]]
        if twin:
            renamed.append((short(dm[k]), short(dm[twin[0]])))
        else:
            added.append((short(dm[k]), len(new[k])))
    renamed.sort(); added.sort()
    gone = sorted(short(dm[k]) for k in gone_keys)
    print("kernels: %d old, %d new" % (len(old), len(new)))
    print("\nIDENTICAL machine code (%d):" % len(same))
    for n in same:
        print("   ", n)
    print("\nIDENTICAL machine code under a new name (%d): new <- old" % len(renamed))
    for n, o in renamed:
        print("    %-60s <- %s" % (n, o))
    print("\nCHANGED (%d): name, instructions old -> new" % len(changed))
    for n, a, b in changed:
        print("    %-90s %6d -> %6d" % (n, a, b))
    print("\nNEW (%d): name, instructions" % len(added))
    for n, b in added:
        print("    %-90s %6d" % (n, b))
    print("\nGONE (%d):" % len(gone))
    for n in gone:
        print("   ", n)


if __name__ == "__main__":
    main()
