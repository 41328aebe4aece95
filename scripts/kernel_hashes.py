#!/usr/bin/env python
"""Machine-code identity of kernels, as a table: {kernel name: sha1 of its gfx950 instruction stream}.

Two uses ("next" item 6):
  * `python scripts/kernel_hashes.py LIB.so`                      print the table of a library;
  * `python scripts/kernel_hashes.py --stamp JSON LIB.so COMMIT`  write `kernel_hash` into every entry of a
    profiles/pmc_traffic.json-style file from the library the counters were taken with, plus `taken_at_commit`;
  * `library_hashes(path)` is what bench.py calls on the library it has LOADED: a counter figure is only quoted
    (`roofline.traffic`) when the kernel it was measured on is bit-identical machine code in that library; otherwise
    `traffic` is null and `traffic_source` says why.
The instruction stream is what scripts/isa_diff.py compares (addresses and encodings dropped, branch targets as offsets).
"""
import hashlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_diff  # noqa: E402


def norm(name):
    """rocprofv3 / c++filt spellings of one kernel -> one key: no return type, no anonymous namespace, no argument
    list, no blanks ("void k_filter_i8<false, 0, false>(Dev, int)" -> "k_filter_i8<false,0,false>")."""
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n).strip()
    n = re.sub(r"^void\s+", "", n)
    return n.replace(" ", "")


def base(name):
    return norm(name).split("<")[0]


def library_hashes(lib):
    ks = isa_diff.kernels(lib)
    dm = isa_diff.demangle(sorted(ks))
    return {norm(dm[k]): hashlib.sha1("\n".join(ks[k]).encode()).hexdigest()[:16] for k in ks}


def stamp(json_path, lib, commit):
    tab = json.load(open(json_path))
    hs = library_hashes(lib)
    by_base = {}
    for n, h in hs.items():
        by_base.setdefault(base(n), {})[n] = h
    for wl in tab["workloads"].values():
        for kname, v in wl.items():
            cands = by_base.get(base(kname), {})
            # the profile's spelling may carry fewer template arguments than the library's (defaults): prefix match
            want = norm(kname).rstrip(">")
            hit = [h for n, h in cands.items() if n == norm(kname) or n.startswith(want + ",") or n.startswith(want + ">")]
            v["kernel_hash"] = hit[0] if len(set(hit)) == 1 else None
    tab["taken_at_commit"] = commit
    tab["kernel_hash"] = "sha1[:16] of the kernel's instruction stream in the library the counters were taken with (scripts/kernel_hashes.py)"
    json.dump(tab, open(json_path, "w"), indent=1)
    return tab


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--stamp":
        t = stamp(sys.argv[2], sys.argv[3], sys.argv[4])
        n = sum(1 for wl in t["workloads"].values() for v in wl.values() if v.get("kernel_hash"))
        print("stamped %d entries" % n)
    else:
        for n, h in sorted(library_hashes(sys.argv[1]).items()):
            print(h, n)
