#!/usr/bin/env python
"""Dependent-memory chain of a kernel's opening, read off the gfx950 disassembly (CPU only).

For a kernel of libbotsort_hip.so this walks the control-flow graph of its machine code from the entry to the first instruction
matching --until (default: the first v_mfma) along FORWARD edges (a loop is taken zero times: the opening of a workgroup, not its
steady state), simulating the two counters the hardware orders memory with:

    vmcnt    vector memory: global_load* ('L'), global_load_lds* ('D': LDS-DMA), stores / atomics ('S'); returns IN ORDER
    lgkmcnt  scalar loads s_load* ('K': the kernel-argument segment and what hangs off it) and LDS operations ('M')

A `s_waitcnt` that retires at least one outstanding LOAD is one DEPENDENT ROUND TRIP of the chain (everything issued before it
overlaps; what is issued behind it cannot start earlier): 'K' = a scalar-cache / L2 round trip of the kernel-argument segment,
'L' = a vector load (cold: HBM ~900 cycles, L2 hit ~200; MI355X_MICROARCH.md), 'D' = an operand stage through the LDS-DMA path.
Waits that retire only LDS reads are counted separately (~64-128 cycles each), so are workgroup barriers.

Output per kernel: over all forward paths that reach --until, the minimum and maximum number of dependent round trips by kind,
the instructions executed on the way, and the wait sequence of the longest chain -- the PREDICTED decomposition of the kernel's
fixed opening that the trace build (ab/ftrace.so, scripts/dbg_filtertrace.py) confirms or refutes on a device.

    python scripts/isa_chain.py [--lib path] [--until REGEX] [--from-label N] KERNEL_REGEX ...
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSN = re.compile(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
HEAD = re.compile(r"^([0-9a-f]+) <(.*)>:$")


def disassemble(lib, patterns=None):
    """{demangled kernel name: [(address, mnemonic, operands), ...]} of the gfx950 code objects in `lib`; with `patterns` only the
    code objects whose symbol table holds a match are disassembled (the library has eight; a kernel lives in one)."""
    d = tempfile.mkdtemp(prefix="isa_chain_")
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    kernels = {}
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        if patterns:
            syms = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", "-C", os.path.join(d, f)], capture_output=True, text=True).stdout
            if not any(re.search(p, syms) for p in patterns):
                continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", os.path.join(d, f)], capture_output=True, text=True, check=True).stdout
        name = None
        for ln in txt.splitlines():
            m = HEAD.match(ln)
            if m:
                name = m.group(2)
                kernels[name] = []
                continue
            m = INSN.match(ln)
            if m and name is not None:
                kernels[name].append((int(m.group(3), 16), m.group(1), m.group(2)))
    shutil.rmtree(d, ignore_errors=True)
    return kernels


def kind_of(mn):
    if mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and "lds" in mn):
        return "vm", "D"
    if mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vm", "L"
    if mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vm", "S"
    if mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "vm", "S"            # (returning atomics are loads too; none sits in an opening chain here)
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "lgkm", "K"
    if mn.startswith("ds_"):
        return "lgkm", "M"
    return None, None


BACK = 2


def walk(insns, until, start_idx=0):
    """Worklist over the CFG.  state = (vm queue, lgkm queue) as strings; value = the best (min) and worst (max) tallies with which
    the state was reached.  A path may take at most BACK backward branches inside the body (the compiler rotates loops: the first
    MFMA of a K loop sits in FRONT of the loop header that is entered first) -- enough to enter every loop once, not to iterate;
    backward branches out of the trampoline region behind the kernel's last s_endpgm (long forward jumps land there and are sent
    back into the body) are free."""
    addr2i = {a: i for i, (a, _, _) in enumerate(insns)}
    n = len(insns)
    last_end = max([i for i, x in enumerate(insns) if x[1] == "s_endpgm"] or [n])
    nodes = [dict() for _ in range(n)]
    dirty = [set() for _ in range(n)]
    zero = (0, 0, 0, 0, 0, 0, ())
    nodes[start_idx][("", "", 0)] = {"min": zero, "max": zero}
    dirty[start_idx].add(("", "", 0))
    work = [start_idx]
    results = []

    def key(t):
        return (t[0] + t[1] + t[2], t[5])

    def push(j, st, tmin, tmax):
        cur = nodes[j].get(st)
        changed = False
        if cur is None:
            if len(nodes[j]) > 3000:
                return                   # (state explosion guard)
            nodes[j][st] = {"min": tmin, "max": tmax}
            changed = True
        else:
            if key(tmin) < key(cur["min"]):
                cur["min"] = tmin
                changed = True
            if key(tmax) > key(cur["max"]):
                cur["max"] = tmax
                changed = True
        if changed:
            if not dirty[j]:
                work.append(j)
            dirty[j].add(st)

    # every instruction decoded once: (class, vm kind, lgkm kind, vmcnt, lgkmcnt, is_until, successors)
    RE_VM, RE_LG = re.compile(r"vmcnt\((\d+)\)"), re.compile(r"lgkmcnt\((\d+)\)")
    dec = []
    for i, (a, mn, ops) in enumerate(insns):
        cnt, kd = kind_of(mn)
        vmc = lgc = None
        if mn == "s_waitcnt":
            m = RE_VM.search(ops)
            vmc = int(m.group(1)) if m else None
            m = RE_LG.search(ops)
            lgc = int(m.group(1)) if m else None
        if mn == "s_endpgm" or mn.startswith(("s_setpc", "s_swappc")):
            succ = ()
        elif mn == "s_branch":
            succ = (a + 4 + 4 * ((int(ops.split()[0]) + 0x8000) % 0x10000 - 0x8000),)
        elif mn.startswith("s_cbranch"):
            succ = (a + 4 + 4 * ((int(ops.split()[0]) + 0x8000) % 0x10000 - 0x8000), insns[i + 1][0] if i + 1 < n else None)
        else:
            succ = (insns[i + 1][0] if i + 1 < n else None,)
        succ = tuple(addr2i[x] for x in succ if x is not None and x in addr2i)
        dec.append((cnt, kd, mn == "s_waitcnt", vmc, lgc, mn == "s_barrier", bool(re.search(until, mn)), succ))

    def step(t, vm, lg, i):
        k_, l_, d_, w_, b_, c_, tr = t
        cnt, kd, is_wait, vmc, lgc, is_bar, _, _ = dec[i]
        c_ += 1
        if cnt == "vm":
            vm = vm + kd
        elif cnt == "lgkm":
            lg = lg + kd
        elif is_wait:
            a = insns[i][0]
            if vmc is not None and len(vm) > vmc:
                gone, vm = vm[:len(vm) - vmc], vm[len(vm) - vmc:]
                if "L" in gone:
                    l_ += 1
                    tr = tr + (("L", a, gone),)
                elif "D" in gone:
                    d_ += 1
                    tr = tr + (("D", a, gone),)
            if lgc is not None and len(lg) > lgc:
                # scalar loads return out of order: only lgkmcnt(0) guarantees them (a wait that must cover one is a full wait)
                gone, lg = (lg, "") if (lgc == 0 or "K" in lg) else (lg[:len(lg) - lgc], lg[len(lg) - lgc:])
                if "K" in gone:
                    k_ += 1
                    tr = tr + (("K", a, gone),)
                elif "M" in gone:
                    w_ += 1
        elif is_bar:
            b_ += 1
        return (k_, l_, d_, w_, b_, c_, tr), vm, lg

    while work:
        i = work.pop()
        todo, dirty[i] = dirty[i], set()
        is_until, succ = dec[i][6], dec[i][7]
        for st in todo:
            val = nodes[i][st]
            vm, lg, nb = st
            if is_until:
                results.append((val["min"][:5] + (val["min"][5] + 1,) + val["min"][6:], val["max"][:5] + (val["max"][5] + 1,) + val["max"][6:]))
                continue
            tmin, vm2, lg2 = step(val["min"], vm, lg, i)
            tmax = tmin if val["max"] is val["min"] or val["max"] == val["min"] else step(val["max"], vm, lg, i)[0]
            for j in succ:
                back = 1 if (j <= i and i <= last_end) else 0
                if nb + back > BACK:
                    continue
                push(j, (vm2, lg2, nb + back), tmin, tmax)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernels", nargs="+", help="regular expressions on demangled kernel names")
    ap.add_argument("--lib", default=os.path.join(REPO, "bot-sort-onnx-tensorrt_amd", "libbotsort_hip.so"))
    ap.add_argument("--until", default=r"^v_mfma", help="regex on the mnemonic that ends the opening (default: first MFMA)")
    ap.add_argument("--skip-branch-taken", type=int, default=0,
                    help="start behind the N-th conditional branch of the kernel, TAKEN (k_filter_*: 1 = the GEMM workgroups' side of the predict / GEMM split)")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    a = ap.parse_args()
    ks = disassemble(a.lib, a.kernels)
    rc = 0
    for pat in a.kernels:
        names = [k for k in ks if re.search(pat, k)]
        if not names:
            print("no kernel matches %r" % pat)
            rc = 1
        for name in names:
            insns = ks[name]
            start = 0
            if a.skip_branch_taken:
                seen = 0
                for i, (ad, mn, ops) in enumerate(insns):
                    if mn.startswith("s_cbranch"):
                        seen += 1
                        if seen == a.skip_branch_taken:
                            tgt = ad + 4 + 4 * ((int(ops.split()[0]) + 0x8000) % 0x10000 - 0x8000)
                            start = [j for j, x in enumerate(insns) if x[0] == tgt][0]
                            break
            res = walk(insns, a.until, start)
            print("## %s" % name)
            print("   %d instructions (%d bytes of code); paths reaching /%s/: %d end points" % (len(insns), insns[-1][0] + 4 - insns[0][0], a.until, len(res)))
            if not res:
                continue
            lo = min((r[0] for r in res), key=lambda t: (t[0] + t[1] + t[2], t[5]))
            hi = max((r[1] for r in res), key=lambda t: (t[0] + t[1] + t[2], t[5]))
            for tag, t in (("shortest", lo), ("longest", hi)):
                k_, l_, d_, w_, b_, c_, tr = t
                cold = (k_ + l_ + d_) * 900 / a.clock_ghz / 1e3
                warm = (k_ * 200 + (l_ + d_) * 500) / a.clock_ghz / 1e3
                print("   %-8s chain: %d dependent round trips (kernel-argument segment %d, vector loads %d, LDS-DMA stages %d) + %d LDS waits + %d barriers"
                      " over %d instructions" % (tag, k_ + l_ + d_, k_, l_, d_, w_, b_, c_))
                print("            = %.2f us if every trip is an HBM miss (900 cycles at %.1f GHz), %.2f us with the argument segment in L2 (200) and the rest at ~500;"
                      " + %.2f us of issue at 1 instruction / 4 cycles" % (cold, a.clock_ghz, warm, c_ * 4 / a.clock_ghz / 1e3))
                print("            waits: " + "  ".join("%s@+0x%x(%s)" % (k, ad - insns[0][0], "".join(sorted(set(g)))) for k, ad, g in tr))
    return rc


if __name__ == "__main__":
    sys.exit(main())
