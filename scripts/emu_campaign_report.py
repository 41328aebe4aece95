#!/usr/bin/env python
"""Assemble the record of an emulator campaign (profiles/rNN_emu_campaign.txt) from the per-worker time reports that
tests/hip_emu/emu_prelude.py writes under BTS_HOST_TIME_REPORT and the pytest log.  CPU only.

    python scripts/emu_campaign_report.py <report prefix, e.g. /tmp/emu_final_times/t.json> <pytest log> <commit> [workers=6] > profiles/r06_emu_campaign.txt"""
import glob
import json
import re
import sys


def main():
    prefix, log, commit = sys.argv[1], sys.argv[2], sys.argv[3]
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    rows = {}
    for f in sorted(glob.glob(prefix + "*")):
        rows.update(json.load(open(f)))
    txt = open(log).read()
    m = re.search(r"(\d+ passed[^\n]*)", txt)
    verdict = m.group(1) if m else "no verdict line found"
    failed = re.findall(r"^FAILED (\S+)", txt, flags=re.M)
    print("Round 6 emulator campaign at %s: BTS_HOST_TIME_REPORT=... BTS_LIB_PATH=<copy of tests/hip_emu/libbotsort_emu.so> pytest tests -m gpu "
          "-p tests.hip_emu.emu_prelude -n %d" % (commit, workers))
    print(verdict + " on 8 cores (%d workers)" % workers + ("; FAILED: " + ", ".join(failed) if failed else ""))
    print()
    print("The emulator of this round: global_load_lds is ASYNCHRONOUS (data lands at the issuing wavefront's explicit s_waitcnt vmcnt, the destination")
    print("is poisoned until then, another wavefront may read it only behind a workgroup barrier; every inline-assembly LDS access of the kernels is")
    print("checked against the transfers in flight and a hazard aborts the process), an empty launch configuration aborts like the hardware refuses it,")
    print("red zones behind every table / work-area array as in round 5.  No hazard, no red-zone hit, no empty launch in any test.")
    print()
    print("host_s = wall time of the test OUTSIDE the library's entry points (oracle, stream generation, NumPy, pytest) -- what the test costs on a GPU box,")
    print("where the library calls take milliseconds; library_s = time inside the emulated library (CPU emulation: says nothing about a GPU).  Measured with")
    print("the workers sharing 8 cores (BLAS threads oversubscribed): an UPPER bound (round 5 re-measured the heaviest entries uncontended: ~10x lower).")
    print("Driver limit for the suite: 1200 s.")
    print()
    hs = sum(r["host_s"] for r in rows.values())
    ls = sum(r["library_s"] for r in rows.values())
    print("tests %d   sum host_s %.1f   sum library_s %.1f" % (len(rows), hs, ls))
    print()
    print("   host_s  library_s  test")
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["host_s"]):
        print("%9.1f  %9.1f  %s" % (r["host_s"], r["library_s"], name))


if __name__ == "__main__":
    main()
