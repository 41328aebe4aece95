"""Static properties of the gfx950 code objects inside libbotsort_hip.so, read on the CPU from the kernel metadata and the
disassembly (llvm-readelf / llvm-objdump of /opt/rocm): no kernel of the step spills or uses scratch (measured cost of ANY
scratch use: ~2 us per launch), the contraction is real MFMA code fed by LDS-DMA, nothing is built for
another architecture.  Guards against the silent regressions this build has seen (a co-compiled template variant, a
by-value struct select, a runtime-indexed register array)."""
import glob
import importlib
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels that are allowed a scratch frame, with the reason (bytes are upper bounds)
SCRATCH_OK = {
    "tk_edges_tall_gate_c": 64,      # the 256-row gated tile keeps the 128-register cap and spills 12 B (52 vs 65 us uncapped)
    "k_lap_exact": 64,               # the exact-tie step (tie_mode 1 only, one workgroup, off the default path): 12 B, its cost is the replay's
}                                    # (keys are EXACT kernel names without template arguments: "k_lap" would not cover "k_lap_exact")


def _base_names(mangled):
    """mangled kernel symbols -> their demangled names without namespace, template arguments and parameter list"""
    p = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True)
    out = {}
    for m, d in zip(mangled, p.stdout.splitlines()):
        d = re.sub(r"\(.*$", "", d.replace("(anonymous namespace)::", "")).strip()
        out[m] = re.sub(r"^void\s+", "", d).split("<")[0]
    return out


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("no ROCm LLVM tools")
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    if not os.path.exists(lib_mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    d = tmp_path_factory.mktemp("cox")          # (llvm-objdump --offloading writes next to its input: work on a copy)
    shutil.copy(lib_mod.LIB_PATH, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    objs = sorted(glob.glob(str(d / "lib.so.*amdgcn*")))
    assert objs, "no device code objects found"
    return objs


def _kernels(objs):
    out = {}
    for o in objs:
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], capture_output=True, text=True, check=True).stdout
        for block in txt.split(".agpr_count:")[1:]:                  # one block per kernel (first key of each entry)
            f = dict(re.findall(r"\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|group_segment_fixed_size):\s+(\S+)", block))
            if "name" in f:
                out[f["name"]] = {k: int(v) for k, v in f.items() if k != "name"}
    return out


def test_only_gfx950_code(code_objects):
    assert all(o.endswith("gfx950") for o in code_objects), code_objects


def test_no_kernel_spills_or_uses_scratch(code_objects):
    ks = _kernels(code_objects)
    assert len(ks) >= 60, len(ks)
    offenders = []
    base = _base_names(sorted(ks))
    assert {"k_lap", "k_lap_exact", "k_filter_i8", "tk_front"} <= set(base.values()), sorted(set(base.values()))
    for name, f in ks.items():
        scratch, spills = f.get("private_segment_fixed_size", 0), f.get("vgpr_spill_count", 0)      # (SGPR spills go to VGPR lanes, not memory)
        if scratch == 0 and spills == 0:
            continue
        cap = SCRATCH_OK.get(base[name])
        if cap is None or scratch > cap:
            offenders.append((base[name], name, scratch, spills))
    assert not offenders, offenders
    assert max(f.get("group_segment_fixed_size", 0) for f in ks.values()) <= 160 * 1024


def test_contraction_is_mfma_fed_by_lds_dma(code_objects):
    asm = "".join(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", o], capture_output=True, text=True, check=True).stdout for o in code_objects)
    assert len(re.findall(r"v_mfma_i32_32x32x32_i8", asm)) >= 32          # int8 pre-filter (k_filter_i8, k_filter_splitk, face forms)
    assert len(re.findall(r"v_mfma_f32_32x32x2_f32", asm)) >= 8           # exact float32 similarity GEMM (dense mode / standalone)
    assert len(re.findall(r"global_load_lds_dwordx4", asm)) >= 32         # operand stages: memory -> LDS, 16 B per lane
    assert len(re.findall(r"v_pk_(?:min|max)_i16", asm)) >= 64            # packed 16-bit box screening of the cost tiles
    assert "v_mfma_scale_f32_32x32x64_f8f6f4" not in asm                  # the fp8 form is gone (round 2)


def test_every_header_is_a_make_dependency():
    """A header missing from the Makefile's HDRS leaves a stale object behind after an edit (round 3: lap_exact.h -- the
    library shipped an older k_lap_exact than the tests had checked on the CPU until this was noticed)."""
    csrc = os.path.join(REPO, "bot-sort-onnx-tensorrt_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    hdrs = re.search(r"^HDRS\s*=\s*(.*)$", mk, flags=re.M).group(1).split()
    have = sorted(os.path.basename(p) for p in glob.glob(os.path.join(csrc, "*.h")))
    assert sorted(h for h in hdrs if "/" not in h) == have, (hdrs, have)


def test_prefilter_tile_shapes_fit_their_occupancy(code_objects):
    """k_filter_t (BTS_FILTER_TILE): what each shape was built FOR is a property of its code object -- two resident workgroups per
    CU for the 160 x 128 tiles (<= 256 registers per lane, <= 80 KB of LDS), three for the 128 x 64 tiles (<= 168 registers,
    <= 53.3 KB), no scratch (any scratch use measured ~2 us per launch in round 2) -- and can be checked without a GPU."""
    ks = _kernels(code_objects)
    dem = dict(zip(sorted(ks), subprocess.run(["c++filt"], input="\n".join(sorted(ks)), capture_output=True, text=True, check=True).stdout.splitlines()))
    want = {"k_filter_t<160, 128, 1, 4, 2, 1>": (256, 160 * 1024 // 2), "k_filter_t<128, 64, 2, 2, 3, 1>": (168, 160 * 1024 // 3)}
    seen = set()
    for name, f in ks.items():
        for key, (vmax, lmax) in want.items():
            if key in dem[name]:
                seen.add(key)
                assert f["vgpr_count"] <= vmax, (key, f)
                assert f["group_segment_fixed_size"] <= lmax, (key, f)
                assert f.get("private_segment_fixed_size", 0) == 0 and f.get("vgpr_spill_count", 0) == 0, (key, f)
    assert seen == set(want), seen
    # the K loop of every shape: one MFMA per 32 x 32 block and K step (4 K steps per stage), fed by LDS-DMA
    for o in code_objects:
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", o], capture_output=True, text=True, check=True).stdout
        for sym, blocks, loads in (("k_filter_tILi160ELi128", 5, 9), ("k_filter_tILi128ELi64", 2, 6)):
            m = re.search(r"<[^>]*%s[^>]*>:\n(.*?)(?=\n\n|\Z)" % sym, asm, flags=re.S)
            if not m:
                continue
            body = m.group(1)
            assert len(re.findall(r"v_mfma_i32_32x32x32_i8", body)) == 4 * blocks, sym
            assert len(re.findall(r"global_load_lds_dwordx4", body)) >= 2 * loads, sym       # (stage + next-tile first stage + records)


def test_prefilter_opens_with_one_lds_dma_drain():
    """Round 6, read off the ISA (scripts/isa_chain.py): behind an LDS-DMA the compiler turns the wait for ANY vector
    load into a full drain (vmcnt(0)).  With the record transfer issued behind the next tile's index loads, and the index-derived
    record offset sunk below the first stage, a tile of the pre-filter opened with TWO serial LDS-DMA drains in front of its first
    MFMA (first stage, then the records).  The source now issues the records first and materialises the offsets before the first
    LDS-DMA: ONE drain on the straight path of the tile workgroups (same change in k_filter_t).  A change that
    brings the second drain back (a new load consumed between the first stage and the stage barrier) shows up here, on the CPU."""
    import importlib.util
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no ROCm LLVM tools")
    spec = importlib.util.spec_from_file_location("isa_chain", os.path.join(REPO, "scripts", "isa_chain.py"))
    ic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ic)
    lib_mod = importlib.import_module("bot-sort-onnx-tensorrt_amd._lib")
    # (the default kernel only: its first MFMA is one loop-entry away from the entry, a walk of seconds; the 160 x 128 shape needs two
    #  and minutes -- its figures are in profiles/r06_isa_chain_kfilter.txt, from the same tool)
    pats = ((r"k_filter_i8<false, 0, false, false>", 3),)
    ks = ic.disassemble(lib_mod.LIB_PATH, [p for p, _ in pats])
    ic.BACK = 1
    for pat, max_trips in pats:
        name = [k for k in ks if re.search(pat, k)]
        assert len(name) == 1, (pat, name)
        insns = ks[name[0]]
        # start behind the predict / tile split: the first conditional branch of the kernel, taken = the tile workgroups' side
        first = next(i for i, x in enumerate(insns) if x[1].startswith("s_cbranch"))
        tgt = insns[first][0] + 4 + 4 * ((int(insns[first][2].split()[0]) + 0x8000) % 0x10000 - 0x8000)
        start = next(j for j, x in enumerate(insns) if x[0] == tgt)
        res = ic.walk(insns, r"^v_mfma", start)
        assert res, pat
        k_, l_, d_, w_, b_, c_, tr = min((r[0] for r in res), key=lambda t: (t[0] + t[1] + t[2], t[5]))
        assert d_ == 1, (pat, "LDS-DMA drains before the first MFMA on the straight path", d_, tr)
        assert k_ + l_ + d_ <= max_trips and b_ == 1, (pat, k_, l_, d_, b_)
