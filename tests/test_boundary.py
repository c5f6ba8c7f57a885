"""CPU: the drop-in boundary -- state-dict compatibility, the C ABI surface, loud failure
without a GPU.  No compute calls."""
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_matches_reference_keys():
    from pips_amd import Pips
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        ref = json.load(f)
    sd = Pips(S=8, stride=8).state_dict()
    assert list(sd.keys()) == list(ref.keys())                 # same names, same order
    assert {k: list(v.shape) for k, v in sd.items()} == ref
    assert sum(v.numel() for v in sd.values()) == 28677713


def test_load_reference_format_checkpoint(tmp_path, weights_tamed):
    """saverloader.py:58-59 style: torch.load(...)['model_state_dict'] -> load_state_dict(strict=False)."""
    from pips_amd import Pips
    path = tmp_path / "model-000000001.pth"
    torch.save({"model_state_dict": weights_tamed, "optimizer_state_dict": {}}, path)
    ck = torch.load(path)
    m = Pips(stride=4)
    res = m.load_state_dict(ck["model_state_dict"], strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    k = "delta_block.to_delta.15.weight"
    assert torch.equal(m.state_dict()[k], weights_tamed[k])


def test_library_exports_every_declared_symbol():
    import ctypes
    from pips_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pips_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pips_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    assert declared == set(_lib.SIGNATURES)                    # binding table mirrors the header
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.pips_abi_version() == 3        # 3: slack behind the bf16 mirror, larger tiled-gather scratch, PIPS_EPI_RES_BF16 public (2: bf16 mirror in the pyramid buffer)
    assert lib.pips_weight_arena_bytes() > 28677713 * 4
    # sizing queries are pure host functions
    assert lib.pips_workspace_bytes(1, 8, 368, 496, 256, 8) > 0
    assert lib.pips_workspace_bytes(1, 7, 368, 496, 256, 8) > 0           # any window length 1..PIPS_S_MAX
    assert lib.pips_workspace_bytes(1, 33, 368, 496, 256, 8) == 0 and lib.pips_weight_arena_bytes_s(33) == 0
    assert lib.pips_weight_arena_bytes_s(8) == lib.pips_weight_arena_bytes()
    assert lib.pips_weight_arena_bytes_s(5) < lib.pips_weight_arena_bytes() < lib.pips_weight_arena_bytes_s(12)
    assert lib.pips_delta_stride(8) == 1040 and lib.pips_delta_stride(5) == 652 and lib.pips_delta_stride(0) == 0
    assert lib.pips_track_workspace_bytes_s(2, 64, 8) == lib.pips_track_workspace_bytes(2, 64)
    assert lib.pips_mixer_workspace_bytes_s(1024, 8) == lib.pips_mixer_workspace_bytes(1024)
    assert lib.pips_pyramid_offset(8, 368, 496, 8, 1) == 8 * 46 * 62 * 128
    assert lib.pips_pyramid_floats(8, 368, 496, 8) >= 8 * 128 * (46 * 62 + 23 * 31 + 11 * 15 + 5 * 7)


def test_no_cpu_fallback():
    from pips_amd import Pips, PipsHipError
    m = Pips()
    with pytest.raises(PipsHipError):
        m(torch.zeros(1, 4, 2), torch.zeros(1, 8, 3, 128, 160), iters=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pips_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} references the oracle"


def test_reference_import_path_resolves_to_the_hip_model():
    """Every reference caller does `from nets.pips import Pips` (demo.py:9, chain_demo.py:9, test_on_flt.py:6, ...)."""
    import nets.pips
    import pips_amd
    assert nets.pips.Pips is pips_amd.Pips


def test_nets_is_a_namespace_package_merged_with_the_reference():
    """``nets`` ships no __init__.py: with this repository ahead of the reference checkout on sys.path the reference's
    sibling modules stay importable (``from nets.raftnet import Raftnet`` is line 5 of test_on_flt.py / test_on_crohd.py
    / test_on_badja.py, ahead of ``from nets.pips import Pips``) while nets.pips resolves to the HIP model."""
    import subprocess, sys
    assert not os.path.exists(os.path.join(ROOT, "nets", "__init__.py"))
    ref_root = "/root/reference"
    have_ref = os.path.isfile(os.path.join(ref_root, "nets", "raftnet.py"))
    code = ("import sys, importlib.util\n"
            f"sys.path[:0] = [{ROOT!r}, {ref_root!r}]\n"
            "import nets.pips, pips_amd\n"
            "assert nets.pips.Pips is pips_amd.Pips\n"
            "assert nets.pips.__file__.startswith(%r)\n" % ROOT +
            (f"spec = importlib.util.find_spec('nets.raftnet')\nassert spec is not None and spec.origin.startswith({ref_root!r}), spec\n"
             if have_ref else "") + "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_reference_saverloader_round_trip(tmp_path, weights_tamed):
    """The reference's own saverloader.save / saverloader.load (saverloader.py:5-69), unmodified, on pips_amd.Pips
    (build container only: needs /root/reference)."""
    ref_root = "/root/reference"
    if not os.path.isfile(os.path.join(ref_root, "saverloader.py")):
        pytest.skip("reference checkout not mounted")
    import importlib.util
    import sys
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_reference_saverloader", os.path.join(ref_root, "saverloader.py"))
    saverloader = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(saverloader)
    from nets.pips import Pips
    src = Pips(stride=8)
    src.load_state_dict(weights_tamed)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    ckpt_dir = str(tmp_path / "ckpt")
    saverloader.save(ckpt_dir, opt, src, global_step=7)
    dst = Pips(stride=8)                                        # seeded init != tamed weights
    assert saverloader.load(ckpt_dir, dst) == 7
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)


def test_packed_weights_follow_parameter_changes():
    """The kernel-side arena is rebuilt when storage or version of a parameter changes; .data writes need invalidate."""
    from pips_amd import Pips
    m = Pips()
    assert m._plist is None and m._arena is None
    m.invalidate_weights()
    m2 = m.float()                                             # _apply path
    assert m2 is m and m._arena is None
    # a REPLACED parameter (load_state_dict(assign=True), node.weight = nn.Parameter(...)) changes the cache key too:
    # the key is built from the live objects, not from a list captured at the first forward
    import pips_amd.ops as ops
    packed = []
    orig = ops.pack_weights
    ops.pack_weights = lambda sd, dev, sections=7, S=8: packed.append({k: v for k, v in sd.items()}) or object()
    try:
        m._packed("cpu")
        m._packed("cpu")
        assert len(packed) == 1
        node = m.delta_block.to_delta._modules["15"]
        node.weight = torch.nn.Parameter(node.weight.detach().clone() * 2, requires_grad=False)
        m._packed("cpu")
        assert len(packed) == 2 and packed[1]["delta_block.to_delta.15.weight"] is node.weight
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_state_dict(sd, assign=True)
        m._packed("cpu")
        assert len(packed) == 3
    finally:
        ops.pack_weights = orig


def test_build_script_dependencies_exist_and_asm_is_current(tmp_path):
    """The build script's dependency list names real files (a stale name makes every rebuild fail), and the committed
    generated assembly is what tools/gen_gather_asm.py / tools/gen_gemm_bf16_t4.py / tools/gen_gemm_bf16_t4up.py emit today."""
    import subprocess, sys
    from pips_amd import _build
    for h in _build.headers():
        assert os.path.exists(h), h
    out = tmp_path / "gather_item_asm.inc"
    env = dict(os.environ, PIPS_GEN_OUT=str(out))
    for k in ("PIPS_GEN_ABLATE", "PIPS_GEN_TRACE"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_gather_asm.py")], env=env, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(root, "pips_amd", "csrc", "gather_item_asm.inc")).read()
    out3 = tmp_path / "gemm_bf16_t4_asm.inc"
    env["PIPS_GEN_OUT"] = str(out3)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_gemm_bf16_t4.py")], env=env, stdout=subprocess.DEVNULL)
    assert out3.read_text() == open(os.path.join(root, "pips_amd", "csrc", "gemm_bf16_t4_asm.inc")).read()
    out4 = tmp_path / "gemm_bf16_t4up_asm.inc"
    env["PIPS_GEN_OUT"] = str(out4)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_gemm_bf16_t4up.py")], env=env, stdout=subprocess.DEVNULL)
    assert out4.read_text() == open(os.path.join(root, "pips_amd", "csrc", "gemm_bf16_t4up_asm.inc")).read()
    out5 = tmp_path / "conv_bf16_t4c_asm.inc"
    env["PIPS_GEN_OUT"] = str(out5)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_conv_bf16_t4c.py")], env=env, stdout=subprocess.DEVNULL)
    assert out5.read_text() == open(os.path.join(root, "pips_amd", "csrc", "conv_bf16_t4c_asm.inc")).read()
    for gen, inc in (("gen_gemm_f32_t4.py", "gemm_f32_t4_asm.inc"), ("gen_conv_f32_t4.py", "conv_f32_t4_asm.inc")):
        out6 = tmp_path / inc
        env["PIPS_GEN_OUT"] = str(out6)
        subprocess.check_call([sys.executable, os.path.join(root, "tools", gen)], env=env, stdout=subprocess.DEVNULL)
        assert out6.read_text() == open(os.path.join(root, "pips_amd", "csrc", inc)).read()


def _lint():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("asm_hazard_lint", os.path.join(root, "tools", "asm_hazard_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, root


def test_asm_hazard_lint_passes_head_and_flags_the_round5_scatter():
    """Every instruction stream of the shipped library -- hipcc's code and the 37 k lines of generated / hand-written assembly that
    its hazard recognizer never sees -- holds the software wait states of tools/asm_hazard_lint.py's table (MFMA result -> any
    reader or writer, VALU-written SGPR / VCC -> VALU / vector memory / lane select, M0 -> LDS-DMA, ...): the disassembly of
    libpips_hip.so has no finding.  The negative fixture is the scatter of gather_mfma_kernel as commit a4e2782~1 built it -- the MFMA ->
    DS-read hazard round 5 met by accident: the lint must flag its four ds_write_b32."""
    L, root = _lint()
    from pips_amd import _build
    findings, st = L.lint_path(_build.build_library(verbose=False))
    assert st["kernels"] > 100 and st["mfma"] > 10000, st          # the whole library was read
    assert findings == [], findings[:5]
    neg, _ = L.lint_path(os.path.join(root, "tests", "golden", "hazard_negative_gather_mfma_pre_a4e2782.s"))
    assert len(neg) == 4 and all("v_mfma_f32_32x32x16_bf16 result -> read by ds_write_b32" in f["rule"] for f in neg), neg
    assert sorted(f["have"] for f in neg) == [4, 6, 8, 10] and all(f["need"] == 12 for f in neg)


def test_hazard_table_matches_the_compilers_recognizer():
    """gfx950's wait-state table is not in this image; the lint's numbers are pinned against hipcc's own hazard recognizer: one probe
    kernel per producer / consumer pair (tools/asm_hazard_probe.hip, builtins + sched_barriers), the s_nops hipcc inserts are the
    requirement."""
    L, _ = _lint()
    rows = L.measure_probes()
    assert len(rows) >= 28
    bad = [r for r in rows if r[1] != r[2]]
    assert not bad, bad


def test_generators_take_their_wait_states_from_one_place():
    """The five pasted `s_nop 15; s_nop 15` pairs of rounds 4-5 are gone: every generator takes its guards from tools/asm_guards.py,
    whose numbers are the lint's table."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gens = sorted(glob.glob(os.path.join(root, "tools", "gen_*.py")))
    assert len(gens) == 6
    for g in gens:
        src = open(g).read()
        assert "import asm_guards" in src and "s_nop 15" not in src, g


def test_profiles_readme_is_generated_from_the_files():
    """The current round's table of profiles/README.md is what tools/profiles_readme.py generates from the committed evidence files
    (rounds 3-5 copied their numbers by hand and the prose drifted from the files three times)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "profiles_readme.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_gather_mfma2_variant_builds_and_passes_the_lint(tmp_path):
    """gather_mfma2_kernel (round 6's re-cut of the bf16 dense gather: LDS-DMA requests as assembly statements with counted waits) is not in
    the product library -- it is 3-4 % slower than gather_mfma_kernel -- but stays buildable for A/B runs (-DPIPS_GM_V_DEFAULT=2): the variant
    object compiles for gfx950 and its hand-written request statements hold their wait states (M0 -> LDS-DMA, fresh SGPR -> vector memory)."""
    import subprocess
    L, root = _lint()
    from pips_amd import _build
    obj = tmp_path / "gather_gm2.o"
    cmd = [_build._hipcc(), *_build.FLAGS, "-DPIPS_GM_V_DEFAULT=2", "-c", os.path.join(_build.CSRC, "gather_tiled.hip"), "-o", str(obj)]
    subprocess.check_call(cmd)
    findings, st = L.lint_path(str(obj))
    assert st["kernels"] >= 6 and findings == [], findings[:5]
    dis = "\n".join(L.disassemble(str(obj)))
    assert "gather_mfma2_kernel" in dis and dis.count("offen lds") >= 16        # the requests are there (two blocks of eight pieces + records)
