"""CPU: the C-ABI library builds, loads and exports every symbol include/dws.h
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "dws.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dws_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from diffwave_sashimi_amd import _lib
    from diffwave_sashimi_amd.build import build
    build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libdws.so lacks {n}"
    # the ctypes binding covers exactly the declared set
    assert sorted(_lib.EXPORTS) == names


def test_abi_identity():
    from diffwave_sashimi_amd import _lib
    lib = _lib.load()
    assert lib.dws_abi_version() == 1
    assert lib.dws_arch() == b"gfx950"


def test_model_create_introspection_matches_state_dict():
    """dws_model_create + param_info need no GPU kernels... but they do allocate
    device memory, so only the host-side mirror is checked here: the module's
    state_dict keys/shapes are what the engine will be handed."""
    from tests import cases
    cfg = cases.WAVENET_CASES["wn_tiny"][0]
    m = cases.build_ours(cfg, 1)
    keys = list(m.state_dict())
    assert "init_conv.0.conv.weight_g" in keys and "final_conv.2.conv.weight" in keys
    assert "residual_layer.residual_blocks.10.skip_conv.weight_v" in keys
    assert m.state_dict()["residual_layer.residual_blocks.0.dilated_conv_layer.conv.weight_v"].shape == (32, 16, 3)
    cfgc = cases.WAVENET_COND_CASES["wn_cond_tiny"][0]
    mc = cases.build_ours(cfgc, 1)
    sd = mc.state_dict()
    assert sd["residual_layer.residual_blocks.0.upsample_conv2d.0.weight_v"].shape == (1, 1, 3, 32)
    assert sd["residual_layer.residual_blocks.0.upsample_conv2d.1.weight_g"].shape == (1, 1, 1, 1)
    assert sd["residual_layer.residual_blocks.0.mel_conv.conv.weight_v"].shape == (32, 80, 1)


def test_forward_requires_gpu_and_never_falls_back():
    import torch
    from tests import cases
    cfg = cases.WAVENET_CASES["wn_tiny"][0]
    m = cases.build_ours(cfg, 1)
    with pytest.raises(RuntimeError, match="GPU only|no CPU fallback"):
        m((torch.zeros(1, 1, 64), torch.zeros(1, 1)))


def test_construct_model_restores_name_and_rejects_unknown():
    from diffwave_sashimi_amd.models import construct_model
    from tests import cases
    cfg = dict(cases.WAVENET_CASES["wn_tiny"][0])
    construct_model(cfg)
    assert cfg["_name_"] == "wavenet"
    with pytest.raises(KeyError):
        construct_model(dict(cfg, _name_="transformer"))


def test_cauchy_mult_module_is_importable_by_its_reference_name():
    """`extensions/cauchy/cauchy.py:5` does `from cauchy_mult import ...`; the standalone binding must import (library
    load + symbol binding, no GPU work) under exactly that top-level name and export the four entry points."""
    import importlib
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffwave-sashimi_amd", "extensions")
    sys.path.insert(0, d)
    try:
        m = importlib.import_module("cauchy_mult")
    finally:
        sys.path.remove(d)
    for name in ("cauchy_mult_fwd", "cauchy_mult_bwd", "cauchy_mult_sym_fwd", "cauchy_mult_sym_bwd"):
        assert callable(getattr(m, name))
    import torch
    import pytest
    with pytest.raises(RuntimeError):      # CHECK_DEVICE of `cauchy.cpp:6` without touching a GPU
        m.cauchy_mult_sym_fwd(torch.zeros(1, 4, dtype=torch.complex64), torch.zeros(3, dtype=torch.complex64),
                              torch.zeros(1, 4, dtype=torch.complex64))


def test_bench_executed_flops_formula_matches_the_counter():
    """`bench.py: wino_executed_work` (tile counts x MFMAs per wave x 4096) against what the hardware counted for the same
    launch (`SQ_INSTS_MFMA` per dispatch, the newest profiles/r*_wavenet_traffic.json): the figure `roofline.frac` is priced on."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import glob
    rec = json.load(open(sorted(glob.glob(os.path.join(root, "profiles", "r*_wavenet_traffic.json")))[-1]))
    assert rec["kernel"].startswith("wn_layer_wino_kernel")
    cfg = bench.CONFIGS["wnet_h256_d36_T200"]
    formula = bench.wino_executed_work(cfg)
    counted = rec["sq_insts_mfma_per_launch"] * 4096
    assert abs(formula - counted) < 5e-3 * counted, (formula, counted)
    algorithmic, _ = bench.layer_algorithmic_work(cfg)
    assert 0.74 < formula / algorithmic < 0.78          # (10 C^2 + 2 C S + padding) / (14 C^2 + 2 C S)


def test_every_device_kernel_is_registered_under_its_own_name():
    """A `__global__` function is registered by the host pass under its mangled name and looked up in the device code
    object under the DEVICE pass's mangled name: a parameter type that differs between the passes (a typedef that is a
    register-pair vector on the device and a struct on the host) builds fine and aborts at the first launch with "Cannot
    find Symbol" -- on the GPU box only.  So: every kernel descriptor (`<name>.kd`) of every embedded gfx950 code object
    must appear as a name string in the host half of libdws.so."""
    import os
    import re
    import shutil
    import subprocess
    import tempfile
    import pytest
    from diffwave_sashimi_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not shutil.which("strings"):
        pytest.skip("ROCm LLVM binutils not available")
    lib = _lib.LIB_PATH
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "stripped")], check=True)
        blob = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]     # one bundle per translation unit
        assert offs
        dev = set()
        for i, o in enumerate(offs):
            part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "b%d.co" % i)
            open(part, "wb").write(blob[o:(offs[i + 1] if i + 1 < len(offs) else len(blob))])
            subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + co], check=True)
            syms = subprocess.run([tools[2], "-sW", co], capture_output=True, text=True, check=True).stdout
            dev |= {l.split()[-1][:-3] for l in syms.splitlines() if l.split() and l.split()[-1].endswith(".kd")}
        host = set(subprocess.run(["strings", "-n", "6", lib], capture_output=True, text=True, check=True).stdout.splitlines())
    assert len(dev) > 100
    missing = sorted(dev - host)
    assert not missing, "device kernels the host never registers under that name: %s" % missing[:5]
