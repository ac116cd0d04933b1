"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol that
include/hat_mi355x.h declares (no compute calls: there is no GPU here), the ctypes mirror of
HatConvDesc has the C layout, and the host-side packing is a pure re-layout of the weights."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hat_mi355x.h")


@pytest.fixture(scope="module")
def lib_path():
    from super_resolution_amd import build
    if not os.path.exists(build.LIB):
        build.build(verbose=False)
    return build.LIB


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hat_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    from super_resolution_amd import _lib
    names = header_functions()
    assert len(names) >= 8
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    lib = C.CDLL(lib_path)
    for n in names:
        assert getattr(lib, n, None) is not None, f"{n} is declared in include/hat_mi355x.h but not exported"
    loaded = _lib.load()
    assert loaded.hat_abi_version() == 2
    assert loaded.hat_target_arch() == b"gfx950"
    assert loaded.hat_layernorm_blocks() > 0


@pytest.mark.parametrize("name", ["HatConvDesc", "HatFfnDesc", "HatCabFoldDesc", "HatAggrCabDesc"])
def test_desc_layout_matches_c(lib_path, tmp_path, name):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from super_resolution_amd import _lib
    S = getattr(_lib, name)
    fields = [f[0] for f in S._fields_]
    prog = f'#include <stdio.h>\n#include <stddef.h>\n#include "hat_mi355x.h"\nint main(){{printf("%zu", sizeof({name}));\n'
    prog += "".join(f'printf(" %zu", offsetof({name}, {f}));\n' for f in fields) + "return 0;}\n"
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert vals[0] == C.sizeof(S)
    assert vals[1:] == [getattr(S, f).offset for f in fields]


def test_rejects_bad_arguments_without_gpu(lib_path):
    from super_resolution_amd import _lib
    lib = _lib.load()
    d = _lib.HatConvDesc()
    assert lib.hat_conv(C.byref(d), None) == -1  # HAT_EINVAL: null pointers
    n = C.c_int32(0)
    d.H, d.W, d.Cin, d.ksize, d.nt, d.n_slices, d.dtype = 720, 1280, 144, 3, 9, 1, _lib.HAT_BF16
    assert lib.hat_conv_tiles(C.byref(d), C.byref(n)) == 0 and n.value == 80 * 45
    d.Cin, d.ksize = 4096, 13  # cannot fit 160 KiB of LDS
    assert lib.hat_conv_tiles(C.byref(d), C.byref(n)) == -2


def test_pack_conv_weight_is_a_relayout():
    from super_resolution_amd import ops
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    b = torch.tensor([1.0, 2.0])
    pw = ops.pack_conv_weight(w, b, ops.HAT_F32, "cpu")
    assert (pw.ksize, pw.cin, pw.nt, pw.n_slices, pw.nout) == (3, 3, 1, 1, 2)
    assert pw.kpad % 32 == 0 and pw.w.shape == (16, pw.kpad)
    for o in range(2):
        for tap in range(9):
            for ci in range(3):
                assert pw.w[o, tap * 8 + ci] == w[o, ci, tap // 3, tap % 3]
    assert pw.w.sum() == w.sum() and pw.bias[:2].tolist() == [1.0, 2.0] and pw.bias[2:].abs().sum() == 0
    assert ops.choose_nt(144) == (9, 1) and ops.choose_nt(576)[0] * ops.choose_nt(576)[1] * 16 == 576
    assert ops.choose_nt(6) == (1, 1) and ops.choose_nt(180) == (12, 1)


def test_product_path_has_no_cpu_fallback():
    import super_resolution_amd.archs  # noqa: F401
    from super_resolution_amd.registry import ARCH_REGISTRY, build_network
    assert "HAT" in ARCH_REGISTRY
    net = build_network(dict(type="HAT", upscale=2, embed_dim=24, depths=[1], num_heads=[2], window_size=8, mlp_ratio=2,
                             upsampler="pixelshuffle", esc_pdim=8, esc_kernel=5, unknown_key_is_swallowed=1)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 3, 16, 16))
    with pytest.raises(RuntimeError, match="eval"):
        net.train()(torch.rand(1, 3, 16, 16))
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "super_resolution_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_registry_semantics():
    from super_resolution_amd.registry import Registry
    r = Registry("t")

    @r.register()
    class A:
        pass

    r.register(dict, suffix="basicsr")
    assert r.get("A") is A and r.get("dict") is dict and "A" in r
    with pytest.raises(AssertionError):
        r.register(A)
    with pytest.raises(KeyError):
        r.get("missing")


def test_c_host_example_compiles_and_links_with_plain_gcc(tmp_path):
    """examples/plan_forward.c — the whole network from a C host through hat_plan_* — builds with gcc against the in-tree
    library and the HIP runtime (it is RUN on the GPU box by tests/test_gpu_plan.py)."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "plan_forward"
    r = subprocess.run(["gcc", os.path.join(root, "examples", "plan_forward.c"), "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
                        "-D__HIP_PLATFORM_AMD__", "-L" + os.path.join(root, "super_resolution_amd"), "-lhat_mi355x", "-L/opt/rocm/lib",
                        "-lamdhip64", "-Wl,-rpath," + os.path.join(root, "super_resolution_amd"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()
