"""Forward plans (include/hat_mi355x.h "Forward plans"; SURVEY §8b's whole-network entry point): the launch list the
Python engine produces for one shape, exported to a file and replayed through hat_plan_load / hat_plan_forward — the
three calls a host without Python makes — must reproduce `net(x)` bit for bit, for a fresh input, in both precisions and
for the fused (HAT-S) and unfused (C = 24, HATX) kernel sequences."""
import pytest
import torch

from helpers import META, W_SEED, X_SEED
from super_resolution_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [("HAT", "hats_1g_x4", "bf16", (1, 3, 32, 48)), ("HAT", "hats_1g_x4", "f32", (2, 3, 16, 32)),
                                  ("HAT", "tiny_ocabesc_x2", "bf16", (1, 3, 16, 24)), ("HATX", "hatx_tiny_plain_x2", "f32", (1, 3, 16, 24))],
                         ids=["hats_bf16", "hats_f32_B2", "tiny_ocabesc", "hatx"])
def test_plan_replay_is_bit_identical(case, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd import plan
    from super_resolution_amd.registry import build_network
    import super_resolution_amd.archs  # noqa: F401
    arch, name, dtype, shape = case
    dev = torch.device("cuda:0")
    net = build_network(dict(type=arch, compute_dtype=dtype, **META["cfgs"][name])).eval()
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
    net = net.to(dev)
    path = str(tmp_path / "net.hatplan")
    info = plan.export_plan(net, shape, path)
    assert info["launches"] > 10 and info["const_bytes"] > 0
    x = synth.synth_input(X_SEED + 1, shape).to(dev)        # not the tensor the plan was recorded with
    ref = net(x).clone()
    p = plan.Plan(path)
    assert p.dims[:4] == list(shape) and p.launches == info["launches"]
    s = META["cfgs"][name]["upscale"]
    y = torch.full((shape[0], 3, shape[2] * s, shape[3] * s), 5.0, device=dev)
    p.forward(x, y, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    y2 = torch.zeros_like(y)                                  # a second forward on the same plan (workspace reuse)
    p.forward(x, y2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(y2, ref)
    p.close()


def test_plan_load_rejects_garbage(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd import plan
    bad = tmp_path / "bad.hatplan"
    bad.write_bytes(b"NOTAPLAN" + b"\0" * 64)
    with pytest.raises(RuntimeError):
        plan.Plan(str(bad))
    with pytest.raises(RuntimeError):
        plan.Plan(str(tmp_path / "missing.hatplan"))


def test_c_host_program_runs_a_plan(tmp_path):
    """The C example (no Python in the process that computes): export a plan here, then run examples/plan_forward.c built with
    gcc as a child process on it and check that it reports the output mean the Python engine gets for the same input."""
    import os
    import re
    import shutil
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    if not shutil.which("gcc"):
        pytest.skip("needs gcc")
    from super_resolution_amd import plan
    from super_resolution_amd.registry import build_network
    import super_resolution_amd.archs  # noqa: F401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "plan_forward"
    r = subprocess.run(["gcc", os.path.join(root, "examples", "plan_forward.c"), "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
                        "-D__HIP_PLATFORM_AMD__", "-L" + os.path.join(root, "super_resolution_amd"), "-lhat_mi355x", "-L/opt/rocm/lib",
                        "-lamdhip64", "-Wl,-rpath," + os.path.join(root, "super_resolution_amd"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    dev = torch.device("cuda:0")
    shape = (1, 3, 32, 32)
    net = build_network(dict(type="HAT", compute_dtype="bf16", **META["cfgs"]["hats_1g_x4"])).eval()
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
    net = net.to(dev)
    path = str(tmp_path / "net.hatplan")
    plan.export_plan(net, shape, path)
    n = shape[0] * shape[1] * shape[2] * shape[3]
    x = ((torch.arange(n, dtype=torch.int64) * 2654435761) % 1000).to(torch.float32).div(1000.0).reshape(shape)   # the example's image (64-bit product)
    ref = float(net(x.to(dev)).double().mean())
    r = subprocess.run([str(exe), path, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"mean of the output (-?[0-9.]+)", r.stdout)
    assert m, r.stdout
    assert abs(float(m.group(1)) - ref) <= 2e-5, (r.stdout, ref)
