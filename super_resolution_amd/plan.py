"""Forward plans: export one input shape's complete HAT forward for hosts without Python.

    python -m super_resolution_amd.plan -opt options/test/x.yml --shape 1 720 1280 -o net_720p.hatplan
    plan.export_plan(net, (1, 3, 720, 1280), "net_720p.hatplan")

The exporter runs ONE forward of the engine on the GPU with a recording proxy in front of the C ABI: every launch call
(`hat_conv`, `hat_hab_tail`, ...) is captured with its arguments, every device pointer in them is resolved to (buffer,
offset) against the tensors the engine owns (packed weights, per-shape workspace) plus the input and the output, and the
list is written with the weights' bytes to a file that `hat_plan_load` / `hat_plan_forward` (csrc/hat_plan.cpp,
include/hat_mi355x.h "Forward plans") replay from C.  Whatever kernel sequence the engine chooses for the model and shape
is what the plan contains; results are bit-identical to `net(x)`.

File format (little endian): "HATPLAN1", u32 version = 1, u32 n_buffers, u32 n_calls, u32 n_functions, i32 dims[8] =
{B, Cin, H, W, scale, Cout, dtype, 0}; buffers {u32 kind (0 const, 1 scratch, 2 input, 3 output), u32 0, u64 bytes,
[bytes padded to 8 if const]}; calls {u32 function id, u32 n_args, args {u32 tag, u32 0, payload}} with payloads
int: i64 | float: f64 | pointer: u32 buffer (0xFFFFFFFF = NULL), u32 0, u64 offset | struct / host array: u32 bytes,
u32 n_fixups, image padded to 8, fixups {u32 field offset, u32 buffer, u64 offset} | stream: nothing.
"""
from __future__ import annotations

import argparse
import bisect
import ctypes as C
import os
import struct
import threading
from typing import List, Tuple

import torch

from . import _lib

# function ids = index in this list (csrc/hat_plan.cpp FN_NAMES mirrors it)
FN_IDS = ["hat_conv", "hat_linear", "hat_conv3x3_small", "hat_cab_fold", "hat_aggr_cab", "hat_ffn", "hat_ffn2", "hat_hab_tail",
          "hat_layernorm", "hat_esc_weights", "hat_eca_scale", "hat_dwconv_gate", "hat_sgfn_gate", "hat_ocab_attention",
          "hat_window_attention", "hat_cab_squeeze", "hat_conv3x3_to_planes", "hat_add_f32", "hat_esc_conv13", "hat_ocab_keybias", "hat_ocab_attention_kb", "hat_ocab_mlp", "hat_ocab_qkv", "hat_hab_tail3", "hat_ocab_attention_log2"]
ARG_INT, ARG_FLOAT, ARG_PTR, ARG_STRUCT, ARG_HOST, ARG_STREAM = range(6)
BUF_CONST, BUF_SCRATCH, BUF_INPUT, BUF_OUTPUT = range(4)
NULL_BUF = 0xFFFFFFFF
_EXPORT_LOCK = threading.Lock()


class _Recorder:
    """Stands in for the loaded library while one forward is recorded: forwards every call, keeps the launches."""

    def __init__(self, lib):
        self._lib, self.calls = lib, []
        self._tid = threading.get_ident()    # only the exporting thread's launches belong to the plan

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in FN_IDS:
            return fn      # queries (tile counts, plans): not launches

        def call(*args):
            if threading.get_ident() == self._tid:   # (another thread's forward passes through unrecorded)
                self.calls.append((name, [_snapshot(a) for a in args]))
            return fn(*args)
        return call


def _snapshot(a):
    """Copy an argument at call time (descriptors are reused / mutated by the caller afterwards)."""
    obj = getattr(a, "_obj", None)           # ctypes.byref(struct)
    if isinstance(obj, C.Structure):
        return ("struct", type(obj), bytes(memoryview(obj)))
    if isinstance(a, C.Array):
        return ("host", bytes(memoryview(a)))
    return a


def _pointer_fields(stype, base=0) -> List[int]:
    out = []
    for name, ftype in stype._fields_:
        off = base + getattr(stype, name).offset
        if isinstance(ftype, type) and issubclass(ftype, C.Structure):
            out += _pointer_fields(ftype, off)
        elif ftype is C.c_void_p:
            out.append(off)
    return out


def _tensors_of(obj, seen, out):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out.append(obj)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            _tensors_of(v, seen, out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tensors_of(v, seen, out)
    elif hasattr(obj, "__dict__") or hasattr(obj, "__slots__"):
        for k in list(getattr(obj, "__dict__", {})) + list(getattr(obj, "__slots__", ())):
            if k not in ("_lock", "_s1"):
                _tensors_of(getattr(obj, k, None), seen, out)


def export_plan(net, x_shape: Tuple[int, int, int, int], path: str) -> dict:
    """Record `net`'s forward for inputs of `x_shape` = (B, Cin, H, W) on the module's device and write the plan file."""
    eng = net.engine()
    dev = eng.dev
    if eng.pe_norm is None:
        raise NotImplementedError("patch_norm=False copies a tensor with a torch op, which a plan cannot record")
    B, Cin, H, W = x_shape
    x = torch.zeros(x_shape, dtype=torch.float32, device=dev)
    real = _lib.load()
    rec = _Recorder(real)
    prev_one = os.environ.get("HAT_ONE_STREAM")
    os.environ["HAT_ONE_STREAM"] = "1"          # the plan replays on one stream: record the launches in that order
    _EXPORT_LOCK.acquire()                     # one export at a time: the recorder stands in for the process-wide library handle
    _lib._lib = rec
    try:
        with torch.no_grad():
            y = eng.forward(x)
        torch.cuda.synchronize(dev)
    finally:
        _lib._lib = real
        _EXPORT_LOCK.release()
        if prev_one is None:
            os.environ.pop("HAT_ONE_STREAM", None)
        else:
            os.environ["HAT_ONE_STREAM"] = prev_one
    ws = eng._workspace(B, H, W)
    consts, scratch = [], []
    _tensors_of({k: v for k, v in eng.__dict__.items() if k != "_ws_cache"}, set(), consts)
    _tensors_of(ws, set(), scratch)

    # buffers = distinct storages; kind by where the tensor came from
    bufs, index = [], {}

    def add(t, kind):
        st = t.untyped_storage()
        key = st.data_ptr()
        if key in index:
            return
        index[key] = len(bufs)
        bufs.append({"kind": kind, "start": key, "nbytes": st.nbytes(), "storage": st, "tensor": t})

    add(x, BUF_INPUT)
    add(y, BUF_OUTPUT)
    for t in scratch:
        add(t, BUF_SCRATCH)
    for t in consts:
        add(t, BUF_CONST)
    starts = sorted((b["start"], i) for i, b in enumerate(bufs))
    keys = [s for s, _ in starts]

    def resolve(p):
        if not p:
            return NULL_BUF, 0
        k = bisect.bisect_right(keys, p) - 1
        if k >= 0:
            b = bufs[starts[k][1]]
            if p < b["start"] + max(b["nbytes"], 1):
                return starts[k][1], p - b["start"]
        raise RuntimeError(f"plan export: device pointer {p:#x} does not belong to the engine's weights, its workspace, the input "
                           f"or the output (a temporary allocated inside forward?)")

    out = bytearray()
    out += b"HATPLAN1" + struct.pack("<4I", _lib.ABI_VERSION, len(bufs), len(rec.calls), len(FN_IDS))
    out += struct.pack("<8i", B, Cin, H, W, eng.scale, y.shape[1], eng.dtype, 0)
    for b in bufs:
        out += struct.pack("<IIQ", b["kind"], 0, b["nbytes"])
        if b["kind"] == BUF_CONST:
            raw = torch.empty(b["nbytes"], dtype=torch.uint8, device="cpu")
            raw.copy_(torch.tensor([], dtype=torch.uint8, device=dev).set_(b["storage"], 0, (b["nbytes"],)))
            data = raw.numpy().tobytes()
            out += data + b"\0" * ((8 - len(data) % 8) % 8)
    for name, args in rec.calls:
        sig = _lib.SIGNATURES[name][1]
        out += struct.pack("<II", FN_IDS.index(name), len(args))
        for k, a in enumerate(args):
            last = k == len(args) - 1
            if isinstance(a, tuple) and a[0] == "struct":
                _, stype, image = a
                fixes = []
                for off in _pointer_fields(stype):
                    (p,) = struct.unpack_from("<Q", image, off)
                    fixes.append((off,) + resolve(p))
                out += struct.pack("<IIII", ARG_STRUCT, 0, len(image), len(fixes)) + image + b"\0" * ((8 - len(image) % 8) % 8)
                for off, bi, bo in fixes:
                    out += struct.pack("<IIQ", off, bi, bo)
            elif isinstance(a, tuple) and a[0] == "host":
                image = a[1]
                out += struct.pack("<IIII", ARG_HOST, 0, len(image), 0) + image + b"\0" * ((8 - len(image) % 8) % 8)
            elif last:                                   # every launch entry point ends with the stream
                out += struct.pack("<II", ARG_STREAM, 0)
            elif sig[k] is C.c_void_p:
                bi, bo = resolve(a)
                out += struct.pack("<IIIIQ", ARG_PTR, 0, bi, 0, bo)
            elif sig[k] is C.c_float:
                out += struct.pack("<IId", ARG_FLOAT, 0, float(a))
            else:
                out += struct.pack("<IIq", ARG_INT, 0, int(a))
    with open(path, "wb") as f:
        f.write(out)
    return {"path": path, "launches": len(rec.calls), "buffers": len(bufs), "file_bytes": len(out),
            "const_bytes": sum(b["nbytes"] for b in bufs if b["kind"] == BUF_CONST),
            "scratch_bytes": sum(b["nbytes"] for b in bufs if b["kind"] == BUF_SCRATCH)}


class Plan:
    """ctypes view of hat_plan_load / hat_plan_forward / hat_plan_free — what a C host does (used by the tests)."""

    def __init__(self, path: str):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.hat_plan_load(path.encode(), C.byref(self._h)), f"hat_plan_load({path})")
        dims = (C.c_int32 * 8)()
        n, nb = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.hat_plan_info(self._h, dims, C.byref(n), C.byref(nb)), "hat_plan_info")
        self.dims, self.launches, self.device_bytes = list(dims), n.value, nb.value

    def forward(self, x: torch.Tensor, y: torch.Tensor, stream: int = 0):
        _lib.check(self._lib.hat_plan_forward(self._h, x.data_ptr(), y.data_ptr(), stream), "hat_plan_forward")

    def close(self):
        if self._h:
            self._lib.hat_plan_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


def main(argv=None):
    ap = argparse.ArgumentParser(description="export a forward plan for hat_plan_load / hat_plan_forward")
    ap.add_argument("-opt", required=True, help="test YAML (network_g, path.pretrain_network_g ...), as for super_resolution_amd.test")
    ap.add_argument("--shape", type=int, nargs=3, metavar=("B", "H", "W"), required=True)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    from .models import HATModel
    from .test import parse_options
    opt = parse_options(args.opt)
    model = HATModel(opt, device=args.device)
    B, H, W = args.shape
    info = export_plan(model.net_g, (B, opt["network_g"].get("in_chans", 3), H, W), args.out)
    print(info)
    return info


if __name__ == "__main__":
    main()
