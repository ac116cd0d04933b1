"""`python -m super_resolution_amd.test -opt options/test/HAT-S_SRx4.yml` — the reference's `hat/test.py` entry
(basicsr `test_pipeline`): parse the YAML, build the test datasets and `HATModel`, validate each dataset."""
from __future__ import annotations

import argparse
import json
import sys

import yaml

from .data import FolderDataset
from .models import HATModel


def parse_options(path: str) -> dict:
    with open(path) as f:
        opt = yaml.safe_load(f)
    opt["is_train"] = False
    for phase, d in (opt.get("datasets") or {}).items():
        d["phase"] = phase.split("_")[0]
        if "scale" in opt:
            d["scale"] = opt["scale"]
    return opt


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    opt = parse_options(args.opt)
    model = HATModel(opt, device=args.device)
    results = {}
    for _, dopt in sorted((opt.get("datasets") or {}).items()):
        ds = FolderDataset(dopt)
        print(f"Testing {dopt['name']} ({len(ds)} images)...", file=sys.stderr)
        mean, rows = model.nondist_validation(ds, save_img=(opt.get("val") or {}).get("save_img", True))
        results[dopt["name"]] = {"mean": mean, "images": rows}
        print(f"Validation {dopt['name']}: " + "  ".join(f"# {k}: {v:.4f}" for k, v in mean.items()), file=sys.stderr)
    print(json.dumps(results))
    return results


if __name__ == "__main__":
    main()
