"""Generate tests/golden/vgg_stages.npz from the REAL reference module (this container only).

TEST INFRASTRUCTURE.  /root/reference/vgg_16_UIE.py is imported with inert stand-ins for cv2 / torchvision (any use
raises); its DifferentiableEnhancement.forward runs on CPU torch.  Seeded inputs and the module's outputs are stored as
small fixtures; tests/test_oracle_golden.py replays them against oracle.uwie_oracle.diff_enhance and the GPU tests
against uwie_diff_enhance_f32.  Only arrays travel.

Run:  python oracle/gen_golden_vgg.py   (torch 2.10 CPU; torch.pow float32 may differ by an ulp between CPU kernels)
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vgg_stages.npz")


def main():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms"):
        mod = gg._Inert(name)
        mod.__path__ = []
        sys.modules.setdefault(name, mod)
    gg.import_reference()
    sys.path.insert(0, gg.REF)
    import torch
    import vgg_16_UIE as V

    enh = V.DifferentiableEnhancement()
    rng = np.random.default_rng(20260101)
    out = {}
    cases = {
        "u8_2x3x24x31": (np.float32(rng.integers(0, 256, (2, 3, 24, 31))) / np.float32(255.0), True, True),
        "rand_3x3x17x40": (rng.random((3, 3, 17, 40), dtype=np.float32), True, True),
        "dark_1x3x33x21": (rng.random((1, 3, 33, 21), dtype=np.float32) * np.float32(0.3), True, False),
        "flat_1x3x8x8": (np.full((1, 3, 8, 8), 0.5, np.float32), False, True),
        "stretch_only_2x3x16x16": (rng.random((2, 3, 16, 16), dtype=np.float32), False, False),
    }
    for tag, (img, has_omega, has_gamma) in cases.items():
        B = img.shape[0]
        par = {"L_low": rng.uniform(1, 30, (B, 1)).astype(np.float32), "L_high": rng.uniform(65, 99, (B, 1)).astype(np.float32)}
        if has_omega:
            par["omega"] = rng.uniform(0.1, 0.9, (B, 1)).astype(np.float32)
        if has_gamma:
            par["gamma"] = rng.uniform(0.5, 3.0, (B, 1)).astype(np.float32)
        with torch.no_grad():
            res = enh(torch.from_numpy(img), {k: torch.from_numpy(v) for k, v in par.items()}).numpy()
        out[f"{tag}/img"] = img
        for k, v in par.items():
            out[f"{tag}/{k}"] = v
        out[f"{tag}/out"] = res
    # extract_all_features (vgg_16_UIE.py:435-466) on uint8 frames: sizes around the 8192-element buffer boundaries
    for tag, shape in (("feat_37x53", (37, 53)), ("feat_64x128", (64, 128)), ("feat_91x90", (91, 90)), ("feat_5x7", (5, 7))):
        u8 = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        if tag == "feat_91x90":
            u8[:, :, 2] //= 5
        out[f"{tag}/u8"] = u8
        out[f"{tag}/features"] = V.extract_all_features(u8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
