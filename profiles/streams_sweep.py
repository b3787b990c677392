"""Whole-job rate of the headline workload with the batch split into sub-batches on 1 .. 4 internal streams (tuning `streams`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import underwater_image_enhancement_amd as uw
from underwater_image_enhancement_amd import _lib
import bench
dev = uw.get_device(0)
B, H, W = 64, 2160, 3840
fr = bench.synth_frames("underwater", B, H, W, dev.torch_device, 0)
for rep in range(2):
    for n in (1, 2, 3, 4):
        with dev.tuning(streams=n):
            ms = bench.timed_enhance(dev, _lib, torch, fr, 2, 8)
        print(f"streams={n}: {ms:.3f} ms per step, {B * H * W / 1e3 / ms:.0f} MP/s")
