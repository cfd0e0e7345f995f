#!/usr/bin/env python
"""One layer of the static mask head through the exact-fp32 MFMA convolution (conv3x3_hip_packed_exact_f32), for counters / A-B:
    python tools/conv_exact_probe.py [layer] [reps]      layer: jia_dcn (256->256 @100x167), lay4, lay1, lay3, lay2"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uninext_amd import ext  # noqa: E402

LAYERS = {"lay3": (256, 256, 25, 42), "lay4": (256, 256, 50, 84), "jia_dcn": (256, 256, 100, 167), "lay1": (256, 64, 100, 167),
          "lay2": (64, 8, 100, 167)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "jia_dcn"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    cin, cout, H, W = LAYERS[name]
    dev = torch.device("cuda:0")
    x = torch.randn(2, cin, H, W, device=dev)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    pe = ext.conv3x3_pack_weight(conv.weight.detach(), exact=True)
    with torch.no_grad():
        for _ in range(5):
            y = ext.conv3x3_packed_forward(x, pe, cout, conv.bias, relu=True, exact=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            y = ext.conv3x3_packed_forward(x, pe, cout, conv.bias, relu=True, exact=True)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            r = torch.relu_(conv(x))
        t1.record()
        torch.cuda.synchronize()
        ref = torch.relu(torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1))
        err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    flop = 2.0 * 2 * H * W * cout * cin * 9
    print("%s: exact fp32 MFMA %.1f us = %.1f TFLOP/s (%.1f %% of 157.3), err %.1e | torch conv + relu %.1f us" % (
        name, us, flop / us * 1e-6, 100 * flop / us * 1e-6 / 157.3, err, t0.elapsed_time(t1) / reps * 1e3))


if __name__ == "__main__":
    main()
