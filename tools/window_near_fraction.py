"""How many level-0 samples a per-tile LDS window would serve (CPU only): for the bench's model-like encoder inputs,
the share of level-0 samples of level-0 queries whose live corners fall inside a tile's window, per tile / margin.
Used to size the LDS-window forward sketched in DESIGN.md section 7."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from uninext_amd import workloads
x = workloads.make_inputs("encoder", "model", batch=1, seed=100, device="cpu")
shapes = x["shapes"].tolist(); H0, W0 = shapes[0]
loc = x["loc"][0].numpy()            # [Lq, M, L, P, 2]
S0 = H0 * W0
q = np.arange(S0); qy, qx = q // W0, q % W0
l0 = loc[:S0, :, 0]                  # [S0, M, P, 2] level-0 samples of level-0 queries
px = l0[..., 0] * W0 - 0.5; py = l0[..., 1] * H0 - 0.5
x0 = np.floor(px); y0 = np.floor(py)
for TW, TH in ((16, 8), (8, 8)):
    tx0 = (qx // TW) * TW; ty0 = (qy // TH) * TH
    for ml, mr in ((3, 4), (4, 5), (5, 6), (6, 7)):
        lo_x = np.maximum(tx0 - ml, 0)[:, None, None]; hi_x = np.minimum(tx0 + TW - 1 + mr, W0 - 1)[:, None, None]
        lo_y = np.maximum(ty0 - ml, 0)[:, None, None]; hi_y = np.minimum(ty0 + TH - 1 + mr, H0 - 1)[:, None, None]
        inr = (py > -1) & (px > -1) & (py < H0) & (px < W0)
        # live corners must be inside: corners x0, x0+1 (clipped to image) 
        cx0 = np.clip(x0, 0, W0 - 1); cx1 = np.clip(x0 + 1, 0, W0 - 1); cy0 = np.clip(y0, 0, H0 - 1); cy1 = np.clip(y0 + 1, 0, H0 - 1)
        near = inr & (cx0 >= lo_x) & (cx1 <= hi_x) & (cy0 >= lo_y) & (cy1 <= hi_y)
        ww, wh = TW + ml + mr, TH + ml + mr
        print("tile %2dx%d margin %d/%d window %2dx%2d = %5.1f KB: near %.1f %% of level-0 samples of level-0 queries (in range %.1f %%)" % (
            TW, TH, ml, mr, ww, wh, ww * wh * 128 / 1024, 100 * near.mean(), 100 * inr.mean()))
# offsets distribution
dx = px - qx[:, None, None]; dy = py - qy[:, None, None]
print("level-0 offset |dx| percentiles 50/90/99:", np.percentile(np.abs(dx), [50, 90, 99]).round(2), " |dy|:", np.percentile(np.abs(dy), [50, 90, 99]).round(2))
