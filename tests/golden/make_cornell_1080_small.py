"""Generates tests/golden/reference_cornell_box_1080_down8.npy from the reference's own render of config C2
(/root/reference/cornell_box_1080x1080.png, 8-bit sRGB): decoded to linear, box-averaged over 8x8 pixel blocks
(135x135x3 float16).  Data only; run where /root/reference exists."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
q = mi.read_image("/root/reference/cornell_box_1080x1080.png")[..., :3].astype(np.float64)
lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
small = lin.reshape(135, 8, 135, 8, 3).mean((1, 3))
np.save(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_1080_down8.npy"), small.astype(np.float16))
print(small.shape, small.mean((0, 1)))
