"""Generates tests/golden/reference_cornell_box_fog_1080_down8.npy from the reference tree's own volpath render of the
Cornell box in homogeneous fog (/root/reference/cornell_box_1080x1080_fog_st_albedo.png, 8-bit sRGB; recipe:
/root/reference/MitsubaRunner.py:8-40: sigma_t 0.2, albedo 0.75, scale 2.5, isotropic, medium on the sensor, volpath
max_depth -1, 1080x1080): decoded to linear, box-averaged over 8x8 pixel blocks (135x135x3 float16).  Data only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
q = mi.read_image("/root/reference/cornell_box_1080x1080_fog_st_albedo.png")[..., :3].astype(np.float64)
lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
small = lin.reshape(135, 8, 135, 8, 3).mean((1, 3))
np.save(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_fog_1080_down8.npy"), small.astype(np.float16))
print(small.shape, small.mean((0, 1)))
