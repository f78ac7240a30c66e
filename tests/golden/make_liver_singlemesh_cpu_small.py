"""Generates tests/golden/reference_liver_singlemesh_cpu_down8.npy from the reference's own scalar_rgb render of
Liver-SingleMesh with the fork's biovolpath integrator and liver medium
(/root/reference/scenes/Liver-SingleMesh/mitsuba3/outputs/Mitsuba3/CPU/liver-singlemesh.png: 1920x1080, 128 spp, 8-bit sRGB):
decoded to linear, box-averaged over 8x8 blocks (135x240x3 float16).  Data only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
q = mi.read_image("/root/reference/scenes/Liver-SingleMesh/mitsuba3/outputs/Mitsuba3/CPU/liver-singlemesh.png")[..., :3].astype(np.float64)
lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
small = lin.reshape(135, 8, 240, 8, 3).mean((1, 3))
np.save(os.path.join(ROOT, "tests", "golden", "reference_liver_singlemesh_cpu_down8.npy"), small.astype(np.float16))
print(small.shape, small.mean((0, 1)))

# scenes/Liver-SingleMesh/mitsuba3/scene.png: the render of scene.xml at the file's own defaults (854x480, 256 spp) that sits
# next to the scene file: box-averaged 4x2 -> 120x427x3 float16 (tests/test_bio_oracle.py)
q = mi.read_image("/root/reference/scenes/Liver-SingleMesh/mitsuba3/scene.png")[..., :3].astype(np.float64)
lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
small = lin.reshape(120, 4, 427, 2, 3).mean((1, 3))
np.save(os.path.join(ROOT, "tests", "golden", "reference_liver_singlemesh_scene_png_down.npy"), small.astype(np.float16))
print(small.shape, small.mean((0, 1)))
