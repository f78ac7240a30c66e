"""Generates tests/golden/reference_{glissoncapsule,parenchyma}_{cpu,gpu}_down8.npy from the reference's own renders of the
two layer scenes (/root/reference/scenes/GlissonCapsule/mitsuba3/outputs/Mitsuba3/{CPU,GPU}/glissoncapsule.png and
/root/reference/scenes/Parenchyma/mitsuba3/outputs/Mitsuba/{CPU,GPU}/parenchyma.png: 1920x1080, 8-bit sRGB), and
reference_liver_multimesh_down8.npy from /root/reference/scenes/Liver-MultiMesh/mitsuba3/liver-multimesh.png (the render of that
directory's scene_temp.xml: both meshes, both tissue media, envmap; 256 spp, 44.6 s in time.txt): decoded to linear, box-averaged
over 8x8 blocks (135x240x3 float16).  Data only.

The renders were made from earlier versions of the scene files than the committed ones: GlissonCapsule under a constant
white environment (the committed file has the envmap), Parenchyma under the envmap block the committed file keeps in a
comment, emitters visible.  tests/test_bio_oracle.py rebuilds those variants from the committed files."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
for name, sub, png in (("GlissonCapsule", "Mitsuba3", "glissoncapsule.png"), ("Parenchyma", "Mitsuba", "parenchyma.png")):
    for dev in ("CPU", "GPU"):
        q = mi.read_image(f"/root/reference/scenes/{name}/mitsuba3/outputs/{sub}/{dev}/{png}")[..., :3].astype(np.float64)
        lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
        small = lin.reshape(135, 8, 240, 8, 3).mean((1, 3))
        out = os.path.join(ROOT, "tests", "golden", f"reference_{name.lower()}_{dev.lower()}_down8.npy")
        np.save(out, small.astype(np.float16))
        print(out, small.shape, small.mean((0, 1)))
q = mi.read_image("/root/reference/scenes/Liver-MultiMesh/mitsuba3/liver-multimesh.png")[..., :3].astype(np.float64)
lin = np.where(q <= 0.04045, q / 12.92, ((q + 0.055) / 1.055) ** 2.4)
small = lin.reshape(135, 8, 240, 8, 3).mean((1, 3))
np.save(os.path.join(ROOT, "tests", "golden", "reference_liver_multimesh_down8.npy"), small.astype(np.float16))
print("liver-multimesh", small.shape, small.mean((0, 1)))
