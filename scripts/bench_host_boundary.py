"""C3 through the host-buffer form of the boundary (lrt_render writing the developed image and the raw film into host
memory): the PCIe-inclusive rate DESIGN.md quotes next to bench.py's device-resident figure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liverrenderer_amd as mi
sc = mi.load_file(os.path.join(ROOT, "scenes/Liver-SingleMesh/mitsuba3/scene.xml"), integrator="volpath", spp=512, res_width=1920, res_height=1080)
sc.render(seed=9, return_raw=True)
t = time.perf_counter()
for s in range(3): sc.render(seed=s, return_raw=True)
dt = (time.perf_counter() - t) / 3
print(f"host-buffer lrt_render (image + raw film copied to host): {dt * 1e3:.1f} ms/render, {1920 * 1080 * 512 / dt / 1e6:.1f} Msamples/s; kernel {sc.stats()['kernel_ms']:.1f} ms")
