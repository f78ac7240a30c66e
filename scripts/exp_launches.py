import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LRT_DEBUG_LAUNCH"] = "1"
import liverrenderer_amd as mi
sc = mi.load_file(os.path.join(ROOT, "scenes/Liver-SingleMesh/mitsuba3/scene.xml"), integrator="volpath", spp=64, res_width=1920, res_height=1080)
sc.render(spp=4)
print("---- spp 64", flush=True)
sc.render()
