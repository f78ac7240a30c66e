import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import liverrenderer_amd as mi
sc = mi.load_file(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/scenes/Liver-SingleMesh/mitsuba3/scene.xml', integrator='volpath', spp=512, res_width=1920, res_height=1080)
p = mi.traverse(sc); p['LiverMedium.phase_function.g'] = 0.7; p.update()
sc.render(seed=9)
t=time.perf_counter(); sc.render(seed=1); dt=time.perf_counter()-t
print('HG g=0.7', round(1920*1080*512/dt/1e6,1), 'Msamples/s', sc.stats())
