import sys, time
sys.path.insert(0,'.')
import gmupt_pkg
pkg=gmupt_pkg.load()
mesh=pkg.scenes.spheres_mesh(1953,4,seed=1234)
time.sleep(2)
t=time.time(); b=pkg.capi.sbvh_build(mesh['verts'],mesh['indices'],mesh['vertex_material']); print('%.3f'%(time.time()-t), b['nodes'].shape)
