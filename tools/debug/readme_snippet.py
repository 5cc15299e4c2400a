import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, voltools_amd as vt
vol = np.random.rand(512, 512, 512).astype(np.float32)
sv = vt.StaticVolume(vol, interpolation='filt_bspline', device='gpu')
out = vt.empty(vol.shape, device='gpu')
for i in range(180):
    sv.rotate((0, i, 0), output=out)
tilt = sv.project(rotation=(30, 0, 0), rotation_order='sxyz')
stack = vt.StaticVolume(vol[:64, :64, :64], device='gpu').affine_batch(np.stack([vt.utils.rotation_matrix((a, b, 0))
                                                                       for a in range(0, 360, 30) for b in range(0, 180, 30)]))
rot = vt.transform(vol, rotation=(10, 20, 30), device='gpu')
print(tilt.shape, stack.shape, rot.shape, float(tilt.mean()))
