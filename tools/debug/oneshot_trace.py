import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import voltools_amd as vt
n = 512
data = np.random.RandomState(1).random_sample((n, n, n)).astype(np.float32)
m = vt.utils.transform_matrix(rotation=(0, 33, 0), translation=(1.5, 2, -3), center=np.divide((n, n, n), 2))
for i in range(4):
    t0 = time.perf_counter(); r = vt.affine(data, m, interpolation='linear', device='gpu'); print('call', i, f'{(time.perf_counter()-t0)*1e3:.2f} ms', flush=True); del r
