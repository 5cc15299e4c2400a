"""Offline study (CPU): regression of the measured per-matrix launch times (profiles/r04_general512_cubic_per_matrix.txt, tools/general_per_matrix.py)\non the bank model conflict factor and the box size.  Result: profiles/r04_block_lane_shapes.txt."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import voltools_amd as vt
n=512
rs=np.random.RandomState(1); rs.random_sample(n*n*n)
rots=rs.uniform(-180,180,(100,3))
mats=[np.asarray(vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n,n,n),2)),np.float64)[:3,:4].reshape(12) for r in rots]
rows=[l.split() for l in open('profiles/r04_general512_cubic_per_matrix.txt')]
assert all(abs(float(r[4])-rots[int(r[0])][0])<1e-2 for r in rows)
ms=np.array([float(r[1]) for r in rows]); lds=np.array([int(r[3]) for r in rows])
kBase=np.array([[8.13,8.27,8.41],[8.44,8.50,8.78],[8.71,8.09,8.33],[8.92,8.66,8.05]])
l=np.arange(32); T=np.stack([l>>4,(l>>2)&3,l&3],1).astype(np.float64)
rng=np.random.RandomState(5); many=rng.uniform(8,9,(64,3))
def conflicts(m,RS,PS,bases):
    M=m.reshape(3,4)[:, :3]; tot=0
    for b in bases:
        f=np.floor(b+24.0+T@M.T).astype(np.int64)
        a=f[:,0]*PS+f[:,1]*RS+(((f[:,2]-1)&~1)>>1)*2
        w=np.unique(a>>1); tot+=np.bincount(w%32,minlength=32).max()
    return tot/len(bases)
RSs=[12,20,28,36,16,24,32]
fs=[]
for m,ldsb in zip(mats,lds):
    M=np.abs(m.reshape(3,4)[:, :3]); Tt=np.array([8,8,16])-1
    L=[int(np.floor((M[r]*Tt).sum()))+3+2 for r in range(3)]
    lx=(L[2]+3+3)&~3
    best=None
    for rsv in RSs:
        if rsv<lx: continue
        for pad in range(0,64,4):
            ps=L[1]*rsv+pad
            if L[0]*ps//4>256*13: break
            f=conflicts(m,rsv,ps,kBase)*(1+0.002*pad)
            if best is None or f<best[0]-1e-9: best=(f,rsv,ps)
        if best: break
    box=((L[0]*best[2]//4+63)//64*64*16+16) if best else -1
    fs.append((conflicts(m,best[1],best[2],many) if best else np.nan, box))
fs=np.array(fs)
ok=(fs[:,1]==lds)
print('planner emulation matches LDS bytes for',ok.sum(),'of 100')
f=fs[ok,0]; t=ms[ok]; lb=lds[ok]
print('corr(model f, ms) = %.3f'%np.corrcoef(f,t)[0,1], ' corr(lds bytes, ms) = %.3f'%np.corrcoef(lb,t)[0,1])
A=np.stack([np.ones_like(f),f,lb/1024.0],1); coef,res,_,_=np.linalg.lstsq(A,t,rcond=None)
print('ms ~ %.3f + %.3f*f + %.4f*KB ; residual rms %.4f ; ms std %.4f'%(coef[0],coef[1],coef[2],np.sqrt(np.mean((A@coef-t)**2)),t.std()))
for lo,hi in ((1.0,1.3),(1.3,1.6),(1.6,2.0),(2.0,4.0)):
    s=(f>=lo)&(f<hi)
    if s.any(): print('f in [%.1f,%.1f): n=%d mean ms %.4f mean KB %.1f'%(lo,hi,s.sum(),t[s].mean(),lb[s].mean()/1024))
