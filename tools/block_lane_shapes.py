"""Offline study (CPU): the lane-block kernel bank model of vt_plan.hip (block_conflicts) over the 100 random rotations of the reference protocol --\nthe planner rule against a joint (row stride, plane padding) search, and every 32-lane block shape with its best strides.  Result: profiles/r04_block_lane_shapes.txt."""
import numpy as np, sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import voltools_amd as vt
n=512
rs=np.random.RandomState(1); rs.random_sample((2,2,2))
rs=np.random.RandomState(1)
_ = rs.random_sample((n,n,n)) if False else None
# reproduce: data drawn first (n^3 samples) then rotations; emulate by advancing the generator cheaply
rs=np.random.RandomState(1); rs.random_sample(n*n*n)
rots=rs.uniform(-180,180,(100,3))
mats=[np.asarray(vt.utils.transform_matrix(rotation=r, rotation_order='sxyz', center=np.divide((n,n,n),2)),np.float64)[:3,:4].reshape(12) for r in rots]
kBase=np.array([[8.13,8.27,8.41],[8.44,8.50,8.78],[8.71,8.09,8.33],[8.92,8.66,8.05]])
l=np.arange(32); T=np.stack([l>>4,(l>>2)&3,l&3],1).astype(np.float64)
def conflicts(m,RS,PS,bases=kBase):
    M=m.reshape(3,4)[:, :3]
    tot=0
    for b in bases:
        f=np.floor(b+24.0+T@M.T).astype(np.int64)
        a=f[:,0]*PS+f[:,1]*RS+(((f[:,2]-1)&~1)>>1)*2
        w=np.unique(a>>1)
        tot+=np.bincount(w%32,minlength=32).max()
    return tot/len(bases)
RSs=[12,20,28,36,16,24,32]
rng=np.random.RandomState(5); many=rng.uniform(8,9,(64,3))
def plan(m,mode):
    M=np.abs(m.reshape(3,4)[:, :3]); Tt=np.array([16,8,8])-1   # TD=16? tile 8x8x16 -> (d,h,w) = (16, 8, 8)?  use T = (8, 8, 16) per DESIGN: tiles of 8x8x16 voxels (d,h,w)
    Tt=np.array([8,8,16])-1
    L=[int(np.floor((M[r]*Tt).sum()))+3+2 for r in range(3)]
    lx=(L[2]+3+3)&~3
    cands=[]
    for rsv in RSs:
        if rsv<lx: continue
        for pad in range(0,64,4):
            ps=L[1]*rsv+pad
            vec=L[0]*ps//4
            if vec>256*13: break
            cands.append((rsv,ps,vec))
    if not cands: return None
    if mode=='cur':
        rsv=cands[0][0]
        best=min((c for c in cands if c[0]==rsv), key=lambda c: conflicts(m,c[0],c[1])*(1+0.002*(c[1]-L[1]*c[0])))
        return best
    # joint: all RS, boxes <= 40 KiB preferred
    def score(c):
        pen=1.0 if c[2]*16<=40*1024 else 1.15
        return conflicts(m,c[0],c[1])*pen*(1+0.002*(c[1]-L[1]*c[0]))
    return min(cands,key=score)
cur=[];jo=[]
for m in mats:
    a=plan(m,'cur'); b=plan(m,'joint')
    if a is None: continue
    cur.append((conflicts(m,a[0],a[1],many),a[2]*16)); jo.append((conflicts(m,b[0],b[1],many),b[2]*16))
cur=np.array(cur);jo=np.array(jo)
print('n',len(cur),'current: model conflicts (64 positions) mean %.3f, box KB mean %.1f, >40KB: %d'%(cur[:,0].mean(),cur[:,1].mean()/1024,(cur[:,1]>40*1024).sum()))
print('joint  : model conflicts mean %.3f, box KB mean %.1f, >40KB: %d'%(jo[:,0].mean(),jo[:,1].mean()/1024,(jo[:,1]>40*1024).sum()))

print('--- lane-block shapes (32 lanes of a ds_read_b64 group), best (RS, PS) per matrix by the model, scored on 64 positions')
import itertools
shapes=[(a,b,c) for a in (1,2,4,8,16,32) for b in (1,2,4,8,16,32) for c in (1,2,4,8,16,32) if a*b*c==32 and a<=16 and b<=8 and c<=8]
def Tof(shape):
    a,b,c=shape
    l=np.arange(32)
    return np.stack([l//(b*c),(l//c)%b,l%c],1).astype(np.float64)
res={}
for shp in shapes:
    Tt_=Tof(shp)
    def conf(m,RS,PS,bases):
        M=m.reshape(3,4)[:, :3]; tot=0
        for b in bases:
            f=np.floor(b+24.0+Tt_@M.T).astype(np.int64)
            a=f[:,0]*PS+f[:,1]*RS+(((f[:,2]-1)&~1)>>1)*2
            w=np.unique(a>>1); tot+=np.bincount(w%32,minlength=32).max()
        return tot/len(bases)
    vals=[]
    for m in mats:
        M=np.abs(m.reshape(3,4)[:, :3]); Tt=np.array([8,8,16])-1
        L=[int(np.floor((M[r]*Tt).sum()))+3+2 for r in range(3)]
        lx=(L[2]+3+3)&~3
        best=None
        for rsv in RSs:
            if rsv<lx: continue
            for pad in range(0,64,4):
                ps=L[1]*rsv+pad
                if L[0]*ps//4>256*13: break
                f=conf(m,rsv,ps,kBase)*(1.0 if L[0]*ps*4<=40*1024 else 1.15)
                if best is None or f<best[0]: best=(f,rsv,ps)
        vals.append(conf(m,best[1],best[2],many[:16]))
    res[shp]=np.array(vals)
    print(shp,'mean %.3f'%res[shp].mean())
allv=np.stack([res[s] for s in shapes])
print('best shape per matrix: mean %.3f'%allv.min(0).mean(), 'shapes chosen:', {shapes[i]:int((allv.argmin(0)==i).sum()) for i in range(len(shapes)) if (allv.argmin(0)==i).sum()})
