"""LDS bank-conflict model of the general-rotation gather: 32 lanes of a half-wave, 32 banks of 4 bytes, conflict degree =
max number of distinct addresses on one bank (equal addresses broadcast).  Compares lane->voxel mappings and LDS strides
for random rotations; the `line32` dense case reproduces the measured 61-63 % conflict cycles (degree 2.7).  CPU only."""
import numpy as np
rs=np.random.RandomState(0)
def rand_rot():
    q=rs.normal(size=4); q/=np.linalg.norm(q)
    w,x,y,z=q
    return np.array([[1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w)],[2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w)],[2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)]])
def conflict(addrs, nb=32):
    # addrs: (32,) ints; degree = max over banks of number of distinct addresses
    worst=1
    banks={}
    for a in set(addrs.tolist()):
        banks[a%nb]=banks.get(a%nb,0)+1
    return max(banks.values())
def lanes_line(T):   # half-wave = 32 lanes along w
    return [np.array([0,0,k]) for k in range(32)]
def lanes_block(shape):   # half wave = block shape (d,h,w) with d*h*w=32
    d,h,w=shape
    return [np.array([i,j,k]) for i in range(d) for j in range(h) for k in range(w)]
def run(lanes, Lx, Lp, cubic, nrot=200):
    tot=0;cnt=0
    for _ in range(nrot):
        A=rand_rot()
        base=rs.uniform(20,21,size=3)+np.array([20,20,20])
        pos=np.array([base+A@l for l in lanes])   # (32,3) source coords (z,y,x)
        fl=np.floor(pos).astype(int)
        taps = range(-1,3) if cubic else range(0,2)
        for dz in taps:
            for dy in taps:
                for dx in taps:
                    a=(fl[:,0]+dz)*Lp+(fl[:,1]+dy)*Lx+fl[:,2]+dx
                    tot+=conflict(a); cnt+=1
    return tot/cnt
for cubic in (False,True):
    print('cubic' if cubic else 'linear')
    print('  line32, Lx=44 Lp=44*28 :', run(lanes_line(None),44,44*28,cubic))
    print('  line32, Lx=33 Lp=33*29+? :', run(lanes_line(None),33,33*29,cubic))
    best=None
    for Lx in range(32,48):
        for Lpm in range(0,32,1):
            Lp=Lx*28+Lpm
            v=run(lanes_line(None),Lx,Lp,cubic,nrot=30)
            if best is None or v<best[0]: best=(v,Lx,Lpm)
    print('  line32 best static strides', best)
    for shape,Lx,Lp in (((2,4,4),36,36*28+16-((36*28)%32)),((2,4,4),37,37*28+ (25-(37*28)%32)%32),((4,4,2),34,34*28+(8-(34*28)%32)%32),((2,2,8),40,40*28+(16-(40*28)%32)%32),((1,4,8),40,40*28)):
        print('  block',shape,'Lx',Lx,'Lp%32',Lp%32,':', run(lanes_block(shape),Lx,Lp,cubic))
    best=None
    for Lx in range(32,48):
        for Lpm in range(0,32,2):
            Lp=Lx*28+Lpm
            v=run(lanes_block((2,4,4)),Lx,Lp,cubic,nrot=30)
            if best is None or v<best[0]: best=(v,Lx,Lp%32)
    print('  block(2,4,4) best static strides', best)

print("per-matrix optimised strides")
def run_one(A, lanes, Lx, Lp, cubic, nbase=3):
    tot=0;cnt=0
    for b in range(nbase):
        base=np.array([40.13+0.31*b,40.27+0.23*b,40.41+0.37*b])
        pos=np.array([base+A@l for l in lanes]); fl=np.floor(pos).astype(int)
        taps = range(-1,3) if cubic else range(0,2)
        for dz in taps:
            for dy in taps:
                for dx in taps:
                    a=(fl[:,0]+dz)*Lp+(fl[:,1]+dy)*Lx+fl[:,2]+dx
                    tot+=conflict(a); cnt+=1
    return tot/cnt
for name,lanes in (('line32',lanes_line(None)),('block244',lanes_block((2,4,4))),('block442',lanes_block((4,4,2))),('block128? (1,2,16)',lanes_block((1,2,16))),('block(2,2,8)',lanes_block((2,2,8)))):
    res=[];res0=[]
    for _ in range(40):
        A=rand_rot()
        best=1e9
        for Lx in range(32,48,1):      # row stride candidates (floats); dense would be ~28-44
            for Lpm in range(0,32,2):
                v=run_one(A,lanes,Lx,Lx*28+Lpm,False,nbase=2)
                best=min(best,v)
        res.append(best); res0.append(run_one(A,lanes,44,44*28,False,nbase=2))
    print(f'  {name}: dense {np.mean(res0):.2f}  per-matrix best {np.mean(res):.2f}  (linear taps)')
