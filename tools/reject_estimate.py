# How many (wave cube, frame) pairs of the integrate role update NO voxel, and
# how large a cube's pixel footprint is (CPU estimate on the bench scene).
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from open3d_amd import synthetic as syn
import _oracle as orc
W,H=640,480
K=syn.intrinsics(W,H)
VOX=0.008; RES=16; TRUNC=8*VOX; DS=1000.0; DMAX=3.0
tot_cubes=0; dead_cubes=0; tot_vox=0; ok_vox=0
foot=[]; tiles_spans=[]
dead_by_z=0
for k in (0, 200, 400, 650, 900):
    d,c,_,T=syn.render_frames(k,1,W,H,device='cpu')
    d=d[0].numpy(); T=T[0].numpy() if hasattr(T[0],'numpy') else np.asarray(T[0])
    keys=orc.depth_touch(d, K, T, RES, VOX, TRUNC, DS, DMAX, 4)
    keys=np.asarray(keys)
    # voxel coordinates of a block
    g=np.arange(RES)
    zz,yy,xx=np.meshgrid(g,g,g,indexing='ij')
    loc=np.stack([xx,yy,zz],-1).reshape(-1,3)  # x fastest
    cube_id=(loc[:,0]//8)+2*(loc[:,1]//4)+8*(loc[:,2]//4)
    df=d.astype(np.float32)/np.float32(DS)
    fx,fy,cx,cy=K[0,0],K[1,1],K[0,2],K[1,2]
    # tile max depth map (8x8)
    dm=df.copy(); dm[(dm<=0)|(dm>DMAX)]=0
    tmax=dm.reshape(H//8,8,W//8,8).max(axis=(1,3))
    for key in keys[::3]:
        p=(key[None,:]*RES+loc)*VOX
        pc=p@T[:3,:3].T+T[:3,3]
        z=pc[:,2]
        u=fx*pc[:,0]/z+cx; v=fy*pc[:,1]/z+cy
        inb=(z>0)&(u>=0)&(u<=W-1)&(v>=0)&(v<=H-1)
        ui=np.clip(u,0,W-1).astype(int); vi=np.clip(v,0,H-1).astype(int)
        dd=np.where(inb, df[vi,ui], 0)
        ok=inb&(dd>0)&(dd<=DMAX)&((dd-z)>=-TRUNC)
        tot_vox+=ok.size; ok_vox+=ok.sum()
        for cidx in range(32):
            m=cube_id==cidx
            tot_cubes+=1
            if not ok[m].any():
                dead_cubes+=1
            # conservative tile test: footprint bbox tiles' max depth
            if inb[m].any():
                uu=u[m][inb[m]]; vv=v[m][inb[m]]
                t0=int(uu.min())//8; t1=int(uu.max())//8; s0=int(vv.min())//8; s1=int(vv.max())//8
                tiles_spans.append((t1-t0+1)*(s1-s0+1))
                foot.append((uu.max()-uu.min()+1)*(vv.max()-vv.min()+1))
                zlo=z[m].min()
                if zlo > tmax[s0:s1+1,t0:t1+1].max()+TRUNC and not ok[m].any():
                    dead_by_z+=1
            else:
                dead_by_z+=1
print('voxels ok frac', ok_vox/tot_vox)
print('cubes dead frac', dead_cubes/tot_cubes, 'detectable by tile-max test', dead_by_z/tot_cubes)
print('footprint px mean', np.mean(foot), 'p90', np.percentile(foot,90), 'tiles mean', np.mean(tiles_spans), 'p90', np.percentile(tiles_spans,90))
