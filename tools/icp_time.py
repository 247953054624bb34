import torch,sys,time
sys.path.insert(0,".")
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
s=make_sequence(3,480,640,seed=0)
K=torch.from_numpy(s["intrinsics"][0]).cuda()
pts=[]
for f in (0,2):
    d=torch.from_numpy(s["depths"][f,...,0]).cuda()
    v,n,_,_=ops.frame_maps(d,K)
    gv,gn=ops.global_maps(v,n,d,torch.from_numpy(s["poses"][0]).cuda())
    pts.append(ops.downsample_frame(gv,gn,None,d,4)[:2])
(tgt,tn),(src,_)=pts
def t(src,mode,it=20):
    ops.icp(src,tgt,tn,mode=mode,numiters=it,return_idx=False); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): ops.icp(src,tgt,tn,mode=mode,numiters=it,return_idx=False)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/20*1e6
for ns in (18273, 4096, 512):
    print("n_src",ns,"gradICP us/iter", (t(src[:ns],1)-t(src[:ns],1,1))/19, "ICP(mode0) us/iter", (t(src[:ns],0)-t(src[:ns],0,1))/19)
