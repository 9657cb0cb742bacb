"""HIP-graph replay of the fused OIL loop at small batches (VERDICT r3 next #3 asked for it): the whole zedo_oil_run call
captured with torch.cuda.CUDAGraph on a side stream and replayed, against the direct call.  Result (profiles/graph_experiment_r04.txt):
identical time - the GPU is 96 % inside kernels at these sizes, the host is not in the way - so the product does not capture graphs."""
import os, sys, time, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"zedo-release_amd"))
import zedo_hip as zh
from lib.dataset import synthetic as syn
from zedo_hip.pipeline import linspace_f32
W=zh.Weights(syn.make_weights(seed=0))
for N,S in ((886,1000),(64,100),(2048,300)):
    d=syn.make_poses(N,seed=3); cl=syn.make_clusters(1,seed=3)
    dev=lambda a: torch.tensor(np.ascontiguousarray(a),dtype=torch.float32,device="cuda")
    geom=zh.reproj_prepare(dev(d["db_2d"][:,:,:2]),dev(d["camera_param"]),dev(d["db_2d"][:,:,2]))
    sched=zh.Schedule(W,linspace_f32(0.1,0.01,S))
    x0=dev(np.broadcast_to((cl-cl[:,0:1])[0][None],(N,17,3)).copy()); T0=dev(d["db_3d"][:,0,:])
    zh.workspace(N)
    def run(x,T): zh.oil_run(W,sched,x,geom,T,0,S,S//5)
    x,T=x0.clone(),T0.clone(); run(x,T); torch.cuda.synchronize()
    ts=[]
    for _ in range(5):
        x,T=x0.clone(),T0.clone(); torch.cuda.synchronize(); t0=time.perf_counter(); run(x,T); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    ref=x.clone()
    gx,gT=x0.clone(),T0.clone()
    g=torch.cuda.CUDAGraph()
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(gx,gT)   # warm on side stream
    torch.cuda.synchronize()
    t0=time.perf_counter()
    with torch.cuda.graph(g, stream=s):
        run(gx,gT)
    tcap=time.perf_counter()-t0
    tg=[]
    for _ in range(5):
        gx.copy_(x0); gT.copy_(T0); torch.cuda.synchronize(); t0=time.perf_counter(); g.replay(); torch.cuda.synchronize(); tg.append(time.perf_counter()-t0)
    print(f"rows {N} steps {S}: direct {min(ts)*1e3:.2f} ms ({min(ts)/S*1e6:.1f} us/step), graph replay {min(tg)*1e3:.2f} ms ({min(tg)/S*1e6:.1f} us/step), capture+instantiate {tcap*1e3:.0f} ms, bitwise equal {bool(torch.equal(gx,ref))}")
