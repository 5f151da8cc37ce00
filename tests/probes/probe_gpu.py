# first GPU parity probe: HIP vs oracle
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from dynode_amd import ModelDesc
from dynode_amd.engine import solve_batch
def cfg3(B, seed=1, A=8, S=4):
    rng=np.random.default_rng(seed)
    w=rng.dirichlet(5*np.ones(A)); M=rng.uniform(0.05,1,(A,A)); C=(M+M.T)/2+2*np.eye(A); C/=np.max(np.real(np.linalg.eigvals(C)))
    r0=rng.uniform(1.8,2.8,(B,S)); Ti=rng.uniform(5,9,(B,S)); Tl=rng.uniform(2,4,(B,S)); Tw=rng.uniform(50,90,(B,S))
    params=np.concatenate([r0/Ti,1/Ti,1/Tl,1/Tw],1)
    D=A*(1+4*S); y0=np.zeros((B,D)); y0[:,:A]=990*w
    dom=r0/r0.sum(1,keepdims=True)
    y0[:,A+A*S:A+2*A*S]=(10*w[None,:,None]*dom[:,None,:]).reshape(B,-1)
    return y0,params,C
B=int(sys.argv[1]) if len(sys.argv)>1 else 512
y0,params,C=cfg3(B)
ts=np.linspace(0,365,366)
mo=O.Model(n_age=8,n_strain=4,has_e=True,has_wane=True,has_c=True)
md=ModelDesc(n_age=8,n_strain=4,has_e=True,has_wane=True,has_c=True)
for dt_np, dt_t, tol in ((np.float64, torch.float64, 1e-9),(np.float32, torch.float32, 1e-5)):
    ys_o,st_o,na_o,nr_o=O.solve(mo,y0,params,C,365,ts,dtype=dt_np,n_threads=16)
    r=solve_batch(md,y0,params,C,365.0,ts,dtype=dt_t)
    torch.cuda.synchronize()
    ys=r.ys.cpu().numpy(); st=r.status.cpu().numpy(); na=r.n_accept.cpu().numpy(); nr=r.n_reject.cpu().numpy()
    err=np.abs(ys-ys_o).max(axis=(1,2))/1000.0
    print(dt_np.__name__, "status", st.max(), "max scaled err", err.max(), "median", np.median(err),
          "steps equal frac", np.mean((na==na_o)&(nr==nr_o)), "acc mean", na.mean(), "rej mean", nr.mean(), flush=True)
    assert err.max() < tol*10, err.max()
# timing
Bb=16384
y0,params,C=cfg3(Bb)
out=torch.empty((Bb,366,136),dtype=torch.float32,device="cuda")
y0t=torch.tensor(y0,dtype=torch.float32,device="cuda"); pt=torch.tensor(params,dtype=torch.float32,device="cuda"); Ct=torch.tensor(C,dtype=torch.float32,device="cuda"); tst=torch.tensor(ts,dtype=torch.float32,device="cuda")
for i in range(3): r=solve_batch(md,y0t,pt,Ct,365.0,tst,out=out)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): r=solve_batch(md,y0t,pt,Ct,365.0,tst,out=out)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/10
print("B",Bb,"ms/launch",ms,"traj/s",Bb/ms*1e3,"GB/s out",Bb*366*136*4/ms/1e6, "status max", int(r.status.max()))
