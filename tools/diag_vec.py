import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from imageanalysis_amd import ba_solver
from test_ba_solver_gpu import _problem, BA_CASES
g,opt,prob=_problem(BA_CASES[0])
rng=np.random.default_rng(0)
n=1000
a=rng.normal(size=n); b=rng.normal(size=n); c=rng.normal(size=n); d=rng.normal(size=n)
A,B,C,D=[torch.from_numpy(v).cuda() for v in (a,b,c,d)]
out=torch.zeros(n,dtype=torch.float64,device='cuda')
prob.mul2(n,A,B,out); print('mul2', np.abs(out.cpu().numpy()-a*b).max())
prob.mul2(n,A,B,out,C,D); print('mul2cd', np.abs(out.cpu().numpy()-(a*b+c*d)).max())
Y=B.clone(); prob.axpby(n,2.0,A,-3.0,Y); print('axpby', np.abs(Y.cpu().numpy()-(2*a-3*b)).max())
Y=B.clone(); prob.axpby(n,1.0,A,0.0,Y); print('axpby b0', np.abs(Y.cpu().numpy()-a).max())
Y=B.clone(); prob.axpby(n,0.0,Y,0.5,Y); print('axpby alias', np.abs(Y.cpu().numpy()-0.5*b).max())
print('dot', prob.dot(A,B,n,False)-a@b)
h,hb,x,v=[torch.from_numpy(t.copy()).cuda() for t in (a,b,c,d)]
prob.lsmr_update(n,h,hb,x,v,0.3,-0.7,1.1)
hb2=a+0.3*b; x2=c-0.7*hb2; h2=d+1.1*a
print('lsmr_update', np.abs(hb.cpu().numpy()-hb2).max(), np.abs(x.cpu().numpy()-x2).max(), np.abs(h.cpu().numpy()-h2).max())
# lsmr few iterations vs CPU fake with downloaded J
from scipy.sparse import diags, vstack
from scipy.sparse.linalg import lsmr
args=(opt.n_cameras,opt.n_points,opt.by_camera_point_indices,opt.by_camera_points_2d)
J=opt.jac(g['x0'],*args)
prob.set_x(g['x0']); prob.residual_jac()
dd=1.0/np.maximum(np.sqrt(np.asarray(J.power(2).sum(axis=0)).ravel()),1e-9); dr=rng.uniform(0.01,0.1,prob.n)
Aop=vstack([J@diags(dd),diags(dr)]).tocsr(); bb=np.concatenate([prob.r.cpu().numpy()[:prob.m],np.zeros(prob.n)])
for k in (1,2,3,5):
    ref=lsmr(Aop,bb,atol=0,btol=0,conlim=0,maxiter=k)
    x,*_=ba_solver.lsmr_device(prob,torch.from_numpy(dd).cuda(),torch.from_numpy(dr).cuda(),atol=0,btol=0,conlim=0,maxiter=k)
    print('lsmr k',k, np.abs(x-ref[0]).max()/np.abs(ref[0]).max())
print('cond-ish: colnorm min/max', dd.min(), dd.max())
