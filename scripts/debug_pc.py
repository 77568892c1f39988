import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swirl_fem_amd.examples import navier_stokes_driver as drv
from swirl_fem_amd.navier_stokes import navier_stokes as ns
from swirl_fem_amd.navier_stokes import pressure_preconditioner as pc
from swirl_fem_amd.linalg.cg import cg
DEV='cuda:0'
n=int(os.environ.get('N','4')); order=int(os.environ.get('ORDER','5'))
sem,u,p,d = drv.taylor_green(n=n, order=order, reynolds=100.0, dt=2e-3, steps=1, time_order=3, device=DEV, tol=1e-9)
M = pc.make_pressure_preconditioner(sem,'schwarz',2e-3,3)
E = ns._PressureOperator(sem,2e-3,3)
g = torch.Generator(device=DEV).manual_seed(9)
npr=p.numel()
r=torch.zeros(npr,dtype=p.dtype,device=DEV); el=M.pel[5]
r[el]=torch.randn(el.numel(),dtype=p.dtype,device=DEV,generator=g); r[el]-=r[el].mean()
z=M.local_solve(r); Ez=E(z)
print('local exact err', float((Ez[el]-Ez[el].mean()-r[el]).abs().max()/r.abs().max()))
yc=torch.randn(M.pel.shape[0],dtype=p.dtype,device=DEV,generator=g)
fine=torch.zeros(npr,dtype=p.dtype,device=DEV); fine[M.pel.reshape(-1)]=yc[:,None].expand(-1,M.pel.shape[1]).reshape(-1)
want=E(fine)[M.pel].sum(dim=1); got=M.coarse_matvec(yc)
print("coarse op err", float((got-want).abs().max()/want.abs().max()), "singular", M.coarse_singular, "iters", M.coarse_iterations, "bounds", M.coarse_bounds)
yc=yc-yc.mean(); bc=M.coarse_matvec(yc); xc=M._coarse_solve(bc); print('coarse solve err', float((xc-yc).abs().max()/yc.abs().max()))
b=E(torch.randn(npr,dtype=p.dtype,device=DEV,generator=g))
P0=ns._NullspaceProjection(sem)
for name,Mx in (('projection',P0),('schwarz',M),('local only',lambda r: P0(M.local_solve(r))),
                ('coarse only+identity', lambda r: P0(r + (M._coarse_solve(r[M.pel].sum(1)))[:,None].expand(-1,M.pel.shape[1]).reshape(-1)[torch.argsort(torch.argsort(M.pel.reshape(-1)))] ))):
  try:
    x,info=cg(E,b,M=Mx,tol=1e-8,maxiter=2000)
    print(name, info['num_iterations'], info['status'])
  except Exception as e:
    print(name,'failed',repr(e)[:200])
# eigen-range of E restricted per element vs FDM
print('lam range', float(M.inv_ev[M.inv_ev>0].min()), float(M.inv_ev.max()))
