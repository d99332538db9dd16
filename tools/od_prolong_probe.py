"""CPU study (NumPy prototype of the preconditioner, oracle/proto_stokes_solver.py): viscosity-weighted (operator-dependent, 1-D
series-resistance weights) prolongation with the rediscretised coarse operators, on the 1e3 falling block at 129^2.  It does not help:
55 -> 65 iterations with V(2,2), 44 -> 47 with V(3,3)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import proto_stokes_solver as P, pylamp_oracle as O

std_prolong = P.prolong

def od_prolong(L, Lc, ez, ex):
    """operator-dependent: 1-D series-resistance weights from the FINE level's viscosities (padded arrays L.es, L.en)"""
    nz,nx=L.nz,L.nx; nzc,nxc=Lc.nz,Lc.nx
    es=L.es[1:-1,1:-1]; en=L.en[1:-1,1:-1]             # unpadded fine viscosities: es at nodes (nz,nx), en at centres (nz,nx) (last row/col ghost)
    fz=np.zeros((nz+2,nx+2)); fx=np.zeros((nz+2,nx+2))
    i=np.arange(nz); j=np.arange(nx)
    def g(a,ii,jj): return a[np.ix_(np.clip(ii,0,nz-1),np.clip(jj,0,nx-1))]
    # ---- vz: z vertex-centred, x cell-centred
    I0=i//2; I1=(i+1)//2
    # z weights for odd rows: conductance of the fine cells above (i-1) and below (i): etan[i-1,j], etan[i,j]
    ku=g(en,i-1,j); kd=g(en,i,j)
    wz0=np.where((i%2==1)[:,None], ku/(ku+kd), 1.0); wz1=1.0-wz0          # weight of coarse row I0 resp. I1
    Jn=j//2; Jo=np.where(j%2==0,Jn-1,Jn+1); Jo=np.clip(Jo,-1,nxc); Jn=np.clip(Jn,0,nxc)
    # x: path to nearest coarse centre: half an interval through the node between fine cols (2Jn, 2Jn+1): node index 2Jn+1
    # path to the other: one interval through node toward it + half an interval through the next node
    jn_node=2*(j//2)+1
    jo_node1=np.where(j%2==0, j, j+1)          # node between fine col j and its neighbour on the Jo side: even j -> neighbour j-1, node j ; odd j -> neighbour j+1, node j+1
    jo_node2=np.where(j%2==0, j-1, j+2)        # next node further on
    k1=g(es,i,jn_node); k2=g(es,i,jo_node1); k3=g(es,i,jo_node2)
    Rn=0.5/k1; Ro=1.0/k2+0.5/k3
    wxn=(1/Rn)/(1/Rn+1/Ro); wxo=1.0-wxn
    e=lambda A,II,JJ: A[np.ix_(II+1,JJ+1)]
    fz[1:nz+1,1:nx+1]=wxn*(wz0*e(ez,I0,Jn)+wz1*e(ez,I1,Jn))+wxo*(wz0*e(ez,I0,Jo)+wz1*e(ez,I1,Jo))
    # ---- vx: x vertex-centred, z cell-centred
    J0=j//2; J1=(j+1)//2
    kl=g(en,i,j-1); kr=g(en,i,j)
    wx0=np.where((j%2==1)[None,:], kl/(kl+kr), 1.0); wx1=1.0-wx0
    In=i//2; Io=np.where(i%2==0,In-1,In+1); Io=np.clip(Io,-1,nzc); In=np.clip(In,0,nzc)
    in_node=2*(i//2)+1; io_node1=np.where(i%2==0,i,i+1); io_node2=np.where(i%2==0,i-1,i+2)
    k1=g(es,in_node,j); k2=g(es,io_node1,j); k3=g(es,io_node2,j)
    Rn=0.5/k1; Ro=1.0/k2+0.5/k3
    wzn=(1/Rn)/(1/Rn+1/Ro); wzo=1.0-wzn
    fx[1:nz+1,1:nx+1]=wzn*(wx0*e(ex,In,J0)+wx1*e(ex,In,J1))+wzo*(wx0*e(ex,Io,J0)+wx1*e(ex,Io,J1))
    return fz*L.mz, fx*L.mx

def problem(n, contrast=1e3):
    nx=[n,n]; Lx=660e3; grid=[np.linspace(0,Lx,n), np.linspace(0,Lx,n)]
    rng=np.random.default_rng(5); nt=n*n*16
    tr=rng.random((nt,2))*Lx
    rho=np.full(nt,3300.0); eta=np.full(nt,1e19)
    b=(tr[:,0]>200e3)&(tr[:,0]<300e3)&(tr[:,1]>280e3)&(tr[:,1]<380e3)
    rho[b]=3350; eta[b]=1e19*contrast
    f=np.stack([rho,eta],axis=1)
    frho,fes=O.trac2grid(tr,f,grid,nx,[5,6])
    fen,=O.trac2grid(tr,f[:,1:2],O.gridmp_of(grid),nx,[6])
    return nx,grid,fes,fen,frho

for n in (129,):
    nx,grid,es,en,rho=problem(n)
    for name,pro in (("standard",std_prolong),("operator-dependent",od_prolong)):
        P.prolong=pro
        for nu in ((2,2),(3,3)):
            M=P.Precond(nx,grid,es,en,rho,[1,1,1,1],nu=nu,mode="arith")
            Sc=P.Scaled(M,nx,grid)
            x,it,res=P.bicgstab(Sc.A,Sc.b,Sc,rtol=1e-8,maxit=150)
            print("n=%d %-18s nu=%s: %d iterations (true residual %.1e)" % (n,name,nu,it,res),flush=True)
