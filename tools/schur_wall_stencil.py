"""CPU prototype (oracle only) of the wall stencil of the pressure block (pl_solver.hip, prec_p_value): BiCGStab with an exact velocity
block, diagonal pressure block against diagonal + the local stencil of the inverse wall block, on grids stretched either way."""
import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def run(nz, nx_, visc="iso"):
    nx=[nz,nx_]; L=1.0; grid=[np.linspace(0,L,nz), np.linspace(0,L,nx_)]
    Z,X=np.meshgrid(grid[0],grid[1],indexing="ij")
    if visc=="iso": eta=np.ones((nz,nx_))
    else: eta=10**(1.5*np.sin(3*np.pi*X)*np.cos(2*np.pi*Z))
    rho=1.0+0.1*np.exp(-((Z-0.4)**2+(X-0.55)**2)/0.02)
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].tocsr(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    lu=spl.splu(Avv)
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1).reshape(nz,nx_)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    dS=np.diag(App-0).copy()
    Sd=np.diag(App-Apv@lu.solve(Avp.toarray()))
    sgn=np.sign(np.median(Sd[cont.reshape(-1)]))
    xd=spl.spsolve(A.tocsc(),b)
    dz=grid[0][1]-grid[0][0]; dx=grid[1][1]-grid[1][0]; a=dx/dz
    def make(kind):
        def f(rp):
            R=rp.reshape(nz,nx_)
            Zp=np.where(cont, sgn*2.0*eta/Kc**2*R, np.where(Sd.reshape(nz,nx_)!=0, R/np.where(Sd.reshape(nz,nx_)!=0,Sd.reshape(nz,nx_),1), 0))
            if kind=="stencil":
                al=2.9; ga=0.45
                if a>=2:
                    bb=0.5*a*a
                    for (j0,j1) in ((0,1),(nx_-2,nx_-3)):
                        r0=np.where(cont[:,j0],R[:,j0],0.0); r1=np.where(cont[:,j1],R[:,j1],0.0)
                        d=r0-r1
                        d2=np.zeros(nz); d2[1:-1]=d[:-2]-2*d[1:-1]+d[2:]
                        d2[0]=d[1]-d[0]; d2[nz-2]=d[nz-3]-d[nz-2]
                        z0=sgn*(al*r0-ga*r1-bb*d2)*eta[:,j0]/Kc**2
                        z1=sgn*(0.5*r0+1.55*r1)*eta[:,j1]/Kc**2
                        Zp[:,j0]=np.where(cont[:,j0],z0,Zp[:,j0]); Zp[:,j1]=np.where(cont[:,j1],z1,Zp[:,j1])
                if 1.0/a>=2:
                    bb=0.5/(a*a)
                    for (i0,i1) in ((0,1),(nz-2,nz-3)):
                        r0=np.where(cont[i0,:],R[i0,:],0.0); r1=np.where(cont[i1,:],R[i1,:],0.0)
                        d=r0-r1
                        d2=np.zeros(nx_); d2[1:-1]=d[:-2]-2*d[1:-1]+d[2:]
                        d2[0]=d[1]-d[0]; d2[nx_-2]=d[nx_-3]-d[nx_-2]
                        z0=sgn*(al*r0-ga*r1-bb*d2)*eta[i0,:]/Kc**2
                        z1=sgn*(0.5*r0+1.55*r1)*eta[i1,:]/Kc**2
                        Zp[i0,:]=np.where(cont[i0,:],z0,Zp[i0,:]); Zp[i1,:]=np.where(cont[i1,:],z1,Zp[i1,:])
            return Zp.reshape(-1)
        return f
    def bicg(sinv, rtol=1e-8, maxit=300):
        def M(r):
            zp=sinv(r[ip]); zv=lu.solve(r[iv]-Avp@zp); z=np.empty_like(r); z[iv]=zv; z[ip]=zp; return z
        x=np.zeros_like(b); r=b.copy(); rt=np.random.default_rng(1).standard_normal(b.size)
        rho_=alpha=omega=1.0; v=np.zeros_like(b); p=np.zeros_like(b); bn=np.linalg.norm(b)
        for it in range(1,maxit+1):
            rn=rt@r; beta=(rn/rho_)*(alpha/omega); p=r+beta*(p-omega*v)
            y=M(p); v=A@y; alpha=rn/(rt@v); s=r-alpha*v
            z=M(s); t=A@z; omega=(t@s)/(t@t)
            x=x+alpha*y+omega*z; r=s-omega*t; rho_=rn
            if np.linalg.norm(r)/bn<rtol: break
        return it, np.linalg.norm(x[iv]-xd[iv])/np.linalg.norm(xd[iv])
    for kind in ("diag","stencil"):
        it,err=bicg(make(kind))
        print("%dx%d %-4s %-8s iterations %3d  velocity error %.1e" % (nz,nx_,visc,kind,it,err), flush=True)
for a in [(65,33),(97,33),(17,65),(33,129)]:
    run(*a,"iso"); run(*a,"var")
