"""CPU study (oracle only): spectrum of S^-1 S and BiCGStab iterations with an EXACT velocity block for the falling block (contrast 1 and 1e3)
at 65^2 -- how much of the GPU's 30 iterations belongs to the pressure block (16 with the exact velocity block, 11 isoviscous)."""
import sys, numpy as np, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def run(n, contrast):
    nx=[n,n]; L=660e3; grid=[np.linspace(0,L,n), np.linspace(0,L,n)]
    Z,X=np.meshgrid(grid[0],grid[1],indexing="ij")
    # node and centre viscosities of a block (sharp): centre grid = midpoints
    inb=lambda Z,X: (np.abs(Z-0.38*L)<0.076*L)&(np.abs(X-0.5*L)<0.076*L)
    etas=np.where(inb(Z,X),1e19*contrast,1e19)
    zm=np.r_[0.5*(grid[0][1:]+grid[0][:-1]),grid[0][-1]]; xm=np.r_[0.5*(grid[1][1:]+grid[1][:-1]),grid[1][-1]]
    Zm,Xm=np.meshgrid(zm,xm,indexing="ij")
    etan=np.where(inb(Zm,Xm),1e19*contrast,1e19)
    rho=np.where(inb(Z,X),3350.0,3300.0)
    A,b=O.stokes_csr(nx,grid,etas,etan,rho,[1,1,1,1]); A=A.tocsr(); N=n*n
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].tocsr(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    lu=spl.splu(Avv)
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,etas,etan)
    S=App-Apv@lu.solve(Avp.toarray())
    Sd=np.diag(S).copy(); sgn=np.sign(np.median(Sd[cont]))
    en=etan.reshape(-1)
    xd=spl.spsolve(A.tocsc(),b)
    d=np.where(cont, sgn*0.5*Kc**2/en, np.where(Sd!=0,Sd,1.0))
    ev=np.sort(np.abs(np.linalg.eigvals(S[np.ix_(np.where(cont)[0],np.where(cont)[0])]/d[cont][:,None])))
    print("n=%d contrast %g: eig(S^-1 S): min %.2e 1%% %.2e 5%% %.2e median %.2f max %.2f" % (n,contrast,ev[0],ev[int(0.01*ev.size)],ev[int(0.05*ev.size)],np.median(ev),ev[-1]))
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
    it,err=bicg(lambda rp: rp/d)
    print("   exact velocity block + diagonal pressure block: %d iterations, velocity error %.1e" % (it,err), flush=True)
for c in (1.0, 1e3):
    run(65,c)
