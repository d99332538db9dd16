"""CPU study (oracle only): BiCGStab with an exact velocity block and the pressure block diagonal except on the cells within `depth`
columns of the x-walls (or of all walls), where the EXACT dense block of the Schur complement is used: how much of the iteration count
belongs to the wall rows of S."""
import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def run(nz, nx_, visc="iso", depth=2):
    nx=[nz,nx_]; L=660e3; grid=[np.linspace(0,L,nz), np.linspace(0,L,nx_)]
    Z,X=np.meshgrid(grid[0],grid[1],indexing="ij")
    if visc=="iso": eta=np.full((nz,nx_),1e21); rho=3300+50*np.exp(-((Z/L-0.4)**2+(X/L-0.55)**2)/0.02)
    else:
        T=273+1350*Z/L+60*np.sin(3*np.pi*X/L)*np.sin(np.pi*Z/L)
        eta=np.clip(1e20*np.exp(120e3/(8.31446*T)-120e3/(8.31446*1623)),1e19,1e23); rho=3300*(1-3.5e-5*(T-1623))
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].tocsr(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    lu=spl.splu(Avv)
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    S=App-Apv@lu.solve(Avp.toarray())
    dS=np.diag(S).copy(); en=eta.reshape(-1)
    xd=spl.spsolve(A.tocsc(),b)
    I,J=np.meshgrid(np.arange(nz),np.arange(nx_),indexing="ij")
    def make(kind):
        d=np.where(cont, 0.5*Kc**2/en*np.sign(np.where(dS!=0,dS,1)), np.where(dS!=0,dS,1.0))
        if kind=="diag": return lambda rp: rp/d
        if kind=="xwalls": wall=((J<depth)|(J>=nx_-1-depth))
        elif kind=="allwalls": wall=((J<depth)|(J>=nx_-1-depth)|(I<depth)|(I>=nz-1-depth))
        W=np.where(wall.reshape(-1)&cont)[0]
        SWW=np.linalg.inv(S[np.ix_(W,W)])
        def f(rp):
            z=rp/d; z[W]=SWW@rp[W]; return z
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
    for kind in ("diag","xwalls","allwalls"):
        it,err=bicg(make(kind))
        print("%dx%d %-6s %-8s depth %d iterations %3d  velocity error %.1e" % (nz,nx_,visc,kind,depth,it,err), flush=True)
for n in (33,65):
    run(n,n,"iso"); run(n,n,"mantle")
run(65,17,"iso"); run(65,17,"mantle")
