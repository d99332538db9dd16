"""CPU study (oracle only): BiCGStab with an EXACT velocity block and the pressure block (a) diagonal, (b) diagonal + the true 2 x 2
blocks of S coupling the two cell columns next to each x-wall -- does a local repair of S^ at the walls pay?  (It does not: DESIGN.md 4.)"""
import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def run(nz, nx_, visc="iso"):
    nx=[nz,nx_]; grid=[np.linspace(0,1,nz), np.linspace(0,1,nx_)]
    Z,X=np.meshgrid(grid[0],grid[1],indexing="ij")
    eta=np.ones((nz,nx_)) if visc=="iso" else 10**(1.5*np.sin(3*np.pi*X)*np.cos(2*np.pi*Z))
    rho=1.0+0.1*np.exp(-((Z-0.4)**2+(X-0.55)**2)/0.02)
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].tocsr(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    lu=spl.splu(Avv)
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    S=App-Apv@lu.solve(Avp.toarray())
    dS=np.diag(S).copy()
    en=eta.reshape(-1)
    xd=spl.spsolve(A.tocsc(),b)
    idx=lambda i,j:i*nx_+j
    def make(kind):
        # returns function rp -> zp
        d=np.where(cont, 0.5*Kc**2/en*np.sign(np.where(dS!=0,dS,1)), np.where(dS!=0,dS,1.0))
        if kind=="diag":
            return lambda rp: rp/d
        # wall 2x2 blocks from the true S (columns 0,1 and nx-2, nx-3), everything else diagonal
        pairs=[]
        for i in range(nz-1):
            for (j0,j1) in ((0,1),(nx_-2,nx_-3)):
                a,b_=idx(i,j0),idx(i,j1)
                if cont[a] and cont[b_]:
                    pairs.append((a,b_,np.linalg.inv(np.array([[S[a,a],S[a,b_]],[S[b_,a],S[b_,b_]]]))))
        def f(rp):
            z=rp/d
            for a,b_,Bi in pairs:
                z[a],z[b_]=Bi@np.array([rp[a],rp[b_]])
            return z
        return f
    def bicg(sinv, rtol=1e-8, maxit=400):
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
    for kind in ("diag","wall2x2"):
        it,err=bicg(make(kind))
        print("%dx%d %s %-8s iterations %3d  velocity error %.1e" % (nz,nx_,visc,kind,it,err), flush=True)
for a in [(33,33),(65,17),(129,9)]:
    run(*a,"iso"); run(*a,"var")
