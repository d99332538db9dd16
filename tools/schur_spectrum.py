"""CPU study (oracle only): spectrum and rows of the exact pressure Schur complement S = A_pp - A_pv A_vv^-1 A_vp of the reference's
Stokes matrix on isoviscous grids with square and stretched cells, in units of the library's S^ = Kc^2 / eta."""
import sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def spec(nz, nx_):
    nx=[nz,nx_]; grid=[np.linspace(0,1,nz), np.linspace(0,1,nx_)]
    eta=np.ones((nz,nx_)); rho=np.ones((nz,nx_))
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].toarray(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    X=spl.splu(Avv).solve(Avp)
    S=App-Apv@X                      # Schur complement on the pressure unknowns
    ic=np.where(cont)[0]
    Sc=S[np.ix_(ic,ic)]
    # the library's S^: continuity rows  +- Kc^2/eta
    ev=np.linalg.eigvals(Sc/(Kc**2))
    ev=np.sort(np.abs(ev))
    print("%dx%d: |eig(S)|/Kc^2 on the %d continuity rows: min %.3e  5%% %.3e  median %.3e  max %.3e ; count < 0.1: %d, < 0.01: %d" % (nz,nx_,ic.size,ev[0],ev[int(0.05*ev.size)],np.median(ev),ev[-1],(ev<0.1).sum(),(ev<0.01).sum()))
for a in [(33,33),(65,17),(129,9),(129,17),(17,65)]:
    spec(*a)

def modes(nz, nx_, k=4):
    nx=[nz,nx_]; grid=[np.linspace(0,1,nz), np.linspace(0,1,nx_)]
    eta=np.ones((nz,nx_)); rho=np.ones((nz,nx_))
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].toarray(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    S=App-Apv@spl.splu(Avv).solve(Avp)
    ic=np.where(cont)[0]
    Sc=S[np.ix_(ic,ic)]/Kc**2
    w,V=np.linalg.eig(Sc); o=np.argsort(np.abs(w))
    for q in list(o[1:1+k])+[o[40], o[80]]:
        full=np.zeros(N); full[ic]=np.real(V[:,q]); F=full.reshape(nz,nx_)**2
        rows=F.sum(1); cols=F.sum(0)
        print("eig %.3e: energy by row (first 4 | middle | last 4): %s | %.2e | %s ; by column (first 3 | last 3): %s | %s" % (abs(w[q]), np.round(rows[:4]/F.sum(),3), rows[4:-5].sum()/F.sum(), np.round(rows[-5:-1]/F.sum(),3), np.round(cols[:3]/F.sum(),3), np.round(cols[-4:-1]/F.sum(),3)))
print("modes 65x17 (cells 4x wider than high)")
modes(65,17)

def wallcol(nz, nx_):
    nx=[nz,nx_]; grid=[np.linspace(0,1,nz), np.linspace(0,1,nx_)]
    eta=np.ones((nz,nx_)); rho=np.ones((nz,nx_))
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].toarray(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    S=(App-Apv@spl.splu(Avv).solve(Avp))/Kc**2
    idx=lambda i,j: i*nx_+j
    i0=nz//2
    print("%dx%d  S rows (units Kc^2/eta) around cell (%d, j):" % (nz,nx_,i0))
    for j in (0,1,2,nx_//2):
        r=S[idx(i0,j)]
        ent=[(di,dj,r[idx(i0+di,j+dj)]) for di in (-3,-2,-1,0,1,2,3) for dj in (-1,0,1,2) if 0<=j+dj<nx_-1 and abs(r[idx(i0+di,j+dj)])>2e-3]
        print("  j=%d:"%j, " ".join("(%+d,%+d) %.3f"%e for e in ent))
wallcol(65,17)
wallcol(33,33)
