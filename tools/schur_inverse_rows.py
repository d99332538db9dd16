"""CPU study (oracle only): rows of the INVERSE of the wall block of the pressure Schur complement for aspect ratios 1, 4, 16 -- the
measurement behind the constants of the wall stencil."""
import sys, numpy as np, scipy.sparse.linalg as spl
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import pylamp_oracle as O
def rows(nz, nx_, depth=2):
    nx=[nz,nx_]; grid=[np.linspace(0,1,nz), np.linspace(0,1,nx_)]
    eta=np.ones((nz,nx_)); rho=np.ones((nz,nx_))
    A,b=O.stokes_csr(nx,grid,eta,eta,rho,[1,1,1,1]); A=A.tocsr(); N=nz*nx_
    iv=np.sort(np.concatenate([np.arange(N)*3,np.arange(N)*3+1])); ip=np.arange(N)*3+2
    Avv=A[iv][:,iv].tocsc(); Avp=A[iv][:,ip].toarray(); Apv=A[ip][:,iv].tocsr(); App=A[ip][:,ip].toarray()
    cls=O.stokes_row_class(nx); cont=(cls[2].reshape(-1)==1)
    Kc,Kb=O.stokes_scaling(grid,eta,eta)
    S=(App-Apv@spl.splu(Avv).solve(Avp))/Kc**2
    I,J=np.meshgrid(np.arange(nz),np.arange(nx_),indexing="ij")
    wall=(J<depth)
    W=np.where(wall.reshape(-1)&cont)[0]
    Si=np.linalg.inv(S[np.ix_(W,W)])
    pos={w:k for k,w in enumerate(W)}
    idx=lambda i,j:i*nx_+j
    i0=nz//2
    print("%dx%d: rows of inv(S_WW) (W = first %d columns), units eta/Kc^2; bulk value would be 2.0" % (nz,nx_,depth))
    for j in range(depth):
        r=Si[pos[idx(i0,j)]]
        ent=[(di,dj,r[pos[idx(i0+di,dj)]]) for dj in range(depth) for di in range(-4,5) if abs(r[pos[idx(i0+di,dj)]])>0.02]
        print("  row (i0,%d):"%j, " ".join("(%+d,j=%d) %.2f"%e for e in ent))
rows(65,17); rows(33,33); rows(129,9)
