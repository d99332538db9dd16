"""Algorithm prototype of the GPU Stokes solver (TEST/DESIGN INFRASTRUCTURE, not product).

Right-preconditioned BiCGStab (seeded random shadow residual) on the row-scaled reference
Stokes matrix with the block upper-triangular preconditioner
        M = [[A_vv, A_vp], [0, S^]],   S^ = diag(Kc^2/eta_n) on continuity rows,
where A_vv^-1 is one geometric-multigrid V-cycle on the staggered velocity block:
rediscretised coarse operators (arithmetic viscosity coarsening), Chebyshev-Jacobi smoothing,
constraint rows closed exactly; the finest level keeps the reference's slaved boundary rows,
coarse levels use natural mirror rows.  Arrays carry a one-node ring like the device planes,
so every function maps 1:1 onto a kernel of pylamp_amd/csrc/pl_solver.hip.  Tests use it to
check the HIP preconditioner component-wise; the product never imports it.
"""
import numpy as np
import scipy.sparse as sp

from . import pylamp_oracle as O


class Lv: pass

def tables(n, h):
    # index k+1 ; rd[k]=1/(x[k+1]-x[k]) for 0<=k<n-1 ; rD[k]=1/(x[k+1]-x[k-1]) for 1<=k<=n-2 ; else 0
    rd=np.zeros(n+3); rD=np.zeros(n+3)
    rd[1:n]=1.0/h
    rD[2:n]=1.0/(2*h)
    return rd,rD

def pad(a):
    return np.pad(a,1)

def setup(nz,nx,hz,hx,etas,etan,bc,quirk):
    L=Lv(); L.nz,L.nx,L.hz,L.hx,L.bc,L.quirk=nz,nx,hz,hx,bc,quirk
    L.es=pad(etas); L.en=pad(etan)
    L.rdz,L.rDz=tables(nz,hz); L.rdx,L.rDx=tables(nx,hx)
    mz=np.zeros((nz,nx)); mx=np.zeros((nz,nx))
    if quirk:
        mz[1:nz-1,1:nx-2]=1; mx[1:nz-2,1:nx-1]=1
    else:
        mz[1:nz-1,0:nx-1]=1
        mx[0:nz-1,1:nx-1]=1
        if bc[0]!=O.BC_FREESLIP: mx[0,:]=0
        if bc[2]!=O.BC_FREESLIP: mx[nz-2,:]=0
    L.mz=pad(mz); L.mx=pad(mx)
    # coefficient arrays for all rows (garbage where masked)
    i=np.arange(nz); j=np.arange(nx)
    rdz_i=L.rdz[i+1][:,None]; rdz_m=L.rdz[i][:,None]; rDz_i=L.rDz[i+1][:,None]; rDz_p=L.rDz[i+2][:,None]
    rdx_j=L.rdx[j+1][None,:]; rdx_m=L.rdx[j][None,:]; rDx_j=L.rDx[j+1][None,:]; rDx_p=L.rDx[j+2][None,:]
    C=slice(1,nz+1),slice(1,nx+1)
    def sh(a,di,dj): return a[1+di:nz+1+di,1+dj:nx+1+dj]
    es,en=L.es,L.en
    L.zN=4*sh(en,0,0)*rdz_i*rDz_i; L.zS=4*sh(en,-1,0)*rdz_m*rDz_i
    L.zE=2*sh(es,0,1)*rDx_p*rdx_j; L.zW=2*sh(es,0,0)*rDx_j*rdx_j
    L.zxE=2*sh(es,0,1)*rDz_i*rdx_j; L.zxW=2*sh(es,0,0)*rDz_i*rdx_j
    L.xE=4*sh(en,0,0)*rdx_j*rDx_j; L.xW=4*sh(en,0,-1)*rdx_m*rDx_j
    L.xN=2*sh(es,1,0)*rDz_p*rdz_i; L.xS=2*sh(es,0,0)*rDz_i*rdz_i
    L.xzN=2*sh(es,1,0)*rDx_j*rdz_i; L.xzS=2*sh(es,0,0)*rDx_j*rdz_i
    dz=-(L.zN+L.zS+L.zE+L.zW); dx=-(L.xE+L.xW+L.xN+L.xS)
    L.dz=np.pad(np.where(mz>0,dz,1.0),1,constant_values=1.0); L.dx=np.pad(np.where(mx>0,dx,1.0),1,constant_values=1.0)
    L.sh=sh
    return L

def apply(L,vz,vx):
    sh=L.sh; nz,nx=L.nz,L.nx
    yz=(L.zN*(sh(vz,1,0)-sh(vz,0,0))-L.zS*(sh(vz,0,0)-sh(vz,-1,0))+L.zE*(sh(vz,0,1)-sh(vz,0,0))-L.zW*(sh(vz,0,0)-sh(vz,0,-1))
        +L.zxE*(sh(vx,0,1)-sh(vx,-1,1))-L.zxW*(sh(vx,0,0)-sh(vx,-1,0)))
    yx=(L.xE*(sh(vx,0,1)-sh(vx,0,0))-L.xW*(sh(vx,0,0)-sh(vx,0,-1))+L.xN*(sh(vx,1,0)-sh(vx,0,0))-L.xS*(sh(vx,0,0)-sh(vx,-1,0))
        +L.xzN*(sh(vz,1,0)-sh(vz,1,-1))-L.xzS*(sh(vz,0,0)-sh(vz,0,-1)))
    return pad(yz)*L.mz, pad(yx)*L.mx

def close(L,vz,vx,gz=None,gx=None):
    """arrays padded (index+1). Enforce constraint rows; g = rhs/Kc on those rows (fine level only)."""
    nz,nx=L.nz,L.nx
    G=lambda g,s: 0.0 if g is None else g[s]
    R=slice(1,nz+1); Cc=slice(1,nx+1)
    if L.quirk:
        vz[R,nx]=G(gz,(R,nx)); vz[1,1:nx]=G(gz,(1,slice(1,nx))); vz[nz,1:nx]=G(gz,(nz,slice(1,nx)))
        I=slice(2,nz)
        vz[I,1]=vz[I,2]+G(gz,(I,1)); vz[I,nx-1]=vz[I,nx-2]+G(gz,(I,nx-1))
        vx[nz,Cc]=G(gx,(nz,Cc)); vx[1:nz,1]=G(gx,(slice(1,nz),1)); vx[1:nz,nx]=G(gx,(slice(1,nz),nx))
        J=slice(2,nx)
        if L.bc[0]==O.BC_FREESLIP: vx[1,J]=vx[2,J]+G(gx,(1,J))
        else: vx[1,J]=(vx[2,J]/(2*L.hz)-G(gx,(1,J)))/(1.5/L.hz)
        if L.bc[2]==O.BC_FREESLIP: vx[nz-1,J]=vx[nz-2,J]+G(gx,(nz-1,J))
        else: vx[nz-1,J]=(vx[nz-2,J]/(2*L.hz)+G(gx,(nz-1,J)))/(1.5/L.hz)
    else:
        vz[1,:]=0; vz[nz,:]=0            # no flow through z walls
        vz[:,nx]=vz[:,nx-1]; vz[:,0]=vz[:,1]      # mirror ghosts (x walls, free slip)
        vx[:,1]=0; vx[:,nx]=0
        J=slice(2,nx)
        if L.bc[0]==O.BC_FREESLIP: vx[0,:]=vx[1,:]
        else: vx[1,J]=vx[2,J]/3.0
        if L.bc[2]==O.BC_FREESLIP: vx[nz,:]=vx[nz-1,:]
        else: vx[nz-1,J]=vx[nz-2,J]/3.0; vx[nz,:]=0

def lmax_est(L,iters=15):
    rng=np.random.default_rng(0)
    vz=rng.standard_normal(L.mz.shape); vx=rng.standard_normal(L.mz.shape)
    lam=2.0
    for _ in range(iters):
        vz*=L.mz; vx*=L.mx
        n0=np.sqrt(np.sum(vz**2)+np.sum(vx**2))
        close(L,vz,vx)
        yz,yx=apply(L,vz,vx); yz/=L.dz; yx/=L.dx
        lam=np.sqrt(np.sum(yz**2)+np.sum(yx**2))/n0
        vz,vx=yz/lam,yx/lam
    return lam

def smooth(L,vz,vx,fz,fx,n,gz=None,gx=None,ratio=6.0):
    lmax=L.lmax; lmin=lmax/ratio
    theta=0.5*(lmax+lmin); delta=0.5*(lmax-lmin); sigma=theta/delta; rho_old=1/sigma
    close(L,vz,vx,gz,gx)
    yz,yx=apply(L,vz,vx)
    rz=(fz-yz)/L.dz*L.mz; rx=(fx-yx)/L.dx*L.mx
    dz=rz/theta; dx=rx/theta
    for k in range(n):
        vz+=dz; vx+=dx
        close(L,vz,vx,gz,gx)
        if k==n-1: break
        yz,yx=apply(L,vz,vx)
        rz=(fz-yz)/L.dz*L.mz; rx=(fx-yx)/L.dx*L.mx
        rho=1/(2*sigma-rho_old)
        dz=rho*rho_old*dz+2*rho/delta*rz; dx=rho*rho_old*dx+2*rho/delta*rx
        rho_old=rho

def restrict(L,Lc,rz,rx):
    """padded in/out. vz: vertex in z (1/4,1/2,1/4), cell-centred in x (1/8,3/8,3/8,1/8)."""
    nzc,nxc=Lc.nz,Lc.nx
    cz=np.zeros((nzc+2,nxc+2)); cx=np.zeros((nzc+2,nxc+2))
    # unpadded fine index fi -> padded fi+1.  Need fine indices -1..n (pad has them: index 0 and n+1) and n+1 -> extend
    rzp=np.pad(rz,((0,0),(0,2))); rxp=np.pad(rx,((0,2),(0,0)))
    I=np.arange(nzc); J=np.arange(nxc)
    # vz coarse (I,J): fine rows 2I-1,2I,2I+1 ; fine cols 2J-1,2J,2J+1,2J+2
    def gz_(di,dj): 
        ii=np.clip(2*I+di+1,0,rzp.shape[0]-1); jj=2*J+dj+1
        return rzp[np.ix_(ii,jj)]
    t=0
    for di,wi in ((-1,.25),(0,.5),(1,.25)):
        for dj,wj in ((-1,.125),(0,.375),(1,.375),(2,.125)):
            t=t+wi*wj*gz_(di,dj)
    cz[1:nzc+1,1:nxc+1]=t
    def gx_(di,dj):
        ii=2*I+di+1; jj=np.clip(2*J+dj+1,0,rxp.shape[1]-1)
        return rxp[np.ix_(ii,jj)]
    t=0
    for di,wi in ((-1,.125),(0,.375),(1,.375),(2,.125)):
        for dj,wj in ((-1,.25),(0,.5),(1,.25)):
            t=t+wi*wj*gx_(di,dj)
    cx[1:nzc+1,1:nxc+1]=t
    return cz*Lc.mz, cx*Lc.mx

def prolong(L,Lc,ez,ex):
    """coarse padded (closed: ghosts/mirrors valid) -> fine padded correction"""
    nz,nx=L.nz,L.nx; nzc,nxc=Lc.nz,Lc.nx
    fz=np.zeros((nz+2,nx+2)); fx=np.zeros((nz+2,nx+2))
    # vz: z vertex-linear, x cell-linear (3/4,1/4).  fine (i,j): 
    i=np.arange(nz); j=np.arange(nx)
    I0=i//2; I1=(i+1)//2                      # z: even -> I0=I1, odd -> average
    Jn=j//2; Jo=np.where(j%2==0,Jn-1,Jn+1)    # x: nearest coarse centre and the other one
    Jo=np.clip(Jo,-1,nxc)                      # ring indices allowed (-1 -> padded 0, nxc -> padded nxc+1)
    Jn=np.clip(Jn,0,nxc)
    a=0.5*(ez[np.ix_(I0+1,Jn+1)]+ez[np.ix_(I1+1,Jn+1)]); b=0.5*(ez[np.ix_(I0+1,Jo+1)]+ez[np.ix_(I1+1,Jo+1)])
    fz[1:nz+1,1:nx+1]=0.75*a+0.25*b
    J0=j//2; J1=(j+1)//2
    In=i//2; Io=np.where(i%2==0,In-1,In+1); Io=np.clip(Io,-1,nzc); In=np.clip(In,0,nzc)
    a=0.5*(ex[np.ix_(In+1,J0+1)]+ex[np.ix_(In+1,J1+1)]); b=0.5*(ex[np.ix_(Io+1,J0+1)]+ex[np.ix_(Io+1,J1+1)])
    fx[1:nz+1,1:nx+1]=0.75*a+0.25*b
    return fz*L.mz, fx*L.mx

def coarsen_visc(es,en,mode):
    f,g={"geom":(np.log,np.exp),"arith":(lambda a:a,lambda a:a),"harm":(lambda a:1/a,lambda a:1/a)}[mode]
    nz,nx=es.shape
    p=np.pad(f(es),1,mode='edge')
    w=(p[:-2,:-2]+p[:-2,2:]+p[2:,:-2]+p[2:,2:])/16+(p[:-2,1:-1]+p[2:,1:-1]+p[1:-1,:-2]+p[1:-1,2:])/8+p[1:-1,1:-1]/4
    esc=g(w[0::2,0::2])
    nzc,nxc=esc.shape
    a=f(en[:nz-1,:nx-1])
    enc=np.ones((nzc,nxc))
    enc[:nzc-1,:nxc-1]=g(0.25*(a[0::2,0::2]+a[1::2,0::2]+a[0::2,1::2]+a[1::2,1::2]))
    enc[nzc-1,:]=enc[nzc-2,:]; enc[:,nxc-1]=enc[:,nxc-2]
    return esc,enc

def hierarchy(nx,grid,etas,etan,bc,mode="geom",natural=True,min_cells=4):
    nz,nxx=nx
    hz=(grid[0][-1]-grid[0][0])/(nz-1); hx=(grid[1][-1]-grid[1][0])/(nxx-1)
    etan=np.array(etan,copy=True); bad=~np.isfinite(etan); etan[bad]=np.exp(np.mean(np.log(etan[~bad])))
    es,en=np.asarray(etas),etan
    Ls=[setup(nz,nxx,hz,hx,es,en,bc,True)]
    while True:
        L=Ls[-1]
        if (L.nz-1)%2 or (L.nx-1)%2 or (L.nz-1)//2<min_cells or (L.nx-1)//2<min_cells: break
        es,en=coarsen_visc(es,en,mode)
        Ls.append(setup((L.nz-1)//2+1,(L.nx-1)//2+1,2*L.hz,2*L.hx,es,en,bc,not natural))
    for L in Ls: L.lmax=1.1*lmax_est(L)
    return Ls

def vcycle(Ls,l,fz,fx,gz=None,gx=None,nu=(3,3),csweeps=12,damp=1.0):
    L=Ls[l]
    vz=np.zeros_like(fz); vx=np.zeros_like(fx)
    if l==len(Ls)-1:
        ratio=max(30.0,0.4*L.nz*L.nx)
        smooth(L,vz,vx,fz,fx,min(150,max(csweeps,int(np.sqrt(ratio)))),gz,gx,ratio=ratio); return vz,vx
    smooth(L,vz,vx,fz,fx,nu[0],gz,gx)
    yz,yx=apply(L,vz,vx)
    cz,cx=restrict(L,Ls[l+1],(fz-yz)*L.mz,(fx-yx)*L.mx)
    ez,ex=vcycle(Ls,l+1,cz,cx,None,None,nu,csweeps,damp)
    pz,px=prolong(L,Ls[l+1],ez,ex)
    vz+=damp*pz; vx+=damp*px
    smooth(L,vz,vx,fz,fx,nu[1],gz,gx)
    return vz,vx


# ---- outer solver ----------------------------------------------------------------------------
def split(x, nx):
    X = x.reshape(nx[0], nx[1], 3)
    return X[:, :, 0].copy(), X[:, :, 1].copy(), X[:, :, 2].copy()


def join(vz, vx, p):
    return np.stack([vz, vx, p], axis=2).reshape(-1)


class Precond:
    """z = M^-1 r for UNSCALED r in the reference DOF order."""

    def __init__(self, nx, grid, etas, etan, rho, bc, nu=(2, 2), mode="arith", lmax=None):
        self.nx = nx
        self.A, self.b = O.stokes_csr(nx, grid, etas, etan, rho, bc)
        self.Kc, self.Kb = O.stokes_scaling(grid, etas, etan)
        N = nx[0] * nx[1]
        iv = np.sort(np.concatenate([np.arange(N) * 3, np.arange(N) * 3 + 1])); ip = np.arange(N) * 3 + 2
        self.Avp = self.A[iv][:, ip].tocsr()
        self.nu = nu
        self.cls = O.stokes_row_class(nx)
        en = np.array(etan, copy=True); en[~np.isfinite(en)] = 1.0
        self.Sinv = np.where(self.cls[2] == 1, en / self.Kc**2, 1.0 / self.Kc)
        self.Ls = hierarchy(nx, grid, etas, etan, bc, mode=mode, natural=True)
        if lmax is not None:
            for L, lm in zip(self.Ls, lmax):
                L.lmax = lm
        self.napply = 0

    def apply(self, r):
        self.napply += 1
        nx = self.nx
        rz, rx, rp = split(r, nx)
        zp = self.Sinv * rp
        for i0 in (0, nx[0] - 2):
            zp[i0, 0] = zp[i0, 1] - rp[i0, 0] / self.Kb
            zp[i0, nx[1] - 2] = zp[i0, nx[1] - 3] - rp[i0, nx[1] - 2] / self.Kb
        rv = np.stack([rz, rx], axis=2).reshape(-1) - self.Avp @ zp.reshape(-1)
        R = rv.reshape(nx[0], nx[1], 2)
        L0 = self.Ls[0]
        fz = np.pad(R[:, :, 0], 1); fx = np.pad(R[:, :, 1], 1)
        gz = fz / self.Kc; gx = fx / self.Kc
        vz, vx = vcycle(self.Ls, 0, fz * L0.mz, fx * L0.mx, gz, gx, self.nu)
        return join(vz[1:-1, 1:-1], vx[1:-1, 1:-1], zp)


class Scaled:
    """Row-scaled system (D_r A, D_r b) with the preconditioner M^-1 D_r^-1."""

    def __init__(self, M, nx, grid):
        self.M = M
        d = np.abs(M.A.diagonal())
        cls = O.stokes_row_class(nx)
        hz = (grid[0][-1] - grid[0][0]) / (nx[0] - 1); hx = (grid[1][-1] - grid[1][0]) / (nx[1] - 1)
        dd = d.reshape(nx[0], nx[1], 3).copy()
        dd[:, :, 2][cls[2] == 1] = M.Kc * (1 / hz + 1 / hx)
        self.s = 1.0 / dd.reshape(-1)
        self.A = sp.diags(self.s) @ M.A
        self.b = self.s * M.b

    def apply(self, r):
        return self.M.apply(r / self.s)


def bicgstab(A, b, M, rtol=1e-10, maxit=200, seed=1234):
    n = b.size
    x = np.zeros(n); r = b.copy()
    rt = np.random.default_rng(seed).standard_normal(n)
    rho = alpha = omega = 1.0
    v = np.zeros(n); p = np.zeros(n)
    bn = np.linalg.norm(b)
    it = 0
    for it in range(1, maxit + 1):
        rho_new = rt @ r
        beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        y = M.apply(p); v = A @ y
        alpha = rho_new / (rt @ v)
        s = r - alpha * v
        z = M.apply(s); t = A @ z
        omega = (t @ s) / (t @ t)
        x = x + alpha * y + omega * z
        r = s - omega * t
        rho = rho_new
        if np.linalg.norm(r) / bn < rtol:
            break
    return x, it, np.linalg.norm(b - A @ x) / bn
