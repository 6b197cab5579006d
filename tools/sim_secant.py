#!/usr/bin/env python3
"""sim_secant.py -- how many Sturm evaluations per eigenvalue the safeguarded secant rounds of csrc/tridiag.hip::bisect3_kernel need
against plain bisection, in an exact-arithmetic model: the spectrum of a BASELINE configs[3] channel (tests/golden/c4_4096.npz) stands
in for the tridiagonal matrix (count(x) = #{lambda < x}, log2|p_n(x)| = sum log2|lambda - x|).  The kernel's rules: first-level grid of
1024 points (count and log2|p_n| at each), bisection while the bracket holds several eigenvalues or has not halved in three rounds,
regula falsi in its Illinois form otherwise, the bisection's stopping rule.  `lf precision` rounds log2|p_n| (the kernel keeps it in
single precision: 2^-11 at its size).  usage: python tools/sim_secant.py [channel ...]      (a minute per channel and variant)"""
import numpy as np, sys, os
g=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "c4_4096.npz")); E=g["E"]
eps=2.220446049250313e-16
def simulate(lam, NG=1024, method='ill', lfprec=None):
    n=len(lam); lam=np.sort(lam); nrm=max(abs(lam[0]),abs(lam[-1])); lam=lam/nrm
    gl=lam[0]-1e-3; gu=lam[-1]+1e-3; w=gu-gl
    grid=gl+w*(np.arange(NG)+1)/(NG+1)
    cntg=np.searchsorted(lam,grid,side='left')
    evals=np.zeros(n,int); res=np.zeros(n)
    count=lambda x: np.searchsorted(lam,x,side='left')
    def logf(x):
        v=np.sum(np.log2(np.abs(lam-x)+1e-300))
        if lfprec: v=np.round(v/lfprec)*lfprec
        return v
    lfg=np.array([logf(x) for x in grid])
    final=lambda lo,hi: (0.5*(lo+hi)<=lo) or (0.5*(lo+hi)>=hi) or (hi-lo<=2*eps*max(abs(lo),abs(hi))+1e-300)
    for m in range(n):
        R=np.searchsorted(cntg, m, side='right'); L=R-1
        lo = gl if L<0 else grid[L]; hi = gu if R>=NG else grid[R]
        clo = 0 if L<0 else cntg[L]; chi = n if R>=NG else cntg[R]
        flo = None if L<0 else lfg[L]; fhi = None if R>=NG else lfg[R]
        ne=1; last=0; hist=[hi-lo]
        while not final(lo,hi) and ne<200:
            wd=hi-lo
            slow = len(hist)>=4 and hist[-1] > 0.5*hist[-4]
            if method=='bisect' or chi-clo>1 or flo is None or fhi is None or slow:
                x=0.5*(lo+hi); kind='b'
            else:
                dl=min(max(flo-fhi,-1000),1000)
                r=2.0**dl; t=r/(1+r)
                x=lo+wd*t; kind='s'
                tiny=2*eps*max(abs(lo),abs(hi))
                x=min(max(x,lo+tiny),hi-tiny)
            c=count(x); ne+=1; fx=logf(x)
            if c>m:
                hi=x; chi=c; fhi=fx
                if kind=='s' and last==+1 and flo is not None: flo-=1.0   # Illinois: the retained end's value halved
                last=+1
            else:
                lo=x; clo=c; flo=fx
                if kind=='s' and last==-1 and fhi is not None: fhi-=1.0
                last=-1
            if kind=='b': last=0
            hist.append(hi-lo)
        evals[m]=ne; res[m]=0.5*(lo+hi)
    return evals,res,lam
for l in ([int(a) for a in sys.argv[1:]] or [0]):
    for meth, prec in (('bisect', None), ('ill', None), ('ill', 2.0 ** -11)):
        ev,res,ls=simulate(E[l],method=meth,lfprec=prec)
        err=np.max(np.abs(res-ls)/np.maximum(np.abs(ls),1e-300))
        groups=ev.reshape(-1,1024)
        print('l',l,meth,'lf precision',prec,'evals mean %.1f median %d p90 %d p99 %d max %d | per WG median'%(ev.mean(),np.median(ev),np.percentile(ev,90),np.percentile(ev,99),ev.max()), np.median(groups,axis=1), 'p90', np.percentile(groups,90,axis=1), 'max',groups.max(axis=1),'relerr %.1e'%err)
