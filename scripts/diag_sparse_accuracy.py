"""Accuracy of the sparse-GP fit against the QR-based oracle AND a dense evaluation when K_uu is ill-conditioned
(cond 1e7): what the CholeskyQR2 repair in agp_sparse_fit_create is for."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, albatross_amd as ab
from oracle import oracle_py as orc
ctx=ab.Context(0)
for n,gs,m,shuf in ((900,300,64,0),(900,300,64,1),(1200,150,100,0),(1200,150,100,1),(512,128,30,1),(900,300,20,1)):
    rng=np.random.default_rng(n+gs)
    x=np.sort(rng.uniform(0.,30.,n)); y=np.sin(x)+0.2*x+0.1*rng.standard_normal(n); yvar=rng.uniform(0.01,0.04,n)
    cov=ab.SquaredExponential(2.5,1.5)+ab.measurement_only(ab.IndependentNoise(0.2))
    u=np.linspace(0.,30.,m)
    rank={float(v):i for i,v in enumerate(x)}
    grouper=lambda f: rank[float(f)]//gs
    model=ab.sparse_gp_from_covariance(cov,grouper,ab.FixedInducingPoints(u),"s",context=ctx)
    model.set_param("inducing_nugget",1e-6)
    perm=rng.permutation(n) if shuf else np.arange(n)
    fm=model.fit(ab.RegressionDataset(x[perm],ab.MarginalDistribution(y[perm],yvar[perm])))
    keys=np.array([grouper(f) for f in x])
    o=orc.OracleSparseFit(cov,x,keys,y,yvar,u,1e-8,1e-6)
    v=o.information
    Kuu=orc.gram(cov,u)+1e-6*np.eye(m); Kfu=orc.gram(cov,x,u,x_meas=True); Kff=orc.gram(cov,x,x_meas=True)
    w_,V_=np.linalg.eigh(Kuu); Q=(Kfu@V_)/w_@(Kfu@V_).T
    K=Q.copy()
    for kk in np.unique(keys):
        idx=np.nonzero(keys==kk)[0]; K[np.ix_(idx,idx)]=Kff[np.ix_(idx,idx)]
    K+=np.diag(yvar)+1e-8*np.eye(n)
    dn=0.5*(np.linalg.slogdet(K)[1]+y@np.linalg.solve(K,y)+n*np.log(2*np.pi))
    print("   dense nll",dn,"hip-dense",fm.get_fit().nll-dn,"oracle-dense",o.nll-dn, "cond Kuu %.1e"%(w_[-1]/w_[0]))
    print(n,gs,m,shuf,"info rel err",np.abs(fm.get_fit().information-v).max()/np.abs(v).max(),"nll diff",fm.get_fit().nll-o.nll)
