import sys, time, numpy as np
sys.path.insert(0,'.')
import gml_amd as gml
from oracle import oracle as O
s=np.loadtxt('tests/golden/mvt_samples.csv',delimiter=',')
counts,spins=O.split_histogram(s)
n=spins.shape[1]
rng=np.random.default_rng(0)
with gml.Problem(s) as p:
    nodes=np.arange(n); th=rng.normal(scale=0.3,size=(n,n))
    for form in ['RISE','logRISE','RPLE']:
        f,g=p.objgrad(form,nodes,th)
        for u in range(n):
            f0,g0=O.objgrad_pair(s,form,u,th[u])
            assert abs(f[u]-f0)<=1e-12*max(1,abs(f0)),(form,u,f[u],f0)
            assert np.abs(g[u]-g0).max()<=1e-12,(form,u,np.abs(g[u]-g0).max())
        print('objgrad',form,'ok')
import __graft_entry__ as ge
ge.smoke()
for name in 'abc':
    s=np.loadtxt(f'tests/golden/{name}_samples.csv',delimiter=',')
    for form,F in [('RISE',gml.RISE),('logRISE',gml.logRISE),('RPLE',gml.RPLE)]:
        m=gml.HIP(tol=1e-11)
        R=gml.learn(s,F(),m)
        G=np.loadtxt(f'tests/golden/{name}_{form}_learned.csv',delimiter=',')
        print(name,form,'vs golden %.2e'%np.abs(R-G).max(), 'kkt %.1e'%m.stats['max_kkt'],'it',m.stats['iterations'])
# multi-body
s=np.loadtxt('tests/golden/c_samples.csv',delimiter=',')
m=gml.HIP(tol=1e-11)
fg=gml.learn(s,gml.multiRISE(0.2,False,3),m)
rec,kk=O.learn_multi(s,c=0.2,symmetrize=False,order=3)
print('multi3 err',max(abs(fg[k]-v) for k,v in rec.items()),len(fg),len(rec))
# synthetic
from importlib import import_module
syn=import_module('gml_amd.synthetic')
for (n,K) in [(32,8192),(256,100000)]:
    spins,J=syn.block_ising(n,K,seed=0)
    t=time.time()
    with gml.Problem(spins=spins) as p:
        t1=time.time()
        out,kkt,st=p.learn('RISE',0.4,tol=1e-10,verbose=1)
        print(n,K,'create %.2fs learn %.2fs'%(t1-t,time.time()-t1),st)
        print('bench',p.bench_pass('RISE',out,steps=3,warmup=1))
    if n==32:
        hist=np.concatenate([np.ones((K,1)),spins],axis=1)
        R0,k0,_=O.learn_pair(hist,'RISE',0.4,symmetrize=False)
        print('vs oracle',np.abs(out-R0).max())
    print('true-model err',np.abs(0.5*(out+out.T)-J).max())
