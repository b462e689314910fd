import sys, time, numpy as np
sys.path.insert(0,'.')
import gml_amd as gml
from oracle import oracle as O
from importlib import import_module
syn=import_module('gml_amd.synthetic')

def quant(theta, LF=5):
    mx=np.abs(theta).max(1)
    ex=np.where(mx>0, np.frexp(mx)[1], 0)
    sg=np.ldexp(1.0, ex-(8*LF-2))
    return np.rint(theta/sg[:,None])*sg[:,None]

rng=np.random.default_rng(0)
s=np.loadtxt('tests/golden/mvt_samples.csv',delimiter=',')
n=s.shape[1]-1
th=rng.normal(scale=0.3,size=(n,n)); thq=quant(th)
with gml.Problem(s) as p:
    for form in ['RISE','logRISE','RPLE']:
        f,g=p.objgrad(form,np.arange(n),th,precision='i8x')
        f2,g2=p.objgrad(form,np.arange(n),thq,precision='f64')
        print('mvt',form,'i8x vs f64(theta_q): f %.2e g %.2e'%(np.abs(f/f2-1).max(), np.abs(g-g2).max()), ' vs f64(theta): g %.2e'%np.abs(g-p.objgrad(form,np.arange(n),th)[1]).max())
for (n,K) in [(64,5000),(256,100000)]:
    spins,J=syn.block_ising(n,K,seed=0)
    th=J+rng.normal(scale=0.01,size=J.shape)*(rng.random(J.shape)<0.1); thq=quant(th)
    with gml.Problem(spins=spins) as p:
        f,g=p.objgrad('RISE',np.arange(n),th,precision='i8x')
        f2,g2=p.objgrad('RISE',np.arange(n),thq,precision='f64')
        print(n,K,'i8x vs f64(theta_q): f %.2e g %.2e'%(np.abs(f/f2-1).max(), np.abs(g-g2).max()))
        fb,gb=p.objgrad('RISE',np.arange(n),th,precision='i8x')
        print('  deterministic:',np.array_equal(f,fb) and np.array_equal(g,gb))
        t=time.time(); out,kkt,st=p.learn('RISE',0.4,tol=1e-9,precision='i8x',verbose=1,raise_on_fail=False); print('  learn i8x %.2fs'%(time.time()-t),{k:st[k] for k in ['iterations','passes','forward_passes','max_kkt','not_converged','t_pass','t_hess','t_host']})
        out2,kkt2,st2=p.learn('RISE',0.4,tol=1e-10,precision='f64')
        print('  i8x vs f64 solution: max abs %.2e rel F %.2e'%(np.abs(out-out2).max(), np.linalg.norm(out-out2)/np.linalg.norm(out2)))
        print('  bench f64',p.bench_pass('RISE',out2,steps=3,warmup=1,precision='f64'))
        print('  bench i8x',p.bench_pass('RISE',out2,steps=3,warmup=1,precision='i8x'))
