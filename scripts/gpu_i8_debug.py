import sys, numpy as np
sys.path.insert(0,'.')
import gml_amd as gml
from importlib import import_module
syn=import_module('gml_amd.synthetic')
n,K=512,200000
spins,J=syn.block_ising(n,K,block=16,seed=5)
theta=J.copy()
with gml.Problem(spins=spins) as p:
    fa,ga=p.objgrad("RISE",np.arange(n),theta,precision="f64")
    fb,gb=p.objgrad("RISE",np.arange(n),theta,precision="i8x")
rel=np.abs(fb/fa-1)
u=int(rel.argmax()); print('worst row',u,rel[u],'sum|theta|',np.abs(theta[u]).sum())
# emulate
LF=5
th=theta[u]; mx=np.abs(th).max(); ex=np.frexp(mx)[1]; sg=2.0**(ex-(8*LF-2))
q=np.rint(th/sg); thq=q*sg
emax=np.abs(q).sum()*sg
w=1.0/K
B=w*np.exp(emax); eb=np.frexp(B)[1]; tau=2.0**(eb-30)
S=spins.astype(np.float64)
x=S.copy(); x[:,u]=1.0
E=S[:,u]*(x@thq)
v=w*np.exp(-E)
vq=np.rint(v/tau)
print('emax',emax,'tau',tau,'max vq',vq.max(),'f exact',v.sum(),'f quant emu',tau*vq.sum(),'gpu i8',fb[u],'gpu f64',fa[u])
print('emu rel err',tau*vq.sum()/v.sum()-1,' gpu-emu',fb[u]-tau*vq.sum())
