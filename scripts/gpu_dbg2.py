import sys, numpy as np
sys.path.insert(0,'.')
import gml_amd as gml
np.set_printoptions(linewidth=200, precision=4)
s=np.loadtxt('tests/golden/mvt_samples.csv',delimiter=',')
lam=gml.lib().gml_lambda(0.2,9,s[:,0].sum())
with gml.Problem(s) as p:
    out,kkt,st=p.learn('RISE',0.2,tol=1e-6,precision='i8x',verbose=0,raise_on_fail=False)
    out2,kkt2,st2=p.learn('RISE',0.2,tol=1e-11,precision='f64')
    n=9
    f,g=p.objgrad('RISE',np.arange(n),out,precision='i8x')
    f2,g2=p.objgrad('RISE',np.arange(n),out,precision='f64')
    for u in [1,0]:
        print('node',u,'kkt',kkt[u])
        print(' x  ',out[u]); print(' x64',out2[u])
        print(' g/lam i8 ',g[u]/lam); print(' g/lam f64',g2[u]/lam)
