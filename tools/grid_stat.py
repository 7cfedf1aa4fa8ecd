import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
from oracle import oracle as O
from util import compare_features
tot=bad=0; mx=0
for seed,(w,h) in enumerate([(240,180),(200,150),(333,257),(160,120),(640,480),(97,75),(500,90)]):
    for kw in (dict(desc_mode=2), dict(desc_mode=2,sift_mode=2,norm_mode=1), dict(desc_mode=2,sift_mode=1,upscale_factor=0.0)):
        img=synth(300+seed,w,h)
        orc=O.Oracle(O.default_params(**kw),threads=16).run(img)
        ctx=hip.Context(hip.default_params(**kw)); ctx.submit(img)
        st=compare_features(*orc.fetch(),*ctx.fetch())
        tot+=st["n_desc"]; bad+=st["desc_bad"]; mx=max(mx,st["max_desc"])
print("grid mode: %d descriptors, %d beyond 1e-3 (%.4f %%), max %.2e" % (tot,bad,100.0*bad/tot,mx))
