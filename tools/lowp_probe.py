import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import conftest
pkg=conftest.load_pkg()
from oracle import oracle as O
import importlib
synth=importlib.import_module(pkg.__name__+'.synth')
net=O.Net()
nhwc=lambda a: np.ascontiguousarray(np.transpose(a,(1,2,0)))
for prec in (0,1,2,3):
    for (H,W,seed,kind) in ((72,104,41,"sparse"),(128,160,42,"dense"),(240,320,43,"sparse")):
        img=synth.make_frame(H,W,seed,0,kind)
        ctx=pkg.Context(pkg.default_config(W,H,precision=prec))
        ctx.inference([img]); ctx.detect_keypoints(1); ctx.detect_lines(1)
        t=net.inference(img,prec)
        out=[]
        for which,k in ((pkg.T_FEAT,"feat"),(pkg.T_JUNC,"junc"),(pkg.T_HEAT_RAW,"heat"),(pkg.T_DESCMAP,"desc")):
            got,ref=ctx.read_tensor(which),nhwc(t[k])
            out.append("%s %.2e"%(k, np.abs(got-ref).max()/np.abs(ref).max()))
        print(prec,H,W,out, flush=True)
        ctx.close()
