"""Kernel time of one rank's share of the 4K frame (rows r, r+G, ...) for G = 1, 2, 4, 8 on one GPU (run on a GPU box)."""
import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
ctx=make_default_context(H,tex)
W,Hh,D=3840,2160,4
for G in (1,2,4,8):
    ts=[]
    for r in range(min(G,3)):
        best=1e9
        for i in range(5):
            _,st=ctx.render(cam,W,Hh,D,row_first=r,row_step=G,want_stats=True); best=min(best,st.kernel_ms)
        ts.append(best)
    print('G=%d: kernel per rank %s ms (ideal %.3f)'%(G,', '.join('%.3f'%t for t in ts),ts[0] if G==1 else 0),flush=True)
