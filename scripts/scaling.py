"""Kernel time of the HIP path against frame size, per scene subset (run on a GPU box)."""
import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
for mask in (31,8,17):
    ctx=make_default_context(H,tex,mask=mask)
    for (W,Hh) in [(480,270),(960,540),(1920,1080),(3840,2160),(7680,4320)]:
        ts=[]
        for i in range(5):
            _,st=ctx.render(cam,W,Hh,4,want_stats=True); ts.append(st.kernel_ms)
        print('mask %2d %5dx%-5d %.3f ms   %.2f ns/pixel'%(mask,W,Hh,min(ts),min(ts)*1e6/(W*Hh)),flush=True)
