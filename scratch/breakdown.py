import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
W,Hh=3840,2160
import itertools
for (mask,d),var in itertools.product([(17,4),(8,4),(31,1),(31,2),(31,4)],[0,128]):
    ctx=make_default_context(H,tex,mask=mask,options=[(100,var)])
    ts=[]
    for i in range(6):
        _,st=ctx.render(cam,W,Hh,d,want_stats=True); ts.append(st.kernel_ms)
    sctx=make_default_context(H,tex,mask=mask,options=[(OPT_STATS,1)])
    _,st=sctx.render(cam,W,Hh,d,want_stats=True)
    print('  longest wave %.1f us, mean wave %.1f us, waves %d, stats kernel %.3f ms'%(st.reserved[0]/100.0, st.reserved[1]/100.0/ (W*Hh/64), W*Hh/64, st.kernel_ms))
    print('var',var,'mask %2d depth %d: %.3f ms  rays %.1fM (p %.1f s %.1f sh %.1f) nodes %.1fM tris %.1fM'%(mask,d,min(ts),st.rays/1e6,st.primary_rays/1e6,st.secondary_rays/1e6,st.shadow_rays/1e6,st.bvh_node_tests/1e6,st.triangle_tests/1e6),flush=True)
