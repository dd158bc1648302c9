import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS, OPT_MESH_GRID, OPT_BVH_BUILDER
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
for mask,opts,tag in ((31,[],"sah"),):
    ctx=make_default_context(H,tex,mask=mask,options=opts)
    for d in (1,4):
        for (W,Hh) in [(3840,2160)]:
            ts=[]
            for i in range(5):
                _,st=ctx.render(cam,W,Hh,d,want_stats=True); ts.append(st.kernel_ms)
            sctx=make_default_context(H,tex,mask=mask,options=opts+[(OPT_STATS,1)])
            _,ss=sctx.render(cam,W,Hh,d,want_stats=True)
            print('%-22s mask %2d d%d %4dx%-4d %.3f ms  nodes %.2fM tris %.2fM'%(tag,mask,d,W,Hh,min(ts),ss.bvh_node_tests/1e6,ss.triangle_tests/1e6),flush=True)
