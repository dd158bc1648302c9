"""Kernel time with and without the schedule feedback, per scene subset, 3840x2160 depth 4 (run on a GPU box)."""
import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_SCHEDULE_FEEDBACK
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
for mask in (17,8,31):
    for fb in (0,1):
        ctx=make_default_context(H,tex,mask=mask,options=[(OPT_SCHEDULE_FEEDBACK,fb)])
        ts=[]
        for i in range(8):
            _,st=ctx.render(cam,3840,2160,4,want_stats=True); ts.append(st.kernel_ms)
        print('mask %2d feedback %d: first %.3f second %.3f best of later %.3f ms'%(mask,fb,ts[0],ts[1],min(ts[2:])),flush=True)
