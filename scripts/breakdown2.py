"""Kernel time of the 4K frame per scene subset and depth, including depth 0 (sky lookups only) -- run on a GPU box."""
import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
W,Hh=3840,2160
for (mask,d) in [(31,0),(0,1),(31,4)] if len(sys.argv) > 1 else [(31,0),(0,1),(0,4),(16,4),(1,4),(17,4),(2,4),(4,4),(23,4),(8,4),(31,4)]:
    ctx=make_default_context(H,tex,mask=mask)
    ts=[]
    for i in range(8):
        _,st=ctx.render(cam,W,Hh,d,want_stats=True); ts.append(st.kernel_ms)
    sctx=make_default_context(H,tex,mask=mask,options=[(OPT_STATS,1)])
    _,st=sctx.render(cam,W,Hh,d,want_stats=True)
    print('mask %2d depth %d: first %.3f best later %.3f ms | rays %.1fM (p %.1f s %.1f sh %.1f) sky %.1fM lit %.1fM sphere tests %.0fM box %.0fM tri %.0fM nodes %.1fM'%(mask,d,ts[0],min(ts[1:]),st.rays/1e6,st.primary_rays/1e6,st.secondary_rays/1e6,st.shadow_rays/1e6,st.sky_lookups/1e6,st.lit_lights/1e6,st.sphere_tests/1e6,st.box_tests/1e6,st.loose_tri_tests/1e6,st.bvh_node_tests/1e6),flush=True)
