"""Kernel time of the default scene at 3840x2160 with the bunny subdivided to 79 k / 318 k / 1.27 M triangles (run on a GPU box)."""
import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from test_gpu_scenes import _subdivide
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
v,f=H.load_model(os.path.join(ASSETS_DIR,'bunny.obj'),45.0,(4.0,-7.0+1.4,-23.0))
mats=H.default_materials()
for levels in (0,2,3,4):
    verts,idx=_subdivide(v,f,levels) if levels else (v,f)
    for stats in (0,1):
        ctx=make_default_context(H,tex,mask=31-8,options=[(OPT_STATS,stats)],commit=False)
        ctx.set_mesh(verts,idx,7)
        t0=time.time(); ctx.commit(); tc=time.time()-t0
        ts=[]
        for i in range(5):
            _,st=ctx.render(cam,3840,2160,4,want_stats=True); ts.append(st.kernel_ms)
        if stats: print('   nodes %.1fM tris %.1fM rays %.1fM'%(st.bvh_node_tests/1e6,st.triangle_tests/1e6,st.rays/1e6),flush=True)
        else: print('%8d triangles: commit %.3f s, first frame %.3f ms, best later %.3f ms'%(len(idx),tc,ts[0],min(ts[1:])),flush=True)
