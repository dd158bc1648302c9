"""Per-tile duration / casts / node trips / triangle trips of the heaviest tiles (needs the instrumented build)."""
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
ctx=make_default_context(H,tex,mask=31,options=[(OPT_STATS,1)])
for i in range(2): img,st=ctx.render(cam,W,Hh,4,want_stats=True)
def field(k):
    p=img[::8,k::8,:].astype(np.int64); return p[...,0]+256*p[...,1]+65536*p[...,2]
dur,casts,nodes,tris=field(0)*0.1,field(1),field(2),field(3)
print('kernel %.3f ms; tiles %d; sum dur %.0f us'%(st.kernel_ms,dur.size,dur.sum()))
idx=np.argsort(dur.ravel())[::-1][:12]
for i in idx:
    y,x=divmod(i,dur.shape[1])
    print('tile (%3d,%3d): %.0f us, max casts/lane %d, node trips %d, tri trips %d -> %.0f ns per (node+tri) trip; %.1f us per cast'%(y,x,dur[y,x],casts[y,x],nodes[y,x],tris[y,x],1000*dur[y,x]/max(1,nodes[y,x]+tris[y,x]),dur[y,x]/max(1,casts[y,x])))
heavy=dur>300
print('tiles >300us: %d; mean casts %.1f, mean node trips %.0f, tri trips %.0f, dur %.0f'%(heavy.sum(),casts[heavy].mean(),nodes[heavy].mean(),tris[heavy].mean(),dur[heavy].mean()))
print('all tiles: total node trips %d tri trips %d casts(max-lane) %d'%(nodes.sum(),tris.sum(),casts.sum()))
