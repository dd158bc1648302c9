import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_STATS, OPT_PRUNE
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
W,Hh=3840,2160
for d in (1,4):
    ctx=make_default_context(H,tex,mask=31,options=[(100,2+64)])
    img=ctx.render(cam,W,Hh,d)
    t=(img[::8,::8,0].astype(int)+256*img[::8,::8,1].astype(int))*0.25
    print('depth',d,'tiles',t.shape,'sum us',t.sum(),'max',t.max(),'p99',np.percentile(t,99),'p99.9',np.percentile(t,99.9),'mean',t.mean())
    ys,xs=np.nonzero(t>t.max()*0.5)
    print(' tiles above half max:',len(ys), list(zip(ys[:20].tolist(),xs[:20].tolist())))
    hist=np.histogram(t,bins=[0,2,5,10,20,50,100,200,500,1000,100000])[0]
    print(' hist',hist.tolist())
    np.save(os.path.join(R,'gpurun_out','costmap_d%d.npy'%d),t.astype(np.float32))
