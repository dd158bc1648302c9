import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR
H=hip_backend()
tex={'water':H.decode_jpeg(ASSETS_DIR+'/water.jpg'),'envmap.jpg':H.decode_jpeg(ASSETS_DIR+'/envmap.jpg')}
cam=H.default_camera()
W,Hh=3840,2160
for mask in (17,31):
    ctx=make_default_context(H,tex,mask=mask)
    for i in range(2): img,st=ctx.render(cam,W,Hh,4,want_stats=True)
    t=(img[::8,::8,0].astype(int)+256*img[::8,::8,1].astype(int))*0.1   # us
    start=img[::8,::8,2].astype(int)*10.0  # us
    print('mask',mask,'kernel %.3f ms'%st.kernel_ms,'sum tile-us %.0f'%t.sum(),'= %.3f ms over 3072 slots'%(t.sum()/3072/1000),'max %.0f'%t.max(),'mean %.1f'%t.mean())
    bins=[0,5,10,20,50,100,200,500,1000,100000]
    print(' bins',bins); print(' count',np.histogram(t,bins=bins)[0].tolist()); print(' time share %',(100*np.histogram(t,bins=bins,weights=t)[0]/t.sum()).round(1).tolist())
    # when do heavy tiles start?
    for thr in (100,200,500):
        m=t>thr
        if m.any(): print('  tiles > %d us: %d, start time us: min %.0f median %.0f max %.0f; end max %.0f'%(thr,m.sum(),start[m].min(),np.median(start[m]),start[m].max(),(start[m]+t[m]).max()))
    late=(start+t)>0.9*(start+t).max()
    ys,xs=np.nonzero(late)
    print('  tiles ending in the last 10%%: %d; their durations: median %.0f max %.0f; rows %d..%d cols %d..%d'%(late.sum(),np.median(t[late]),t[late].max(),ys.min(),ys.max(),xs.min(),xs.max()))
    np.save(os.path.join(R,'gpurun_out','cost_m%d.npy'%mask),np.stack([t,start]).astype(np.float32))
