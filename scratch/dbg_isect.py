import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np
from conftest import make_default_context
from raytracing_amd.rtx import Backend, hip_backend, ASSETS_DIR
import oracle
H=hip_backend(); O=Backend(oracle.ORACLE_LIB_PATH)
tex={'water':O.decode_jpeg(ASSETS_DIR+'/water.jpg'),'map.jpg':O.decode_jpeg(ASSETS_DIR+'/map.jpg')}
k=dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests/golden/kats.npz')))
got=make_default_context(H,tex,sky='map.jpg').intersect_rays(k['orig'],k['dir'])
want=make_default_context(O,tex,sky='map.jpg').intersect_rays(k['orig'],k['dir'])
print('oracle==golden', np.array_equal(want.view(np.uint32), k['isect'].view(np.uint32)))
both=(got[:,0]>0)&(want[:,0]>0)
neq=(got.view(np.uint32)!=want.view(np.uint32))
rows=np.nonzero(both & neq.any(axis=1))[0]
print(len(rows), 'col mismatch counts', neq[rows].sum(axis=0))
for i in rows[:12]:
    cols=np.nonzero(neq[i])[0]
    print(i,'spec',want[i,15],'a',want[i,8:12],'cols',cols,'got',got[i][cols],'want',want[i][cols], 'N', want[i,4:7])
