import sys, time, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as e; e.build()
from advntr_amd import _lib, workloads
rng=np.random.default_rng(3)
t=time.time(); loc=workloads.make_locus(rng,100,30,40,error_rate=0.3); print('build',time.time()-t)
a=loc.model.baked_arrays(); print('m',a['m'],'E',len(a['in_src']))
reads=[workloads.make_reads(rng,loc,1,int(n),locus_fraction=1.0,sub_rate=0.1)[0] for n in rng.integers(900,1500,int(sys.argv[1]) if len(sys.argv) > 1 else 3000)]
bases,off=_lib.encode_reads(reads)
dm=loc.model.device_model()
info=np.zeros(4,np.int32)
_lib.load().advntr_hmm_info(dm.handle, info[0:].ctypes.data, info[1:].ctypes.data, info[2:].ctypes.data, info[3:].ctypes.data)
NC=int(info[3]); print('NC',NC)
for flags,name in (((0,'tiled'),) if len(sys.argv) > 2 else ((0,'tiled'),(_lib.FLAG_STREAM,'stream'),(_lib.FLAG_FORCE_GENERIC,'generic'))):
    B=_lib.DeviceBatch([dm],bases,off,np.zeros(len(reads),np.int32),flags=flags)
    B.run(); B.sync()
    ms=B.run_timed(2)
    cells=sum(len(r) for r in reads)*NC
    print(name,'ms',ms,'reads/s',len(reads)/ms*1e3,'cells/s %.3g'%(cells/ms*1e3))
    lp,_=B.fetch()
    if name=='tiled': ref=lp
    else: print('  equal to tiled:', np.array_equal(ref,lp))
    B.close()
