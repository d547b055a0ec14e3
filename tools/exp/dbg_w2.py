import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from roskfpos_amd import capi
from roskfpos_amd.synth import Workload
T=65536+4096+37
w=Workload(T,8)
for storage in (0,2,1):
    real = np.float64 if storage == 0 else np.float32
    err=w.err_est(real)
    os.environ["KFPOS_ONE_WAVE_BUILD"]="0"; a=capi.KfposBank(0,T,w.anchors,storage=storage,init_pos=w.init_positions())
    os.environ["KFPOS_ONE_WAVE_BUILD"]="1"; b=capi.KfposBank(0,T,w.anchors,storage=storage,init_pos=w.init_positions())
    for s in range(3):
        r=w.ranges_mm(s)
        sa,sb=a.step_toa(r,err,w.dt_of(s)),b.step_toa(r,err,w.dt_of(s))
        pa,pb=a.get_pose(0.0)[0],b.get_pose(0.0)[0]
        d=np.abs(pa-pb)
        bad=np.where(d.max(1)>0)[0]
        xa,Pa,_=a.get_state(); xb,Pb,_=b.get_state()
        print("storage",storage,"epoch",s,"status equal",np.array_equal(sa,sb),"pose max diff",d.max(),"n differing",len(bad),"first",bad[:5], "P max diff", np.abs(Pa-Pb).max())
    a.close(); b.close()
