import sys, time; import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch, make_clip
from oracle import pyin_ref as P
SR=22050
ctx=N.Context(0); plan=N.Plan(ctx, N.make_params(SR,1024,256,13))
# parity detail on a few synthetic clips
for idx in (0,1,2):
    y=make_clip(idx,SR,2.0,speechy=(idx==2))
    out=plan.f0_batch(y,np.zeros(1,np.int64),np.array([y.size],np.int64),P.C2_HZ,P.C7_HZ,flags=0,want_frames=True)
    f0=out['f0_flat'][:1+y.size//256]
    r,_,_=P.pyin(y)
    same=(np.isnan(f0)==np.isnan(r)); v=~np.isnan(f0)&~np.isnan(r); same[v]&=np.abs(f0[v]-r[v])<=1e-9*r[v]
    print('clip',idx,'frames',len(r),'match',same.mean(),'voiced ref',(~np.isnan(r)).mean(),'stats',out['stats'][0], P.extract_f0(y))
# throughput
n=1000
samples,offsets,lengths=make_batch(n,SR,10.0,workers=16)
d=N.DeviceBuffer(ctx,samples.nbytes); d.upload(samples)
plan.f0_batch(d,offsets,lengths,P.C2_HZ,P.C7_HZ)
t0=time.perf_counter(); out=plan.f0_batch(d,offsets,lengths,P.C2_HZ,P.C7_HZ); dt=time.perf_counter()-t0
fr=int((1+lengths//256).sum())
print('f0 batch',n,'clips',fr,'frames',dt*1e3,'ms ->',fr/dt,'frames/s; voiced mean', out['stats'][:,3].mean())
