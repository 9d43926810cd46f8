import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from audio_feature_extraction_amd import _native as N
from audio_feature_extraction_amd.synth import make_batch
samples, offsets, lengths = make_batch(1000, 22050, 10.0, workers=16)
ctxs=[N.Context(0), N.Context(0)]
dbuf=N.DeviceBuffer(ctxs[0], samples.nbytes); dbuf.upload(samples)
plans=[N.Plan(c, N.make_params(22050,1024,256,13)) for c in ctxs]
outs=[None,None]; busy=[False,False]
def run(k):
    for i in range(k):
        p=i%2
        if busy[p]: outs[p]=plans[p].extract_collect()
        plans[p].extract_submit(dbuf, offsets, lengths, out=outs[p]); busy[p]=True
    for p in (0,1):
        if busy[p]: outs[p]=plans[p].extract_collect(); busy[p]=False
run(10)
for p in plans: p.set_timing(True); p.timings(reset=True)
run(12)
sp=[]
for i,p in enumerate(plans):
    for k in ("frames","trim_decide","dct","stats"):
        for a,b in p.intervals(k): sp.append((a,b,i,k))
sp.sort()
t0=sp[0][0]
for a,b,i,k in sp[8:40]:
    print(f"{a-t0:9.3f} {b-t0:9.3f} {b-a:7.3f} ms  plan {i} {k}")
