#!/usr/bin/env python3
"""CPU only: how compressible are the decoded (h, v) maps the `maps` exchange puts on the links?  Deltas to the previous valid pixel inside 64-pixel runs,
by synthetic capture (NOTES.md: "Not tried yet").  python tests/analysis/map_delta_stats.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,"oracle")); sys.path.insert(0, os.path.join(ROOT,"tools")); sys.path.insert(0, ROOT)
import oracle_c as oc, oracle_np as onp
from benchlib.common import SCENES, calibration
oc.set_threads(8)
W,H,N=1920,1080,44; pw,ph=1920,1200
def stats(name,h,v):
    h=h.reshape(-1).astype(np.int64); v=v.reshape(-1).astype(np.int64)
    n=h.size//64*64
    ok=((h!=-1)&(v!=-1))[:n].reshape(-1,64); hh=h[:n].reshape(-1,64); vv=v[:n].reshape(-1,64)
    # invalid if either is -1?  a pixel can have h valid and v invalid: treat separately -> need 2 valid bits; check how often exactly one is -1
    one=((h==-1)^(v==-1)).mean()
    # delta vs previous VALID pixel in the run (forward fill)
    def run_ok(a, okm, lo, hi):
        idx=np.where(okm, np.arange(64)[None,:], -1)
        last=np.maximum.accumulate(idx,axis=1)
        prev=np.concatenate([np.full((a.shape[0],1),-1), last[:,:-1]],axis=1)
        base_ok = prev>=0
        prevval=np.take_along_axis(a, np.maximum(prev,0), axis=1)
        d=a-prevval
        bad = okm & base_ok & ((d<lo)|(d>hi))
        return ~bad.any(axis=1), d[okm&base_ok]
    okh=(hh!=-1); okv=(vv!=-1)
    ch, dh = run_ok(hh, okh, -1, 2)
    cv, dv = run_ok(vv, okv, -1, 1)
    comp = ch & cv
    print(f"{name:16s} valid {ok.mean()*100:5.1f}%  exactly-one-invalid {one*100:5.2f}%  runs compressible: h {ch.mean()*100:5.1f}% v {cv.mean()*100:5.1f}% both {comp.mean()*100:5.1f}%  | dh hist", {int(k):int(c) for k,c in zip(*np.unique(np.clip(dh,-3,4),return_counts=True))}, "dv hist", {int(k):int(c) for k,c in zip(*np.unique(np.clip(dv,-3,3),return_counts=True))})
for name in ("s-scene","physical","noisy-physical","s-uniform"):
    cfg=SCENES[name]
    if cfg["kind"]=="physical":
        cal=calibration(W,H,pw,ph) if cfg["rig"]!="covering" else None
        from benchlib.common import calibration as calib
        try:
            import inspect
            cal=calib(W,H,pw,ph,rig=cfg["rig"]) if "rig" in inspect.signature(calib).parameters else calib(W,H,pw,ph)
        except Exception as e:
            print("calib", e); continue
        st,_,_,_=onp.synth_physical(N,H,W,(pw,ph),cal,seed=3,noise=cfg["noise"],gains=cfg["gains"],r2_max=cfg["r2_max"])
    elif cfg["kind"]=="uniform":
        st=onp.synth_uniform(N,H,W,seed=7)
    else:
        st=onp.synth_scene_int(N,H,W,seed=1,noise=cfg["noise"])[0]
    h,v=oc.decode(st)
    stats(name,h,v)
