#!/usr/bin/env python3
"""Would a NODE table of tan(beta / 2) (every S-th projector column, cubic through four nodes -- what the scan kernels do for the camera rays) be accurate
enough to replace the per-projector-pixel table the fused kernel gathers from?  CPU estimate with the oracle's cv2.undistortPoints restatement, for
bench.py's two rigs.  Answer (notes/r05.md): no -- the reference's own projector lens (k1 -0.28, k2 6.7, k3 -31.6) makes the 5-iteration
undistortion discontinuous towards the raster's edges (3-5 % of the pixels off by more than 2.4e-7 rad even at S = 4), and on the covering rig's mild
lens the truncated iteration is only smooth enough at S = 4 (a 2 MB table: no longer L2-resident next to the streams).   python tests/analysis/proj_node_error.py"""
import os
import sys

import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import bench, oracle_np as onp
def tan_half(rays, T):
    px, py = rays[:,0].astype(np.float32), rays[:,1].astype(np.float32)
    qn = np.sqrt((px*px+py*py)+np.float32(1)).astype(np.float32)
    tl = np.linalg.norm(T)
    c = ((T[0]*px.astype(np.float64)+T[1]*py.astype(np.float64))+T[2])/(tl*qn.astype(np.float64))
    s = np.sqrt(np.maximum(0,1-c*c))
    th = np.where(c>=0, s/(1+c), (1-c)/np.maximum(s,1e-300))
    return np.minimum(th,1e9).astype(np.float32)
for wl in ["c2_1920x1080x44","c1_1280x720x42"]:
  W,H,pw,ph,_ = bench.WORKLOADS[wl]
  for rig in ("covering","survey"):
    K,cd,pk,pd,R,T = bench.calibration(W,H,pw,ph, rig=rig)
    T = np.asarray(T,float).reshape(3)
    print(wl, rig, "proj dist", np.asarray(pd).ravel())
    for sh in (2,3,4):
        S=1<<sh; allerr=[]
        for y in np.arange(0,ph,max(1,ph//16)):
            xn = np.arange(-S, pw+3*S, S, dtype=np.float32)
            nodes = tan_half(onp.undistort_points(np.stack([xn,np.full_like(xn,y)],1), pk, pd, None).reshape(-1,2), T)
            xs = np.arange(pw, dtype=np.float32)
            exact = tan_half(onp.undistort_points(np.stack([xs,np.full(pw,y,np.float32)],1), pk, pd, None).reshape(-1,2), T)
            pu = np.arange(pw); m = pu>>sh; f=(pu&(S-1))
            t=(f/S); lm=-t*(t-1)*(t-2)/6; l1=-(t+1)*t*(t-2)/2; l2=(t+1)*t*(t-1)/6
            n=nodes.astype(float); n0=n[m+1]
            got = n0 + lm*(n[m]-n0)+l1*(n[m+2]-n0)+l2*(n[m+3]-n0)
            e = 2*np.abs(got-exact.astype(float))/(1+exact.astype(float)**2)
            allerr.append(e)
        e=np.concatenate(allerr)
        print("  S=%2d: max %.3g  p99.9 %.3g  p99 %.3g  median %.3g  frac>1.2e-7 %.4f  frac>2.4e-7 %.4f"%(S,e.max(),np.quantile(e,.999),np.quantile(e,.99),np.median(e),(e>1.2e-7).mean(),(e>2.4e-7).mean()))
