#!/usr/bin/env python3
"""Where does the fused mode (device-evaluated maps) differ from the oracle by more than one level in ONE case of
tests/fuzz/fuzz_fused.py?  Prints, per view, the differing pixels' oracle coordinates, rows and columns, and how many of the
plan's quantised coordinates differ from the oracle's (p2p_job_get_coords).  The case is fuzz_fused's: panorama seed 700 + case.
    python3 tests/fuzz/fused_case_report.py --pw 1024 --ow 474 --oh 344 --fov 120 --yaws 297 314 --pitches 30 129 --pano-seed 735
    (the defaults: round 6's seam row, tests/test_gpu_fused_exceptions.py; named options only, as everywhere under tests/fuzz/)
Uses the oracle (test infrastructure): lives under tests/."""
import importlib, sys, os
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
from _util import oracle_views, oracle_maps
pkg=importlib.import_module("360-to-planer-images_amd"); nat=pkg._native
synth=importlib.import_module("360-to-planer-images_amd.synth")
import argparse
_p = argparse.ArgumentParser(description=__doc__, allow_abbrev=False, formatter_class=argparse.RawDescriptionHelpFormatter)
_p.add_argument("--pw", type=int, default=1024); _p.add_argument("--ow", type=int, default=474); _p.add_argument("--oh", type=int, default=344)
_p.add_argument("--fov", type=int, default=120); _p.add_argument("--pano-seed", type=int, default=735)
_p.add_argument("--yaws", type=int, nargs="+", default=[297, 314]); _p.add_argument("--pitches", type=int, nargs="+", default=[30, 129])
_a = _p.parse_args()
if not (16 <= _a.pw <= 16384 and 1 <= _a.ow <= 4096 and 1 <= _a.oh <= 4096 and len(_a.yaws) <= 16 and len(_a.pitches) <= 16):
    _p.error("sizes out of range (pw 16..16384, views up to 4096 x 4096, at most 16 yaws and 16 pitches)")
pw,ow,oh,fov,yaws,pitches,seed=_a.pw,_a.ow,_a.oh,_a.fov,list(_a.yaws),list(_a.pitches),_a.pano_seed
ph=pw//2
pano=synth.synth_pano(pw,ph,seed,"S")
got=nat.remap_views(pano,yaws,pitches,fov,ow,oh)
want=oracle_views(pano,yaws,pitches,ow,oh,fov)
_,U,V=oracle_maps(yaws,pitches,ow,oh,pw,ph,fov)
d=np.abs(got.astype(int)-want.astype(int)).max(axis=-1)
for yi in range(len(yaws)):
    for pi in range(len(pitches)):
        idx=np.argwhere(d[yi,pi]>1)
        if len(idx)==0: continue
        vs=V[pi][idx[:,0],idx[:,1]]; us=U[pi][idx[:,0],idx[:,1]]
        print("yaw",yaws[yi],"pitch",pitches[pi],len(idx),"pixels; U values",sorted(set(np.round(us,3).tolist()))[:6],"; V range %.2f..%.2f"%(vs.min(),vs.max()), "diffs", sorted(d[yi,pi][idx[:,0],idx[:,1]].tolist())[-5:], "cols", sorted(set(idx[:,1].tolist()))[:12], "rows", sorted(set(idx[:,0].tolist()))[:12])
# device coords vs oracle coords
ctx=nat.Context(0); job=nat.Job(ctx,pw,ph,1,yaws,pitches,fov,ow,oh); job.set_pano(0,pano); job.run()
c=job.get_coords()   # [n_pitch][oh][ow][2] quantised
for pi in range(len(pitches)):
    qu=np.rint(U[pi].astype(np.float32)*32).astype(np.int64); qv=np.rint(V[pi].astype(np.float32)*32).astype(np.int64)
    du=np.abs(c[pi][...,0].astype(np.int64)-qu); dv=np.abs(c[pi][...,1].astype(np.int64)-qv)
    du=np.minimum(du, np.abs(du-pw*32))
    print("pitch",pitches[pi],"coords differing: U %d (max %d / 32 px)  V %d (max %d)"%((du>0).sum(),du.max(),(dv>0).sum(),dv.max()))
job.close(); ctx.close()
