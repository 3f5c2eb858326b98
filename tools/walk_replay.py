"""CPU replay of the reference's stage-04 walk (04:137-205) over ONE skeleton component, with the walker's bounce memo, to count what a change of the
memo would save (development aid).  Input: gpurun_out/bigcomp_l<layer>.npz as written by `ORIP_SAVE_BIG=1 python tools/skel_stats.py`
on the GPU box (pixel coordinates of the largest component of a heavy layer).  Prints fresh / no-fresh steps, split by "on a listed chain (>= 24 degree-2
pixels)" or not, and how many no-fresh steps a later walk could skip if runs that ended at a start pixel or at a fresh pixel were kept as open records.
usage: python tools/walk_replay.py [MIN_SKIP] [NPZ]"""
import numpy as np, sys, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z=np.load(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'gpurun_out', 'bigcomp_l3.npz')); ys=Z['ys']; xs=Z['xs']
NE=[(-1,-1),(0,-1),(1,-1),(-1,0),(1,0),(-1,1),(0,1),(1,1)]
pix=set(zip(xs.tolist(),ys.tolist())); fg=len(pix)
nbrs={p:[(p[0]+dx,p[1]+dy) for dx,dy in NE if (p[0]+dx,p[1]+dy) in pix] for p in pix}
deg={p:len(nbrs[p]) for p in pix}
# chains: deg-2 maximal groups >=24 -> pixel in long chain?
import scipy.ndimage as ndi
H=ys.max()+2; W=xs.max()+2
m=np.zeros((H,W),bool); m[ys,xs]=True
d2=np.zeros((H,W),bool)
for p in pix:
    if deg[p]==2: d2[p[1],p[0]]=True
lab,n=ndi.label(d2,structure=np.ones((3,3))); sz=np.bincount(lab.ravel())
inchain={p:(deg[p]==2 and sz[lab[p[1],p[0]]]>=24) for p in pix}
MINL=int(sys.argv[1]) if len(sys.argv)>1 else 16
visited=set()
order=sorted(pix,key=lambda p:(p[1],p[0]))
for p0 in [p for p in order if deg[p]==1]:
    if p0 in visited: continue
    visited.add(p0); p=p0; prev=None
    while True:
        nb=[q for q in nbrs[p] if q!=prev and q not in visited]
        if not nb: break
        q=nb[0]; visited.add(q); prev=p; p=q
        if deg[p]>=3 or deg[p]==1: break
closed={}; openrec={}  # state -> (record id, pos)
recs=[]
stepped=0; stepped_nochain=0; skipped=0; skipped_nochain=0; nskips=0; fresh=0; fresh_nochain=0
for p0 in order:
    if p0 in visited: continue
    visited.add(p0); p=p0; prev=None; guard=0
    run=[]; runset={}
    def close_open():
        global run,runset
        if run:
            rid=len(recs); recs.append(list(run))
            for j,t in enumerate(run): openrec[t]=(rid,j)
        run=[]; runset={}
    while True:
        nb=[q for q in nbrs[p] if q!=prev and q not in visited]
        if nb:
            close_open()
            q=nb[0]; visited.add(q); fresh+=1; fresh_nochain+= (not inchain[q])
        else:
            nb=[q for q in nbrs[p] if q!=prev]
            if not nb: close_open(); break
            s=(p,prev)
            if s in closed or s in runset:
                for t in run: closed[t]=True; openrec.pop(t,None)
                run=[]; break
            if s in openrec:
                rid,j=openrec[s]; R=recs[rid]; L=len(R)-1-j
                if L>=MINL and guard+L<fg*4-80:
                    # skip: append R[j..end] to run, move to R[-1] state position: standing at R[-1] pixel with its prev
                    for t in R[j:]:
                        runset[t]=len(run); run.append(t)
                    skipped+=L; nskips+=1; skipped_nochain+=sum(1 for t in R[j+1:] if not inchain[t[0]])
                    guard+=L
                    p,prev=R[-1]
                    # the last state R[-1] is re-examined by the loop (it will be found in runset!) -> emulate: remove it from run so it gets stepped normally
                    run.pop(); del runset[R[-1]]
                    continue
            runset[s]=len(run); run.append(s); stepped+=1; stepped_nochain+=(not inchain[p])
            q=nb[0]
        prev=p; p=q
        if p==p0: close_open(); break
        guard+=1
        if guard>fg*4: close_open(); break
print("MINL",MINL,"fresh",fresh,"(nochain",fresh_nochain,") stepped nofresh",stepped,"(nochain",stepped_nochain,") skipped",skipped,"(nochain",skipped_nochain,") in",nskips,"skips")
