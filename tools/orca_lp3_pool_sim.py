"""usage (CPU, here): python tools/orca_lp3_pool_sim.py -- what pooling linearProgram3's tickets over the FOUR wavefronts of a workgroup (through LDS, a barrier
per round) would change, counted on the restatement's own lines like tools/orca_lp3_stats.py: passes and vote-loop trips per wavefront, today (each wavefront deals
its own agents) against the pooled deal (a round's passes dealt round-robin over the four wavefronts; the round ends with the slowest of them).  A statistics
tool, not a parity oracle.  Result (HISTORY.md 0.6): the mean trips per wavefront drop by 22 - 26 %, the trips of the slowest wavefront of every round do not."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import importlib.util
spec=importlib.util.spec_from_file_location('st','/root/repo/tools/orca_lp3_stats.py'); st=importlib.util.module_from_spec(spec); spec.loader.exec_module(st)
from oracle import crowd_oracle as orc
from social_navigation_pyenvs_amd import scenarios as sc
f32=np.float32
W,n=128,25; wpb=2
pos,yaw,g=sc.circular_crossing(W,n,7.0,1000)
S=sc.make_states(pos,yaw,g).astype(np.float32); g=g.astype(np.float32)
d=g[:,:,0]-S[:,:,0:2]; S[:,:,5:7]=d/np.linalg.norm(d,axis=-1,keepdims=True)
margin=np.full((W,n),0.01,np.float32)
done=0
for ph in (100,300,500,600):
    S,g,_=orc.orca_step_block(S,g,margin,0.0125,ph-done); done=ph
    allpaths=[]
    for w in range(W):
        rad=S[w,:,8]+margin[w]
        _,lines,nl=orc.orca_new_velocities(S[w,:,0:2],S[w,:,3:5],S[w,:,5:7],rad,S[w,:,12],time_step=0.0125,return_lines=True)
        paths=[]
        for a in range(n):
            L=[tuple(f32(x) for x in lines[a,k]) for k in range(nl[a])]
            failed,r,_=st.lp2(L,f32(S[w,a,12]),f32(S[w,a,5]),f32(S[w,a,6]),False,None)
            paths.append(st.lp3_path(L,failed,f32(S[w,a,12]),r) if failed<len(L) else [])
        allpaths.append(paths)
    waves=[sum(allpaths[w0:w0+wpb],[]) for w0 in range(0,W,wpb)]
    cur=np.mean([st.wave_cost(wv,'lane') for wv in waves],axis=0)
    pools=[sum(waves[k:k+4],[]) for k in range(0,len(waves),4)]
    # pooled: passes of a round are dealt round-robin over 4 waves: per-wave trips in a round = max over waves of the sum of its passes' trips
    def pooled_cost(paths):
        rounds=passes=0; wave_trips=0.0; tot_trips=0
        depth=max((len(p) for p in paths),default=0)
        for rd in range(depth):
            pend=[(a,p[rd]) for a,p in enumerate(paths) if len(p)>rd]
            if not pend: break
            rounds+=1
            per=[0,0,0,0]; k=0
            for grp,width in (([x for x in pend if x[1][0]<=8],8),([x for x in pend if x[1][0]==9],4)):
                grp=sorted(grp,key=lambda x:x[1][0])
                for p0 in range(0,len(grp),width):
                    c=[x[1][1] for x in grp[p0:p0+width]]
                    t=min(9,max(c)+1); per[k%4]+=t; k+=1; passes+=1; tot_trips+=t
            wave_trips+=max(per)
        return rounds,passes,wave_trips,tot_trips
    pc=np.mean([pooled_cost(p) for p in pools],axis=0)
    print(f"substep {ph}: current per wave: rounds {cur[0]:.2f} passes {cur[1]:.2f} trips {cur[2]:.1f} | pooled over 4 waves: rounds {pc[0]:.2f} passes/wave {pc[1]/4:.2f} trips on the slowest wave {pc[2]:.1f} (mean {pc[3]/4:.1f})")
