"""Event-driven model of per-lane state-machine schedules for the UCT kernel (profiles/r04_uct_fsm.md): rollout lengths
of the real headline MDP under the uniform policy, cost R per rollout trip and T per tree phase (wave-wide: a phase costs
the wave its instructions whenever ONE lane needs it).  CPU only:  PYTHONPATH=. python tools/sim_uct_fsm.py"""
import numpy as np
from rl_agents_amd.envs import generators
cfg = generators.highway_shaped(10, 10, 100, seed=0)
t, r, term = cfg["transition"], cfg["reward"], np.asarray(cfg["terminal"])
E,H=33,30
rs=np.random.Generator(np.random.PCG64(1))
non_term=np.flatnonzero(~term)
NR=64*256
roots=rs.choice(non_term,size=NR)
# episode lengths: random-policy steps from root until terminal[s] (source rule: done when acting FROM terminal) or H
L=np.zeros((NR,E),dtype=np.int32)
for e in range(E):
    s=roots.copy(); alive=np.ones(NR,bool); n=np.zeros(NR,np.int32)
    for h in range(H):
        a=rs.integers(0,5,size=NR)
        done=term[s]
        n+=alive
        s=np.where(alive,t[s,a],s)
        alive&=~done
    L[:,e]=n
print("mean len",L.mean(),"per root total mean",L.sum(1).mean())
R=1.0
def current(T):
    w=L.reshape(-1,64,E)
    return (w.max(1)*R+T).sum(1).mean()          # per wave
def fsm(T,theta,lanes=64,roots_per_wave=64,dyn=False):
    # event simulation per wave; tree-phase cost T charged to whole wave whenever run; rollout trip cost R
    tot=[]
    w=L.reshape(-1,roots_per_wave,E)
    for wave in w[:64]:
        nxt=lanes                      # next root to hand out (dynamic)
        root=list(range(lanes)); ep=[0]*lanes; rem=[0]*lanes   # rem = remaining rollout steps; 0 => needs tree
        state=[1]*lanes                # 1 = waiting for tree, 2 = rolling, 0 = done
        time=0.0
        while True:
            # tree phase for waiting lanes
            if any(s==1 for s in state):
                time+=T
                for i in range(lanes):
                    if state[i]==1:
                        if ep[i]==E:
                            if dyn and nxt<roots_per_wave:
                                root[i]=nxt; nxt+=1; ep[i]=0
                            else:
                                state[i]=0; continue
                        rem[i]=wave[root[i],ep[i]]; ep[i]+=1; state[i]=2
            if all(s==0 for s in state): break
            # rollout trips until waiting >= theta or none rolling
            while True:
                time+=R
                nw=0; nr=0
                for i in range(lanes):
                    if state[i]==2:
                        rem[i]-=1
                        if rem[i]<=0: state[i]=1
                    if state[i]==1: nw+=1
                    elif state[i]==2: nr+=1
                if nr==0 or nw>=theta: break
        tot.append(time)
    return np.mean(tot)
for T in (4.0,7.0,10.0):
    c=current(T)
    print("T=%g current per wave %.0f"%(T,c))
    for th in (8,16,24,32,48):
        f=fsm(T,th)
        print("   theta %d: fsm %.0f (x%.2f)"%(th,f,c/f), end='')
        f2=fsm(T,th,lanes=32,roots_per_wave=64,dyn=True)   # 32 lanes, 64 roots: same work per wave object, half the lanes -> compare throughput per lane-time
        print("   dyn 32 lanes/64 roots: %.0f (lane-time x%.2f)"%(f2, c/(f2/2)))
