"""processor-sharing simulation of the conv grid on 256 CUs (tile works from scratch/lvl_masks_*.npy)"""
import numpy as np, sys, heapq
def tiles(level, rows=32):
    m=np.load(f'scratch/lvl_masks_{level}.npy').astype(np.uint32)
    ms=np.sort(m); pad=(-len(ms))%rows
    ms=np.concatenate([ms,np.zeros(pad,np.uint32)]).reshape(-1,rows)
    s=ms.reshape(len(ms),rows//16,16); so=np.bitwise_or.reduce(s,axis=2)
    sw=np.array([[bin(x).count('1') for x in r] for r in so]).sum(1)  # strip-offsets per tile
    return sw
def sim(work, ncu=256, slots=4, prio=False, fixed=0.0):
    # work: per-WG cost (exclusive SIMD time); order given
    cus=[[] for _ in range(ncu)]  # remaining works
    t=0.0; nxt=0; n=len(work)
    # initial fill round-robin
    for s in range(slots):
        for c in range(ncu):
            if nxt<n: cus[c].append(work[nxt]+fixed); nxt+=1
    now=np.zeros(ncu)
    # each CU independent except the shared queue -> event simulation
    # compute next finish per CU
    def next_finish(c):
        jobs=cus[c]
        if not jobs: return None
        if prio:
            j=max(range(len(jobs)), key=lambda i: jobs[i])  # heaviest remaining gets the pipe
            return jobs[j], j
        j=min(range(len(jobs)), key=lambda i: jobs[i])
        return jobs[j]*len(jobs), j
    heap=[]
    for c in range(ncu):
        nf=next_finish(c)
        if nf: heapq.heappush(heap,(nf[0],c))
    end=0
    while heap:
        tf,c=heapq.heappop(heap)
        dt=tf-now[c]
        jobs=cus[c]
        if prio:
            j=max(range(len(jobs)), key=lambda i: jobs[i]); jobs.pop(j)
        else:
            k=len(jobs); mn=min(jobs); 
            # all progress by dt/k
            d=dt/k
            jobs[:]=[x-d for x in jobs]
            j=min(range(len(jobs)), key=lambda i: jobs[i]); jobs.pop(j)
        now[c]=tf; end=max(end,tf)
        if nxt<n: jobs.append(work[nxt]+fixed); nxt+=1
        nf=next_finish(c)
        if nf: heapq.heappush(heap,(now[c]+nf[0],c))
    return end
if __name__=='__main__':
    lvl=int(sys.argv[1]); C=int(sys.argv[2]); ntw=int(sys.argv[3]) if len(sys.argv)>3 else 4
    sw=tiles(lvl)
    J=C//16; gy=C//(16*ntw); W=4
    # per-wave exclusive cycles of a WG: strip-offsets * J slabs... each strip-offset-slab = ntw*4 MFMAs*32cyc, split over W waves
    cyc=sw*J*ntw*4*32/W
    order=np.argsort(-cyc)
    work=np.repeat(cyc[order],gy)  # column groups consecutive
    ideal=work.sum()/256
    for slots in (2,4,5,8):
        for fixed in (0,3000):
            print(lvl,C,'slots',slots,'fixed',fixed,'ideal us',round(ideal/2100,1),'PS',round(sim(list(work),slots=slots,fixed=fixed)/2100,1),'prio',round(sim(list(work),slots=slots,prio=True,fixed=fixed)/2100,1))
