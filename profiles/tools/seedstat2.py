import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from arcs_amd import synth
t0=time.time()
sites=[]
contigs = synth.make_draft(100_000_000, repeats="human", repeat_sites=sites)
lens = np.array([len(c) for c in contigs]); cstart = np.concatenate([[0], np.cumsum(lens)])
g = np.concatenate(contigs)
lut = np.full(256, 4, np.uint8); lut[ord('A')]=0; lut[ord('C')]=1; lut[ord('G')]=2; lut[ord('T')]=3
c = lut[g]
N = len(c); M=21
def canon(mat):  # mat: (n, M) codes 0..3 -> canonical code
    f = np.zeros(len(mat), np.uint64); r = np.zeros(len(mat), np.uint64)
    for j in range(M):
        xx = mat[:,j].astype(np.uint64)
        f |= xx << np.uint64(2*(M-1-j)); r |= (np.uint64(3)-xx) << np.uint64(2*j)
    return np.minimum(f,r)
valid = np.ones(N-M+1, bool)
f = np.zeros(N-M+1, np.uint64); r = np.zeros(N-M+1, np.uint64)
for j in range(M):
    x = c[j:N-M+1+j]; valid &= x<4
    xx = (x&3).astype(np.uint64)
    f |= xx << np.uint64(2*(M-1-j)); r |= (np.uint64(3)-xx) << np.uint64(2*j)
can = np.minimum(f, r); del f, r
inidx = np.zeros(N, bool)
for s,L in zip(cstart[:-1], lens):
    if L>=500:
        e=min(30000,L); inidx[s:s+e]=True; inidx[s+L-e:s+L]=True
inidx = inidx[:N-M+1] & valid
u, cnt = np.unique(can[inidx], return_counts=True)
print("unique", time.time()-t0, len(u))
def lookup(codes):
    ii = np.searchsorted(u, codes); ii[ii>=len(u)] = 0
    hit = u[ii]==codes
    return np.where(hit, cnt[ii], 0)
rng = np.random.default_rng(5)
k=60
for RL in (128,151):
    n=600000
    p = rng.integers(0, N-RL-1, size=n)
    ok = inidx[p] & inidx[p+RL-M]
    p = p[ok]
    nwin = RL-k+1; w=k-M+1
    G=(nwin+w-1)//w
    seedpos=[min((gi+1)*w-1, nwin-1) for gi in range(G)]
    reads = c[p[:,None] + np.arange(RL)[None,:]].copy()
    good = (reads<4).all(1); reads=reads[good]; p=p[good]
    err = rng.random(reads.shape) < 0.005
    newb = rng.integers(0,4,size=reads.shape).astype(np.uint8)
    err &= newb!=reads
    reads[err]=newb[err]
    sc = np.stack([lookup(canon(reads[:,o:o+M])) for o in seedpos],1)
    flagged = (sc>2).any(1)
    print(RL, "reads", len(p), "flagged", flagged.mean())
    fr = reads[flagged]; fe = err[flagged]
    nerr = fe.sum(1)
    print("  flagged reads with 0/1/2+ errors:", (nerr==0).mean(), (nerr==1).mean(), (nerr>=2).mean())
    one = nerr>=1
    fr1 = fr[one]; e1 = fe[one].argmax(1)   # first error
    for name, off in (("e-20", np.clip(e1-20,0,RL-M)), ("e", np.clip(e1,0,RL-M)), ("e-10", np.clip(e1-10,0,RL-M))):
        mm = fr1[np.arange(len(fr1))[:,None], off[:,None]+np.arange(M)[None,:]]
        cc = lookup(canon(mm))
        print("   m-mer at", name, ": zero entries", (cc==0).mean(), " 1-2:", ((cc>=1)&(cc<=2)).mean(), " 3-8:", ((cc>=3)&(cc<=8)).mean(), " heavy:", (cc>8).mean())
    # both e-20 and e zero
    mm1 = fr1[np.arange(len(fr1))[:,None], np.clip(e1-20,0,RL-M)[:,None]+np.arange(M)[None,:]]
    mm2 = fr1[np.arange(len(fr1))[:,None], np.clip(e1,0,RL-M)[:,None]+np.arange(M)[None,:]]
    z = (lookup(canon(mm1))==0)&(lookup(canon(mm2))==0)
    print("   both zero:", z.mean())
