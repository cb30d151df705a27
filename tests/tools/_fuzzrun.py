"""ad-hoc: compare emu vs oracle on random sessions"""
import sys, random, time; sys.path.insert(0,'..')
import _oracle, _emu, _fuzz
from loro_amd import wire
def main(n=40, base=0, steps=40, **kw):
    docs=[]
    for seed in range(base, base+n):
        kinds=[("text",),("text","list"),("text","list","map"),("map",)][seed%4]
        reps=_fuzz.random_session(seed, n_peers=2+seed%3, n_steps=steps+seed%50, kinds=kinds, **kw)
        docs.append(_fuzz.blobs_of(reps, random.Random(seed)))
    t0=time.time(); o=_oracle.merge_batch(docs); t1=time.time(); e=_emu.merge_batch(docs); t2=time.time()
    bad=[i for i,(x,y) in enumerate(zip(o,e)) if x!=y]
    print('oracle %.2fs emu %.2fs bad %d/%d'%(t1-t0,t2-t1,len(bad),n), bad[:10])
    for i in bad[:3]:
        print(i, o[i]); print(i, e[i])
    return bad
if __name__=='__main__':
    main(int(sys.argv[1]) if len(sys.argv)>1 else 40, int(sys.argv[2]) if len(sys.argv)>2 else 0, int(sys.argv[3]) if len(sys.argv)>3 else 40)
