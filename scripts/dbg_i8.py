import sys; sys.path.insert(0,'.')
import numpy as np
from semcode_amd import _native
from oracle import sc_oracle as orc
rt=_native.Runtime(0)
def run(X,Q,tag,metric="L2",k=10):
    ix=_native.Index(rt,X.shape[1],metric=metric); ix.add(X); ix.set_search_mode("batched")
    od,orow=orc.search(X,Q,k,metric)
    for st in (16,8):
        ix.set_coarse_stage(st); d,r=ix.search(Q,k=k)
        bad=(r!=orow).any(1)
        print(tag,"stage",st, ix.last_search_stats(), "wrong queries", int(bad.sum()), "of", len(Q))
    ix.close()
run(orc.synth(20000,128,seed=1),orc.synth(40,128,seed=2),"gauss128")
run(orc.synth(20000,128,seed=1),orc.synth(40,128,seed=2),"gauss128 IP",metric="IP")
run(orc.synth(100000,768,seed=1),orc.synth(200,768,seed=2),"gauss768 COS",metric="COSINE")
