import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
from oracle import bert_oracle as bo
from semcode_amd import _native
import os
from pathlib import Path
if os.environ.get('SC_LIBVAR'): _native.LIB_PATH = Path(_native.LIB_PATH).parent / ('var_%s.so' % os.environ['SC_LIBVAR'])
rt=_native.Runtime(0)
L=int(sys.argv[1]) if len(sys.argv)>1 else 1
B=int(sys.argv[2]) if len(sys.argv)>2 else 256
NR=int(sys.argv[3]) if len(sys.argv)>3 else 6
cfg=dict(bo.BERT_BASE, layers=L)
enc=_native.Encoder(rt,cfg,weights=None,synth_seed=0)
enc.set_path("batch")
rng=np.random.default_rng(3)
ids=rng.integers(1000,30000,size=(B,256)).astype(np.int32)
lens=np.full(B,256,np.int32)
M=B*256
def read(i,sz):
    buf=np.empty(sz,np.uint8)
    _native._check(_native.lib().sc_diag_encoder_read(enc.handle,i,buf.ctypes.data_as(C.c_void_p),sz))
    return buf
def bf(buf): return (buf.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
ys=[]; fins=[]; ctxs=[]
for rep in range(NR):
    enc.embed_ids(ids,lens)
    ys.append(read(1,M*768*2).view(np.uint16).reshape(M,768).copy()); fins.append(read(8,M*8).view(np.float32).reshape(M,2).copy()); ctxs.append(read(3,M*768*2).copy())
if os.environ.get('SC_LIBVAR'):
    h=C.CDLL(str(_native.LIB_PATH)); out=np.zeros(64*16,np.float32); n=C.c_uint(0)
    h.sc_diag_resln_dump(out.ctypes.data_as(C.c_void_p), C.byref(n)); print("in-kernel mismatches:", n.value)
    np.set_printoptions(linewidth=250, precision=6, suppress=False)
    for rec in out.reshape(64,16)[:min(n.value,24)]: print("  mt %d badmask %d w %d mi %d ni %d r %d lane %d | d pk %.6g sc %.6g | p pk %.6g sc %.6g | ab pk %.6g sc %.6g | t pk %.6g sc %.6g" % (*rec[:7].astype(int), *rec[7:15]))
print("pairwise y diffs vs run0:", [int((ys[0]!=y).sum()) for y in ys], " vs run%d:"%(NR-1), [int((ys[-1]!=y).sum()) for y in ys])
print("fin equal:", [bool(np.array_equal(fins[0],f)) for f in fins], "ctx equal:", [bool(np.array_equal(ctxs[0],c)) for c in ctxs])
W=bo.unpack(cfg, bo.make_blob(cfg,0,"bench"))
names=[n for n in W if not n.startswith("l")]; print("non-layer weights:", names)
def bfr(a):
    u=np.ascontiguousarray(a,np.float32).view(np.uint32).astype(np.uint64); u=(u+0x7FFF+((u>>16)&1))&0xFFFF0000
    return u.astype(np.uint32).view(np.float32)
we=[n for n in names if "word_emb" == n][0]; pe=[n for n in names if "pos_emb" == n][0]; te=[n for n in names if "type_emb" == n][0]
xraw=bfr((W[we][ids.reshape(-1)] + W[pe][np.tile(np.arange(256),B)]) + W[te].reshape(-1,768)[0]).reshape(M,768)
mu=xraw.mean(1); var=(xraw.astype(np.float64)**2).mean(1)-mu.astype(np.float64)**2; rs=1/np.sqrt(var+1e-12)
print("fin check: max|mu-fin|", np.abs(mu-fins[0][:,0]).max(), "max rel rs", np.abs(rs/fins[0][:,1]-1).max())
wo=W["l0.wo"].astype(np.float32); ctx=bf(ctxs[0]).reshape(M,768)
# majority vote = reference run
for a in range(NR):
    for b in range(a+1,NR):
        bad=np.argwhere(ys[a]!=ys[b])
        if len(bad)==0: continue
        print(f"runs {a} vs {b}: {len(bad)} differing")
        ya=bf(ys[a].reshape(-1)).reshape(M,768); yb=bf(ys[b].reshape(-1)).reshape(M,768)
        for (r,c) in bad[:10]:
            ref0=float(ctx[r].astype(np.float64)@bfr(wo[c]).astype(np.float64))
            term=(xraw[r,c]-fins[0][r,0])*fins[0][r,1]
            rva=(ya[r,c]-ref0)/fins[0][r,1]+fins[0][r,0]; rvb=(yb[r,c]-ref0)/fins[0][r,1]+fins[0][r,0]
            print(f"  row {r} (mt {r//256} wm {(r%256)//128} mi {((r%256)%128)//16} fr {r%16}) col {c} (nt {c//256} wn {(c%256)//64} ni {((c%256)%64)//16} fq {(c%16)//4} r {c%4}): ya {ya[r,c]:.4f} yb {yb[r,c]:.4f} expect {ref0+term:.4f} (ctx.wo {ref0:.4f}) xraw {xraw[r,c]:.5f} implied rv a {rva:.5f} b {rvb:.5f}; xraw row nbrs {xraw[r,max(c-2,0):c+3]}")
        rows=np.unique(bad[:,0]); print("   mi:", np.bincount(((rows%256)%128)//16,minlength=8), "wm:", np.bincount((rows%256)//128,minlength=2), "fr:", np.bincount(rows%16,minlength=16))
        print("   col%16:", np.bincount(bad[:,1]%16,minlength=16), "ni:", np.bincount(((bad[:,1]%256)%64)//16,minlength=4), "wn:", np.bincount((bad[:,1]%256)//64,minlength=4), "nt:", np.bincount(bad[:,1]//256,minlength=3))
        print("   m-tiles:", np.unique(bad[:,0]//256)[:40])
        break
    else: continue
    break
