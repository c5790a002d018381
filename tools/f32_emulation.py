"""Float32 emulation of the forward path with numpy (pocketfft in complex64, float32 windows):
the accuracy floor of ANY complex64 implementation of the reference algorithm.  Prints
relative RMSE against the complex128 oracle.  CPU only."""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import swiftly_oracle as orc
# float32 emulation of the forward path: every stage result rounded to c64 and FFTs done by numpy in c64
def c64(a): return a.astype(np.complex64)
class F32Core(orc.OracleCore):
    pass
def fwd(core, facet_items, facets, sg_items, f32):
    cast = (lambda a: a.astype(np.complex64)) if f32 else (lambda a:a)
    # monkeypatch: numpy fft preserves c64 in numpy>=2
    BF=[cast(core.prepare_facet(cast(d) if f32 else d, f.off0, 0)) for f,d in zip(facet_items,facets)]
    out=[]
    for sg in sg_items:
        cols=[cast(core.prepare_facet(cast(core.extract_from_facet(b, sg.off0,0)), f.off1,1)) for f,b in zip(facet_items,BF)]
        contribs=[core.extract_from_facet(c, sg.off1,1) for c in cols]
        acc=None
        for off1 in sorted({f.off1 for f in facet_items}):
            col=None
            for f,c in zip(facet_items,contribs):
                if f.off1==off1: col=core.add_to_subgrid(c,f.off0,0,out=col); col=cast(col)
            acc=core.add_to_subgrid(col,off1,1,out=acc); acc=cast(acc)
        r=core.finish_subgrid(acc,[sg.off0,sg.off1],sg.size)
        out.append(r)
    return out
def run(P, nf=None, ns=3):
    core=orc.OracleCore(P['W'],P['N'],P['xM_size'],P['yN_size'])
    if True:
        # make windows float32 too
        pass
    fi=orc.make_full_cover(P['N'],P['yB_size']); si=orc.make_full_cover(P['N'],P['xA_size'])
    if nf: fi=fi[:nf]
    si=si[:ns]
    yB=P['yB_size']
    facets=[]
    for j,f in enumerate(fi):
        r=np.random.default_rng(77+j); d=(r.standard_normal((yB,yB))+1j*r.standard_normal((yB,yB))).astype(np.complex64)
        facets.append((d*f.mask0[:,None]*f.mask1[None,:]).astype(complex))
    want=fwd(core,fi,facets,si,False)
    core32=orc.OracleCore(P['W'],P['N'],P['xM_size'],P['yN_size'])
    core32.pswf=core32.pswf.astype(np.float32); core32.Fn=core32.Fn.astype(np.float32)
    _fw=core32.facet_window
    core32.facet_window=lambda n: _fw(n).astype(np.float32)
    got=fwd(core32,fi,[f.astype(np.complex64) for f in facets],si,True)
    for a,b in zip(got,want):
        print(a.dtype, 'relrms', np.sqrt(np.mean(abs(a-b)**2)/np.mean(abs(b)**2)), 'max/max', abs(a-b).max()/abs(b).max())
print('small W=13.56'); run(dict(W=13.5625,N=512,yB_size=208,yN_size=256,xA_size=100,xM_size=128))
print('8k W=11'); run(dict(W=11.0,N=8192,yB_size=1408,yN_size=2048,xA_size=1024,xM_size=2048), nf=2)
