"""Enumerates the LDS bank conflicts of the NTT pass kernels' access patterns (csrc/ntt.hip): for every tile shape and phase
(load, each radix-4 step, the odd level, the bit-reversed store) the worst number of distinct addresses per bank within a 32-lane
half-wave (what one ds_*_b32 cycle serves from 64 banks), for the plain row-major image and for XOR swizzles of the row index;
prints the best swizzle per shape.  CPU only.  This is where `at()` in ntt.hip comes from."""
import itertools
def bitrev(x,b): return int(format(x,'0%db'%b)[::-1],2) if b else 0
def conflicts(addrs):
    # addrs: list of 64 word addresses (one wave); ds_*_b32 serves the two 32-lane halves separately
    worst=1
    for h in (addrs[:32],addrs[32:]):
        banks={}
        for a in h: banks.setdefault(a%64,set()).add(a)
        worst=max(worst,max(len(v) for v in banks.values()))
    return worst
def analyse(DEG,LOG_T,TH,hashf):
    D,T=1<<DEG,1<<LOG_T; half=D>>1; E=D*T
    at=lambda row,c:(hashf(row)<<LOG_T)+c
    res={}
    # load phase
    w=[at(e>>LOG_T,e&(T-1)) for e in range(64)]
    res['load']=conflicts(w)
    for st in range(DEG//2):
        rnd=2*st; bit=half>>rnd; hb=bit>>1
        worst=1
        for base in range(0,(half>>1)*T,64):
            for off in (0,hb,bit,bit+hb):
                ad=[]
                for w_ in range(base,base+64):
                    c=w_&(T-1); wg=w_>>LOG_T; dj=wg>>rnd
                    r0=(((wg&((1<<rnd)-1))*bit)<<1)+dj
                    ad.append(at(r0+off,c))
                worst=max(worst,conflicts(ad))
        res['r4_%d'%st]=worst
    if DEG&1:
        rnd=DEG-1; bit=half>>rnd; worst=1
        for base in range(0,half*T,64):
            for off in (0,bit):
                ad=[]
                for w_ in range(base,base+64):
                    c=w_&(T-1); wb=w_>>LOG_T; di=wb>>rnd
                    b=((wb&((1<<rnd)-1))*bit)|di; i0=(b<<1)-di
                    ad.append(at(i0+off,c))
                worst=max(worst,conflicts(ad))
        res['odd']=worst
    worst=1
    for base in range(0,E,64):
        ad=[at(bitrev(e>>LOG_T,DEG),e&(T-1)) for e in range(base,base+64)]
        worst=max(worst,conflicts(ad))
    res['store']=worst
    return res
for DEG,LOG_T in ((9,2),(8,3),(7,4),(6,4),(5,5),(4,6)):
    G=max(1,64>>LOG_T); 
    print(DEG,LOG_T,'identity',analyse(DEG,LOG_T,0,lambda r:r))
    best=None
    for s1,s2,s3 in itertools.product(range(1,9),range(1,10),range(0,10)):
        if s2<=s1 or (s3 and s3<=s2): continue
        f=lambda r,s1=s1,s2=s2,s3=s3:r^(((r>>s1)^(r>>s2)^((r>>s3) if s3 else 0))&(G-1))
        res=analyse(DEG,LOG_T,0,f); score=(max(res.values()),sum(res.values()))
        if best is None or score<best[0]: best=(score,(s1,s2,s3),res)
    print('   best',best)
print("---- candidates")
for cand in ((3,5,0),(1,2,5),(1,2,4),(2,4,0)):
    for DEG,LOG_T in ((9,2),(8,3),(7,4),(6,4),(5,5),(4,6)):
        G=max(1,64>>LOG_T); s1,s2,s3=cand
        f=lambda r:r^(((r>>s1)^(r>>s2)^((r>>s3) if s3 else 0))&(G-1))
        print(cand,(DEG,LOG_T),analyse(DEG,LOG_T,0,f))
