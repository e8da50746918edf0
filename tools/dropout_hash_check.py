"""Statistical check of the dropout hash of csrc/common.cuh (dropout_quad) against the first form (h_old) and other candidates:
drop rates, mask correlations between draws / neighbours / keys, uniformity.  numpy only; python tools/dropout_hash_check.py"""
import numpy as np
M32=np.uint64(0xFFFFFFFF)
def mul24(a,c):
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(c)) & M32
def rot16(k): return ((k>>np.uint64(16)) | (k<<np.uint64(16))) & M32
def h_old(idx,key):
    x=(idx^key)&M32
    x^=x>>np.uint64(16); x=(x*np.uint64(0x7feb352d))&M32; x^=rot16(key); x^=x>>np.uint64(15); x=(x*np.uint64(0x846ca68b))&M32; x^=x>>np.uint64(16)
    return x
def h_a(idx,key):
    x=(idx^key)&M32
    x^=x>>np.uint64(15)
    x=mul24(x,0xB5297B) ^ (x>>np.uint64(24))
    x^=rot16(key)
    x^=x>>np.uint64(13)
    x=mul24(x,0x8DA6B5) ^ (x>>np.uint64(24))
    x^=x>>np.uint64(16)
    return x
def h_b(idx,key):
    # two 24-bit multiplies over overlapping windows, then one more
    x=(idx^key)&M32
    p=mul24(x,0xB5297B) ^ mul24(x>>np.uint64(8),0x9E3779)
    p^=rot16(key)
    p^=p>>np.uint64(15)
    q=mul24(p,0x8DA6B5) ^ mul24(p>>np.uint64(8),0xC2B2AF)
    q^=q>>np.uint64(16)
    return q
def h_c(idx,key):
    x=(idx^key)&M32
    x^=x>>np.uint64(16)
    x=(mul24(x,0x7feb35)+ (mul24(x>>np.uint64(8),0x2d1b54)<<np.uint64(8)))&M32     # ~ 32x24 product
    x^=rot16(key)
    x^=x>>np.uint64(15)
    x=(mul24(x,0x846ca7)+ (mul24(x>>np.uint64(8),0x68b1d3)<<np.uint64(8)))&M32
    x^=x>>np.uint64(16)
    return x
def stats(h,name):
    n=1<<22
    idx=np.arange(n,dtype=np.uint64)+np.uint64(12345)
    key=np.uint64(0x9d2c5681)
    t=int(0.0635*65536+0.5)
    out=[]
    v=h(idx,key)
    lo=(v&np.uint64(0xFFFF)).astype(np.int64); hi=(v>>np.uint64(16)).astype(np.int64)
    dl=(lo<t).astype(np.float64); dh=(hi<t).astype(np.float64)
    def corr(a,b): return float(np.corrcoef(a,b)[0,1])
    res=dict(rate_lo=dl.mean(), rate_hi=dh.mean(), lag1=corr(dl[:-1],dl[1:]), lag256=corr(dl[:-256],dl[256:]), halves=corr(dl,dh),
             lag1_hi_lo=corr(dh[:-1],dl[1:]))
    worst=0
    for bit in range(32):
        v2=h(idx,key^np.uint64(1<<bit))
        d2=((v2&np.uint64(0xFFFF)).astype(np.int64)<t).astype(np.float64)
        worst=max(worst,abs(corr(dl,d2)))
        d3=((v2>>np.uint64(16)).astype(np.int64)<t).astype(np.float64)
        worst=max(worst,abs(corr(dh,d3)))
    res['key_bit_worst']=worst
    # value correlation (full 16-bit) lag1
    res['val_lag1']=corr(lo[:-1].astype(float),lo[1:].astype(float))
    res['val_halves']=corr(lo.astype(float),hi.astype(float))
    # uniformity chi2 over 256 bins of lo
    cnt=np.bincount(lo>>8,minlength=256); exp=n/256
    res['chi2_256']=float(((cnt-exp)**2/exp).sum())
    # 2-D structure: rows of 256 pairs; column-wise drop rates
    m=dl.reshape(-1,256); res['col_rate_sd']=float(m.mean(0).std()); res['row_rate_sd']=float(m.mean(1).std())
    print(name, {k:round(v,5) for k,v in res.items()})
for f,n in ((h_old,'old'),(h_a,'A'),(h_b,'B'),(h_c,'C')): stats(f,n)
print("---- quad scheme")
def quad(idx,key,C1=0xB5297B,C2=0x8DA6B5,C3=0x3C6EF3):
    x=(idx^key)&M32
    x^=x>>np.uint64(15)
    x=mul24(x,C1)
    x^=rot16(key)
    y=x^(x>>np.uint64(13))
    q1=mul24(y,C2); q1^=q1>>np.uint64(16)
    q2=mul24(y,C3); q2^=q2>>np.uint64(16)
    return q1,q2
n=1<<22
idx=np.arange(n,dtype=np.uint64)+np.uint64(777)
key=np.uint64(0x9d2c5681)
t=int(0.0635*65536+0.5)
q1,q2=quad(idx,key)
d=[((q1&np.uint64(0xFFFF)).astype(np.int64)<t),((q1>>np.uint64(16)).astype(np.int64)<t),((q2&np.uint64(0xFFFF)).astype(np.int64)<t),((q2>>np.uint64(16)).astype(np.int64)<t)]
d=[x.astype(float) for x in d]
print("rates",[round(x.mean(),5) for x in d])
def corr(a,b): return float(np.corrcoef(a,b)[0,1])
print("cross",[round(corr(d[i],d[j]),5) for i in range(4) for j in range(i+1,4)])
print("lag1",[round(corr(d[i][:-1],d[j][1:]),5) for i in range(4) for j in range(4)])
print("lag128",[round(corr(d[i][:-128],d[i][128:]),5) for i in range(4)])
worst=0
for bit in range(32):
    a,b=quad(idx,key^np.uint64(1<<bit))
    e=[((a&np.uint64(0xFFFF)).astype(np.int64)<t),((a>>np.uint64(16)).astype(np.int64)<t),((b&np.uint64(0xFFFF)).astype(np.int64)<t),((b>>np.uint64(16)).astype(np.int64)<t)]
    for i in range(4):
        worst=max(worst,abs(corr(d[i],e[i].astype(float))))
print("key bit worst",round(worst,5))
vals=[(q1&np.uint64(0xFFFF)).astype(np.int64),(q1>>np.uint64(16)).astype(np.int64),(q2&np.uint64(0xFFFF)).astype(np.int64),(q2>>np.uint64(16)).astype(np.int64)]
for v in vals:
    cnt=np.bincount(v>>8,minlength=256); print("chi2",round(float(((cnt-n/256)**2/(n/256)).sum()),1), end=" ")
print()
# several thresholds
for p in (0.01,0.1,0.3817,0.5):
    tt=int(p*65536+0.5); print(p,[round(float((v<tt).mean()),5) for v in vals])
# many keys: rate stability
rs=[]
for k in range(20):
    a,b=quad(idx[:1<<20],np.uint64((k*2654435761+12345)&0xFFFFFFFF))
    rs.append(float(((a&np.uint64(0xFFFF)).astype(np.int64)<t).mean()))
print("rate over keys: mean %.5f sd %.5f"%(np.mean(rs),np.std(rs)))
