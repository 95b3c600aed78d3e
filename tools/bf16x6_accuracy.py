#!/usr/bin/env python3
"""Accuracy of the piece-product forms against float64 on a conv-shaped sum (CPU, torch): bf16 three-piece split with 3 / 6 / 9
products and the fp16 two-piece split with 3 products, next to the fp32 conv itself.  DESIGN.md section 6d quotes these figures."""
import torch, math
torch.manual_seed(0)
torch.set_num_threads(8)
def split3(a):
    a0=a.bfloat16().float(); r=a-a0; a1=r.bfloat16().float(); r2=r-a1; a2=r2.bfloat16().float()
    return a0,a1,a2
C=256;T=600;B=2
x=torch.randn(B,C,T); w=torch.randn(C,C,7)/math.sqrt(C*7)
ref=torch.nn.functional.conv1d(x.double(),w.double(),padding=3)
y32=torch.nn.functional.conv1d(x,w,padding=3)
xs=split3(x); ws=split3(w)
def conv(a,b): return torch.nn.functional.conv1d(a,b,padding=3)
pairs6=[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]
pairs3=[(0,0),(0,1),(1,0)]
pairs9=[(i,j) for i in range(3) for j in range(3)]
for name,pairs in (("x3",pairs3),("x6",pairs6),("x9",pairs9)):
    # accumulate small terms first is not what MFMA would do; emulate big-first in fp32
    acc=torch.zeros_like(y32)
    for (i,j) in pairs: acc=acc+conv(xs[i],ws[j])
    # and exact sum of terms in fp64 (isolates the truncation error from accumulate error)
    acc64=sum(torch.nn.functional.conv1d(xs[i].double(),ws[j].double(),padding=3) for (i,j) in pairs)
    e=(acc.double()-ref); e64=(acc64-ref)
    print(name,"rms err fp32-acc %.3e  truncation-only %.3e"%(e.pow(2).mean().sqrt(), e64.pow(2).mean().sqrt()))
e=(y32.double()-ref); print("fp32 conv rms err %.3e   rms(y)=%.3f"%(e.pow(2).mean().sqrt(), ref.pow(2).mean().sqrt()))

def split2h(a):
    am=a.abs().amax(); S=2.0**(13-int(torch.floor(torch.log2(am))))
    h0=(a*S).half().float(); h1=(a*S-h0).half().float(); return h0,h1,S
xh=split2h(x); wh=split2h(w)
acc=(conv(xh[0],wh[1])+conv(xh[1],wh[0])+conv(xh[0],wh[0]))/(xh[2]*wh[2])
acc64=sum(torch.nn.functional.conv1d(a.double(),b.double(),padding=3) for a,b in ((xh[0],wh[1]),(xh[1],wh[0]),(xh[0],wh[0])))/(xh[2]*wh[2])
print("f16x3 rms err fp32-acc %.3e  truncation-only %.3e"%((acc.double()-ref).pow(2).mean().sqrt(),(acc64-ref).pow(2).mean().sqrt()))
