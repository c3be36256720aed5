#!/usr/bin/env python3
"""Cook-Toom construction of the Winograd matrices the kernels use (F(4,3): points 0, +-1, +-2, inf; F(2,4): 0, +-1, 2, inf), in exact
rational arithmetic, each checked against the direct correlation on random integer data.  `python tools/wino_matrices.py`."""
from fractions import Fraction as Fr
import itertools
def polymul(a,b):
    r=[Fr(0)]*(len(a)+len(b)-1)
    for i,x in enumerate(a):
        for j,y in enumerate(b): r[i+j]+=x*y
    return r
def cook_toom(m, r, pts):
    n=m+r-1; assert len(pts)==n-1
    # finite points pts + infinity
    f=[]
    for i,a in enumerate(pts):
        v=Fr(1)
        for j,b in enumerate(pts):
            if i!=j: v*= (a-b)
        f.append(v)
    AT=[[ (pts[i]**j if i<n-1 else (Fr(1) if j==m-1 else Fr(0))) for i in range(n)] for j in range(m)]
    G=[[ pts[i]**k / f[i] for k in range(r)] for i in range(n-1)] + [[Fr(0)]*(r-1)+[Fr(1)]]
    BT=[]
    for i in range(n-1):
        poly=[Fr(1)]
        for j,b in enumerate(pts):
            if j!=i: poly=polymul(poly,[-b,Fr(1)])
        # degree n-2 -> pad to n coeffs
        BT.append(poly+[Fr(0)]*(n-len(poly)))
    poly=[Fr(1)]
    for b in pts: poly=polymul(poly,[-b,Fr(1)])
    BT.append(poly)
    return AT,G,BT
def check(m,r,AT,G,BT):
    n=m+r-1
    import random
    for _ in range(20):
        g=[Fr(random.randint(-5,5)) for _ in range(r)]
        d=[Fr(random.randint(-5,5)) for _ in range(n)]
        U=[sum(G[i][k]*g[k] for k in range(r)) for i in range(n)]
        V=[sum(BT[i][k]*d[k] for k in range(n)) for i in range(n)]
        M=[U[i]*V[i] for i in range(n)]
        y=[sum(AT[j][i]*M[i] for i in range(n)) for j in range(m)]
        ref=[sum(g[k]*d[j+k] for k in range(r)) for j in range(m)]
        assert y==ref,(y,ref)
    return True
for (m,r,pts) in ((4,3,[0,1,-1,2,-2]),(2,4,[0,1,-1,2]),(2,4,[0,1,-1,Fr(1,2)])):
    pts=[Fr(p) for p in pts]
    AT,G,BT=cook_toom(m,r,pts)
    print("F(%d,%d) pts"%(m,r),pts, check(m,r,AT,G,BT))
    print(" AT",[[str(x) for x in row] for row in AT])
    print(" G ",[[str(x) for x in row] for row in G])
    print(" BT",[[str(x) for x in row] for row in BT])
