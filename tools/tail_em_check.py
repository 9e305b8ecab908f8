"""The Euler-Maclaurin block sums of k_tail_nu (cbs.hip) against term-by-term sums (math.fsum) over the range of arguments and block sizes a WGS sample meets: prints the\nlargest absolute difference of a block (the series itself is ~10).  CPU only: python tools/tail_em_check.py"""
import numpy as np, math
from scipy.special import erfc
from numpy.polynomial.legendre import leggauss
gx, gw = leggauss(16)
def f(t, a): return erfc(a*np.sqrt(t))/t
def fp(t, a): return -erfc(a*np.sqrt(t))/t**2 - a*np.exp(-a*a*t)/(math.sqrt(math.pi)*t**1.5)
def block_exact(D, a):
    d = np.arange(D+1, 2*D+1, dtype=np.float64)
    return math.fsum((erfc(a*np.sqrt(d))/d).tolist())
def block_em(D, a, nsub=16):
    t1, t2 = D+0.5, 2*D+0.5
    v1, v2 = math.log(a*math.sqrt(t1)), math.log(a*math.sqrt(t2))
    h = (v2-v1)/nsub; I = 0.0
    for s in range(nsub):
        c = v1 + (s+0.5)*h
        v = c + 0.5*h*gx
        I += 0.5*h*np.sum(gw*2*erfc(np.exp(v)))
    return I - (fp(t2,a)-fp(t1,a))/24.0
worst=0
for x in [0.0101,0.012,0.02,0.05,0.1,0.3,0.7,1.0,2.0,3.5,6.0]:
    a = x/(2*math.sqrt(2))
    for j in range(9, 21):
        D = 2**j
        if D > 2**20: break
        ex = block_exact(D,a); em = block_em(D,a)
        err = abs(em-ex)
        worst=max(worst,err)
        if j in (9,10,14,20): print("x %.4f D 2^%d exact %.15e em %.15e abs err %.2e" % (x,j,ex,em,err))
print("worst abs err", worst)
