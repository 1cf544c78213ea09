#!/usr/bin/env python3
"""Rigorous bound on |c_fast - c_pocketfft| for the coefficients of an 8x8 block of pixels in [0, 255] - the constant
behind svs::TIE_SLOPE in csrc/svs_block.hpp (FAST extraction at n >= 8 redoes a block with the pocketfft-identical
transform when a quantiser input lies this close to a rounding tie).

Method (standard forward error analysis of a linear algorithm; Higham, "Accuracy and Stability", ch. 3):
both transforms are linear in the 64 pixels p_j >= 0.  Every floating-point operation satisfies
fl(a op b) = (a op b)(1 + d), |d| <= u = 2^-24 (an FMA: one such factor for a*b + c), and every stored constant is
C(1 + e), |e| <= u.  So the computed coefficient is  sum_paths  w_path * prod(1 + d_i) * p_j  over the paths of the
data-flow graph from pixel j to the output, and

    |c_computed - c_exact|  <=  gamma_m * sum_j A_j p_j  <=  gamma_m * max_j(A_j) * sum_j p_j,

where A_j = sum over the paths from p_j of |w_path| (the algorithm run with |constants| and with subtractions turned
into additions), m = the largest number of (1 + d) factors on any path, gamma_m = m u / (1 - m u).  sum_j p_j = 8 * c_00
(the DC coefficient the kernel has anyway).  This script runs both algorithms - svs::forward_rows<U> (FAST) and the
pocketfft N = 8 DCT-II restated in svs::pf - on a number type that tracks (A, m) and prints the resulting slope

    |c_fast - c_pf| <= (gamma_mf * Af + gamma_mp * Ap) * 8 * c_00   for every coefficient index and every U.

Operations on small integers (the packed 16-bit stages, ubyte -> float conversions, multiplications by 2) are exact and
add no factor.
"""
import numpy as np

U = 2.0 ** -24


class T:
    """tracked value: A = |path weights| per pixel (64), m = max rounding factors on a path"""
    __slots__ = ("A", "m")

    def __init__(self, A, m=0):
        self.A, self.m = A, m

    @staticmethod
    def pixel(j):
        a = np.zeros(64)
        a[j] = 1.0
        return T(a, 0)

    def add(self, o, exact=False):           # a + b or a - b
        return T(self.A + o.A, max(self.m, o.m) + (0 if exact else 1))

    def mulc(self, c, exact_const=False):    # a * constant
        if exact_const:                      # power of two: exact
            return T(abs(c) * self.A, self.m)
        return T(abs(c) * self.A, self.m + 2)   # constant's representation error + the rounding

    def fma(self, c, o):                     # a * c + o, one rounding
        return T(abs(c) * self.A + o.A, max(self.m + 1, o.m) + 1)


A0 = 0.35355339059327373
C = [None, 0.49039264020161522, 0.46193976625564337, 0.41573480615127262, 0.35355339059327373,
     0.27778511650980114, 0.19134171618254492, 0.09754516100806417]


def fdct8(x, nout):
    """svs::fdct8<NOUT>"""
    s = [x[i].add(x[7 - i]) for i in range(4)]
    d = [x[i].add(x[7 - i]) for i in range(4)]
    t0, t1, t2, t3 = s[0].add(s[3]), s[1].add(s[2]), s[0].add(s[3]), s[1].add(s[2])
    X = [None] * 8
    X[0] = t0.add(t1).mulc(A0)
    odd = lambda a, b, c_, e: d[0].fma(a, d[1].fma(b, d[2].fma(c_, d[3].mulc(e))))
    X[1] = odd(C[1], C[3], C[5], C[7])
    X[2] = t2.fma(C[2], t3.mulc(C[6]))
    X[3] = odd(C[3], C[7], C[1], C[5])
    X[4] = t0.add(t1).mulc(C[4])
    X[5] = odd(C[5], C[1], C[7], C[3])
    X[6] = t2.fma(C[6], t3.mulc(C[2]))
    X[7] = odd(C[7], C[5], C[3], C[1])
    return X[:nout] + [None] * (8 - nout)


def fast_forward(u_rows):
    """svs::forward_rows<U>: D[u][v] for u < U"""
    px = [[T.pixel(8 * y + x) for x in range(8)] for y in range(8)]
    V = [[None] * 8 for _ in range(u_rows)]
    for x in range(8):
        col = [px[y][x] for y in range(8)]
        if u_rows <= 2:                       # packed 16-bit vertical pass: exact integer sums / differences
            tot = col[0]
            for y in range(1, 8):
                tot = tot.add(col[y], exact=True)
            V[0][x] = tot.mulc(A0)
            if u_rows == 2:
                d = [col[k].add(col[7 - k], exact=True) for k in range(4)]
                V[1][x] = d[0].fma(C[1], d[1].fma(C[3], d[2].fma(C[5], d[3].mulc(C[7]))))
        else:                                 # float columns (exact ubyte -> float), fdct8<U>
            out = fdct8(col, u_rows)
            for u in range(u_rows):
                V[u][x] = out[u]
    return [fdct8(V[u], 8) for u in range(u_rows)]


TW = [np.cos((i + 1) * np.pi / 16) for i in range(7)]
W = np.cos(np.pi / 4)


def pf_dct2(x):
    """svs::pf::dct2_8 (pocketfft T_dcst23<float>::exec, N = 8, with the exactness-preserving rewrites of svs_block.hpp)"""
    c = [None] * 8
    c[0] = x[0].mulc(2.0, True)
    c[7] = x[7].mulc(2.0, True)
    for k in (1, 3, 5):
        c[k + 1] = x[k + 1].add(x[k])
        c[k] = x[k].add(x[k + 1])
    # rfft8_backward
    h0, h4 = c[0].add(c[7]), c[0].add(c[7])
    h1, tr2 = c[1].add(c[5]), c[1].add(c[5])
    ti2, h2 = c[2].add(c[6]), c[2].add(c[6])
    h6 = ti2.mulc(W).add(tr2.mulc(W))
    h5 = tr2.mulc(W).add(ti2.mulc(W))
    r = [None] * 8

    def f2(a, b):                                       # fma(+-2, a, b): the product by 2 is exact, one rounding
        t = a.fma(2.0, b)
        t.m = max(a.m, b.m) + 1
        return t
    a0, b0 = f2(c[3], h0), f2(c[3], h0)
    r[0], r[4], r[6], r[2] = f2(h1, a0), f2(h1, a0), f2(h2, b0), f2(h2, b0)
    a1, b1 = f2(c[4], h4), f2(c[4], h4)
    r[1], r[5], r[7], r[3] = f2(h5, a1), f2(h5, a1), f2(h6, b1), f2(h6, b1)
    X = [None] * 8
    for (i, j, ta, tb) in ((1, 7, TW[0], TW[6]), (2, 6, TW[1], TW[5]), (3, 5, TW[2], TW[4])):
        t1 = r[j].mulc(ta * 0.125).add(r[i].mulc(tb * 0.125))
        t2 = r[i].mulc(ta * 0.125).add(r[j].mulc(tb * 0.125))
        X[i], X[j] = t1.add(t2), t1.add(t2)
    X[4] = r[4].mulc(TW[3] * 0.25)
    X[0] = r[0].mulc(np.sqrt(2) * 0.125)
    return X


def pf_forward():
    px = [[T.pixel(8 * y + x) for x in range(8)] for y in range(8)]
    V = [[None] * 8 for _ in range(8)]
    for x in range(8):
        out = pf_dct2([px[y][x] for y in range(8)])
        for u in range(8):
            V[u][x] = out[u]
    return [pf_dct2(V[u]) for u in range(8)]


def gamma(m):
    return m * U / (1 - m * U)


def main():
    pf = pf_forward()
    worst = 0.0
    for u_rows in range(1, 9):
        fast = fast_forward(u_rows)
        row_worst = 0.0
        for k in range(1, 8 * u_rows):
            u, v = divmod(k, 8)
            f, p = fast[u][v], pf[u][v]
            slope = (gamma(f.m) * f.A.max() + gamma(p.m) * p.A.max()) * 8.0
            row_worst = max(row_worst, slope)
        print(f"U = {u_rows}: |c_fast - c_pf| <= {row_worst:.3e} * c00  = {row_worst / U:.1f} u * c00")
        worst = max(worst, row_worst)
    print(f"all U: slope {worst:.3e} = {worst / U:.1f} u;  at c00 = 1024 (mid-gray): {worst * 1024:.2e}")
    return worst


if __name__ == "__main__":
    main()
