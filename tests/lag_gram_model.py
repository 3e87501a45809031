"""numpy model of the lag-product Gram decomposition used by k_gram (DESIGN.md "k_gram"):

  T(u,v) = sum_{p in I} X(p+u) X(p+v),  X = replicate-padded image, u,v in {-1,0,1}^2
         = sum_{q in I+u} X(q) X(q+delta),  delta = v-u made lexicographically >= 0 by swapping u,v
         = M[lag(delta)]                      (main: q in Core, unclamped)
         + sum_{q in (I+u) minus Core} X(q) X(q+delta)   (border frame)

Core = {1 <= r <= R-3, 2 <= c <= C-3}: inside every shifted rectangle I+u, and every q+delta stays
inside the image for all 13 lags (no clamping in the main sum).
Used by tests to check the term table and the frame enumeration against the direct Gram."""
import numpy as np

DR = [-1, -1, -1, 0, 0, 1, 1, 1]
DC = [-1, 0, 1, -1, 1, -1, 0, 1]
LAGS = [(0, 0), (0, 1), (0, 2)] + [(1, b) for b in range(-2, 3)] + [(2, b) for b in range(-2, 3)]


def terms():
    """44 terms: (ur, uc, lag index) in the order 36 upper-triangle Rx entries, then 8 rx entries"""
    out = []
    pairs = [(i, j) for i in range(8) for j in range(i, 8)]
    for t in range(44):
        if t < 36:
            i, j = pairs[t]
            u, v = (DR[i], DC[i]), (DR[j], DC[j])
        else:
            i = t - 36
            u, v = (DR[i], DC[i]), (0, 0)
        d = (v[0] - u[0], v[1] - u[1])
        if d[0] < 0 or (d[0] == 0 and d[1] < 0):
            u, v = v, u
            d = (-d[0], -d[1])
        out.append((u[0], u[1], LAGS.index(d)))
    return out


def X(x, r, c):
    R, C = x.shape
    return float(x[min(max(r, 0), R - 1), min(max(c, 0), C - 1)])


def gram_by_lags(x):
    x = x.astype(np.float64)
    R, C = x.shape
    core_empty = R < 4 or C < 5
    M = np.zeros(13)
    if not core_empty:
        for l, (a, b) in enumerate(LAGS):
            M[l] = np.sum(x[1:R - 2, 2:C - 2] * x[1 + a:a + R - 2, 2 + b:C - 2 + b])
    B = np.zeros(44)
    T = terms()

    def in_core(r, c):
        return (not core_empty) and 1 <= r <= R - 3 and 2 <= c <= C - 3

    for r in range(-1, R + 1):
        for c in range(-1, C + 1):
            if in_core(r, c):
                continue
            xq = X(x, r, c)
            prods = [xq * X(x, r + a, c + b) for (a, b) in LAGS]
            for t, (ur, uc, l) in enumerate(T):
                if ur <= r <= R - 1 + ur and uc <= c <= C - 1 + uc:
                    B[t] += prods[l]
    tot = np.array([M[T[t][2]] + B[t] for t in range(44)])
    Rx = np.zeros((8, 8))
    k = 0
    for i in range(8):
        for j in range(i, 8):
            Rx[i, j] = Rx[j, i] = tot[k]
            k += 1
    return Rx, tot[36:]
