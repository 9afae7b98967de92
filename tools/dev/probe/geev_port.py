"""Prototype: netlib dgeev (jobvr='V') for tiny real matrices, in numpy, to learn which conventions decide eigenvalue
ORDER and eigenvector SIGN (the reference's Hessian clean-up V' diag(l+) V depends on both)."""
import numpy as np

EPS = np.finfo(float).eps          # dlamch('P') = eps*base = 2.2e-16
SAFMIN = np.finfo(float).tiny
ULP = EPS

def sign(a, b):
    return abs(a) if b >= 0 else -abs(a)   # Fortran SIGN (b = +0 -> +)

def dlarfg(n, alpha, x):
    """returns beta, tau, v (x overwritten)"""
    if n <= 1:
        return alpha, 0.0, x
    xnorm = np.sqrt((x * x).sum())
    if xnorm == 0.0:
        return alpha, 0.0, x
    beta = -sign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    x = x / (alpha - beta)
    return beta, tau, x

def dgebal(A):
    """job='B' : permutation + scaling. returns A, ilo, ihi (0-based inclusive), scale"""
    n = A.shape[0]
    A = A.copy()
    scale = np.ones(n)
    k, l = 0, n - 1
    # search for rows isolating an eigenvalue and push them down
    noconv = True
    while noconv:
        noconv = False
        for i in range(l, -1, -1):
            canswap = True
            for j in range(0, l + 1):
                if i != j and A[i, j] != 0.0:
                    canswap = False
                    break
            if canswap:
                scale[l] = i
                if i != l:
                    A[:, [i, l]] = A[:, [l, i]]
                    A[[i, l], k:] = A[[l, i], k:]
                noconv = True
                if l == 0:
                    return A, 0, 0, scale
                l -= 1
    noconv = True
    while noconv:
        noconv = False
        for j in range(k, l + 1):
            canswap = True
            for i in range(k, l + 1):
                if i != j and A[i, j] != 0.0:
                    canswap = False
                    break
            if canswap:
                scale[k] = j
                if j != k:
                    A[:l + 1, [j, k]] = A[:l + 1, [k, j]]
                    A[[j, k], k:] = A[[k, j], k:]
                noconv = True
                k += 1
    # scaling loop
    sclfac, factor = 2.0, 0.95
    sfmin1 = SAFMIN / EPS; sfmax1 = 1 / sfmin1
    sfmin2 = sfmin1 * sclfac; sfmax2 = 1 / sfmin2
    for i in range(k, l + 1):
        scale[i] = 1.0
    noconv = True
    while noconv:
        noconv = False
        for i in range(k, l + 1):
            c = np.sqrt((A[k:l + 1, i] ** 2).sum())
            r = np.sqrt((A[i, k:l + 1] ** 2).sum())
            ica = np.argmax(np.abs(A[:l + 1, i])); ca = abs(A[ica, i])
            ira = k + np.argmax(np.abs(A[i, k:])); ra = abs(A[i, ira])
            if c == 0.0 or r == 0.0:
                continue
            g = r / sclfac; f = 1.0; s = c + r
            while c < g and max(f, c, ca) < sfmax2 and min(r, g, ra) > sfmin2:
                f *= sclfac; c *= sclfac; ca *= sclfac; r /= sclfac; g /= sclfac; ra /= sclfac
            g = c / sclfac
            while g >= r and max(r, ra) < sfmax2 and min(f, c, g, ca) > sfmin2:
                f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac; r *= sclfac; ra *= sclfac
            if (c + r) >= factor * s:
                continue
            if f < 1.0 and scale[i] < 1.0 and f * scale[i] <= sfmin1:
                continue
            if f > 1.0 and scale[i] > 1.0 and scale[i] >= sfmax1 / f:
                continue
            g = 1.0 / f
            scale[i] *= f
            noconv = True
            A[i, k:] *= g
            A[:l + 1, i] *= f
    return A, k, l, scale

def dgehd2_orghr(A, ilo, ihi):
    n = A.shape[0]
    A = A.copy()
    Q = np.eye(n)
    refl = []
    for i in range(ilo, ihi):
        # reflector H(i) to annihilate A[i+2:ihi+1, i]
        alpha = A[i + 1, i]
        x = A[i + 2:ihi + 1, i].copy()
        beta, tau, v = dlarfg(ihi - i, alpha, x)
        vv = np.concatenate([[1.0], v])
        A[i + 1, i] = beta
        A[i + 2:ihi + 1, i] = 0.0
        # apply from the right to A[0:ihi+1, i+1:ihi+1]
        W = A[:ihi + 1, i + 1:ihi + 1] @ vv
        A[:ihi + 1, i + 1:ihi + 1] -= tau * np.outer(W, vv)
        # apply from the left to A[i+1:ihi+1, i+1:n]
        W = vv @ A[i + 1:ihi + 1, i + 1:]
        A[i + 1:ihi + 1, i + 1:] -= tau * np.outer(vv, W)
        refl.append((i, tau, vv))
    # Q = H(ilo) H(ilo+1) ... H(ihi-1)
    for (i, tau, vv) in reversed(refl):
        W = vv @ Q[i + 1:ihi + 1, :]
        Q[i + 1:ihi + 1, :] -= tau * np.outer(vv, W)
    return A, Q

def dlanv2(a, b, c, d):
    multpl = 4.0
    eps = EPS
    if c == 0.0:
        cs, sn = 1.0, 0.0
    elif b == 0.0:
        cs, sn = 0.0, 1.0
        a, d = d, a
        b, c = -c, 0.0
    elif (a - d) == 0.0 and sign(1.0, b) != sign(1.0, c):
        cs, sn = 1.0, 0.0
    else:
        temp = a - d
        p = 0.5 * temp
        bcmax = max(abs(b), abs(c))
        bcmis = min(abs(b), abs(c)) * sign(1.0, b) * sign(1.0, c)
        scale = max(abs(p), bcmax)
        z = (p / scale) * p + (bcmax / scale) * bcmis
        if z >= multpl * eps:
            z = p + sign(np.sqrt(scale) * np.sqrt(z), p)
            a = d + z
            d = d - (bcmax / z) * bcmis
            tau = np.hypot(c, z)
            cs = z / tau
            sn = c / tau
            b = b - c
            c = 0.0
        else:
            sigma = b + c
            tau = np.hypot(sigma, temp)
            cs = np.sqrt(0.5 * (1.0 + abs(sigma) / tau))
            sn = -(p / (tau * cs)) * sign(1.0, sigma)
            aa = a * cs + b * sn; bb = -a * sn + b * cs
            cc = c * cs + d * sn; dd = -c * sn + d * cs
            a = aa * cs + cc * sn; b = bb * cs + dd * sn
            c = -aa * sn + cc * cs; d = -bb * sn + dd * cs
            temp = 0.5 * (a + d)
            a = temp; d = temp
            if c != 0.0:
                if b != 0.0:
                    if sign(1.0, b) == sign(1.0, c):
                        sab = np.sqrt(abs(b)); sac = np.sqrt(abs(c))
                        p = sign(sab * sac, c)
                        tau = 1.0 / np.sqrt(abs(b + c))
                        a = temp + p; d = temp - p
                        b = b - c; c = 0.0
                        cs1 = sab * tau; sn1 = sac * tau
                        temp = cs * cs1 - sn * sn1
                        sn = cs * sn1 + sn * cs1
                        cs = temp
                else:
                    b = -c; c = 0.0
                    temp = cs; cs = -sn; sn = temp
    rt1r, rt2r = a, d
    if c == 0.0:
        rt1i = rt2i = 0.0
    else:
        rt1i = np.sqrt(abs(b)) * np.sqrt(abs(c)); rt2i = -rt1i
    return a, b, c, d, rt1r, rt1i, rt2r, rt2i, cs, sn

def drot(x, y, c, s):
    t = c * x + s * y
    y2 = c * y - s * x
    return t, y2

def dlahqr(H, Z, ilo, ihi, old=False):
    """wantt, wantz. 0-based inclusive ilo..ihi. returns wr, wi, info"""
    n = H.shape[0]
    wr = np.zeros(n); wi = np.zeros(n)
    if ilo == ihi:
        wr[ilo] = H[ilo, ilo]
        return wr, wi, 0
    for j in range(ilo, ihi - 2):
        H[j + 2, j] = 0.0; H[j + 3, j] = 0.0
    if ilo <= ihi - 2:
        H[ihi, ihi - 2] = 0.0
    nh = ihi - ilo + 1
    smlnum = SAFMIN * (nh / ULP)
    i1, i2 = 0, n - 1
    itmax = 30 * max(10, nh)
    kdefl = 0
    i = ihi
    while True:
        l = ilo
        if i < ilo:
            break
        converged = False
        for its in range(itmax + 1):
            k = i
            while k > l:
                if abs(H[k, k - 1]) <= smlnum:
                    break
                tst = abs(H[k - 1, k - 1]) + abs(H[k, k])
                if tst == 0.0:
                    if k - 2 >= ilo: tst += abs(H[k - 1, k - 2])
                    if k + 1 <= ihi: tst += abs(H[k + 1, k])
                if abs(H[k, k - 1]) <= ULP * tst:
                    ab = max(abs(H[k, k - 1]), abs(H[k - 1, k]))
                    ba = min(abs(H[k, k - 1]), abs(H[k - 1, k]))
                    aa = max(abs(H[k, k]), abs(H[k - 1, k - 1] - H[k, k]))
                    bb = min(abs(H[k, k]), abs(H[k - 1, k - 1] - H[k, k]))
                    s = aa + ab
                    if ba * (ab / s) <= max(smlnum, ULP * (bb * (aa / s))):
                        break
                k -= 1
            l = k
            if l > ilo:
                H[l, l - 1] = 0.0
            if l >= i - 1:
                converged = True
                break
            kdefl += 1
            if old:
                ex_top = (its == 10); ex_bot = (its == 20)
            else:
                ex_bot = (kdefl % 20 == 0); ex_top = (not ex_bot) and (kdefl % 10 == 0)
            if ex_bot:
                s = abs(H[i, i - 1]) + abs(H[i - 1, i - 2])
                h11 = 0.75 * s + H[i, i]; h12 = -0.4375 * s; h21 = s; h22 = h11
            elif ex_top:
                s = abs(H[l + 1, l]) + abs(H[l + 2, l + 1])
                h11 = 0.75 * s + H[l, l]; h12 = -0.4375 * s; h21 = s; h22 = h11
            else:
                h11 = H[i - 1, i - 1]; h21 = H[i, i - 1]; h12 = H[i - 1, i]; h22 = H[i, i]
            s = abs(h11) + abs(h12) + abs(h21) + abs(h22)
            if s == 0.0:
                rt1r = rt1i = rt2r = rt2i = 0.0
            else:
                h11 /= s; h21 /= s; h12 /= s; h22 /= s
                tr = (h11 + h22) / 2.0
                det = (h11 - tr) * (h22 - tr) - h12 * h21
                rtdisc = np.sqrt(abs(det))
                if det >= 0.0:
                    rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i
                else:
                    rt1r = tr + rtdisc; rt2r = tr - rtdisc
                    if abs(rt1r - h22) <= abs(rt2r - h22):
                        rt1r = rt1r * s; rt2r = rt1r
                    else:
                        rt2r = rt2r * s; rt1r = rt2r
                    rt1i = rt2i = 0.0
            m = i - 2
            while True:
                h21s = abs(H[m + 1, m])
                s = abs(H[m, m] - rt2r) + abs(rt2i) + h21s
                h21s = H[m + 1, m] / s
                v = np.zeros(3)
                v[0] = h21s * H[m, m + 1] + (H[m, m] - rt1r) * ((H[m, m] - rt2r) / s) - rt1i * (rt2i / s)
                v[1] = h21s * (H[m, m] + H[m + 1, m + 1] - rt1r - rt2r)
                v[2] = h21s * H[m + 2, m + 1]
                s = abs(v).sum()
                v /= s
                if m == l:
                    break
                h00 = abs(H[m - 1, m - 1]); h10 = abs(H[m, m - 1]); h11_ = abs(H[m, m])
                if abs(H[m, m - 1]) * (abs(v[1]) + abs(v[2])) <= ULP * abs(v[0]) * (abs(H[m - 1, m - 1]) + abs(H[m, m]) + abs(H[m + 1, m + 1])):
                    break
                m -= 1
            for k in range(m, i):
                nr = min(3, i - k + 1)
                if k > m:
                    v = np.zeros(3); v[:nr] = H[k:k + nr, k - 1]
                beta, t1, x = dlarfg(nr, v[0], v[1:nr].copy())
                v[0] = beta; v[1:nr] = x
                if k > m:
                    H[k, k - 1] = v[0]; H[k + 1, k - 1] = 0.0
                    if k < i - 1: H[k + 2, k - 1] = 0.0
                elif m > l:
                    H[k, k - 1] = H[k, k - 1] * (1.0 - t1)
                v2 = v[1]; t2 = t1 * v2
                if nr == 3:
                    v3 = v[2]; t3 = t1 * v3
                    for j in range(k, i2 + 1):
                        sm = H[k, j] + v2 * H[k + 1, j] + v3 * H[k + 2, j]
                        H[k, j] -= sm * t1; H[k + 1, j] -= sm * t2; H[k + 2, j] -= sm * t3
                    for j in range(i1, min(k + 3, i) + 1):
                        sm = H[j, k] + v2 * H[j, k + 1] + v3 * H[j, k + 2]
                        H[j, k] -= sm * t1; H[j, k + 1] -= sm * t2; H[j, k + 2] -= sm * t3
                    for j in range(n):
                        sm = Z[j, k] + v2 * Z[j, k + 1] + v3 * Z[j, k + 2]
                        Z[j, k] -= sm * t1; Z[j, k + 1] -= sm * t2; Z[j, k + 2] -= sm * t3
                else:
                    for j in range(k, i2 + 1):
                        sm = H[k, j] + v2 * H[k + 1, j]
                        H[k, j] -= sm * t1; H[k + 1, j] -= sm * t2
                    for j in range(i1, i + 1):
                        sm = H[j, k] + v2 * H[j, k + 1]
                        H[j, k] -= sm * t1; H[j, k + 1] -= sm * t2
                    for j in range(n):
                        sm = Z[j, k] + v2 * Z[j, k + 1]
                        Z[j, k] -= sm * t1; Z[j, k + 1] -= sm * t2
        if not converged:
            return wr, wi, i + 1
        if l == i:
            wr[i] = H[i, i]; wi[i] = 0.0
        elif l == i - 1:
            a, b, c, d, r1r, r1i, r2r, r2i, cs, sn = dlanv2(H[i - 1, i - 1], H[i - 1, i], H[i, i - 1], H[i, i])
            H[i - 1, i - 1], H[i - 1, i], H[i, i - 1], H[i, i] = a, b, c, d
            wr[i - 1], wi[i - 1], wr[i], wi[i] = r1r, r1i, r2r, r2i
            if i2 > i:
                for j in range(i + 1, i2 + 1):
                    H[i - 1, j], H[i, j] = drot(H[i - 1, j], H[i, j], cs, sn)
            for j in range(i1, i - 1):
                H[j, i - 1], H[j, i] = drot(H[j, i - 1], H[j, i], cs, sn)
            for j in range(n):
                Z[j, i - 1], Z[j, i] = drot(Z[j, i - 1], Z[j, i], cs, sn)
        kdefl = 0
        i = l - 1
    return wr, wi, 0

def dtrevc_real(T, Z):
    """right eigenvectors of upper triangular T (all eigenvalues real), back-transformed by Z; each scaled by 1/max|.|"""
    n = T.shape[0]
    smlnum = SAFMIN * (n / ULP)
    VR = np.zeros((n, n))
    for ki in range(n - 1, -1, -1):
        wr = T[ki, ki]
        smin = max(ULP * abs(wr), smlnum)
        x = np.zeros(n)
        x[ki] = 1.0
        for k in range(ki):
            x[k] = -T[k, ki]
        for j in range(ki - 1, -1, -1):
            # dlaln2 1x1: solve (T[j,j] - wr) * xx = scale * x[j]  (smin perturbation)
            den = T[j, j] - wr
            if abs(den) < smin:
                den = smin
            xx = x[j] / den          # (no overflow scaling in our range)
            x[j] = xx
            x[:j] -= xx * T[:j, j]
        v = Z[:, :ki + 1] @ x[:ki + 1]
        emax = np.abs(v).max()
        VR[:, ki] = v / emax
    return VR

def dgeev(A, old=False):
    n = A.shape[0]
    Ab, ilo, ihi, scale = dgebal(A)
    Hs, Q = dgehd2_orghr(Ab, ilo, ihi)
    H = np.triu(Hs, -1)
    Z = Q.copy()
    wr, wi, info = dlahqr(H, Z, ilo, ihi, old=old)
    # isolated eigenvalues (outside ilo..ihi) are the diagonal entries
    for i in list(range(0, ilo)) + list(range(ihi + 1, n)):
        wr[i] = H[i, i]
    if info or np.any(wi != 0):
        return wr, wi, None, info
    VR = dtrevc_real(H, Z)
    # dgebak: undo scaling (rows ilo..ihi times scale) then permutation
    if ilo != ihi:
        for i in range(ilo, ihi + 1):
            VR[i, :] *= scale[i]
    for ii in list(range(ilo - 1, -1, -1)) + list(range(ihi + 1, n)):
        k = int(scale[ii])
        if k != ii:
            VR[[ii, k], :] = VR[[k, ii], :]
    VR /= np.sqrt((VR ** 2).sum(axis=0))
    return wr, wi, VR, 0

if __name__ == "__main__":
    import sys, torch
    np.random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    from collections import Counter
    c = Counter()
    for n in (1, 2, 3, 4):
        for t in range(1500):
            A = np.random.randn(n, n); A = A + A.T
            w, V = np.linalg.eigh(A); w[0] = -1e-4 * np.random.rand()
            A = (V * w) @ V.T
            if t % 3 == 1:
                A = A + 1e-15 * np.random.randn(n, n)
            if t % 7 == 3 and n > 1:
                z = np.random.randint(n); A[z, :] = 0; A[:, z] = 0
            w1, V1 = np.linalg.eig(A)
            w2, V2 = torch.linalg.eig(torch.tensor(A)); w2 = w2.real.numpy(); V2 = V2.real.numpy()
            w3, wi3, V3, info = dgeev(A)
            ok_np = V3 is not None and np.allclose(w1, w3, atol=1e-9) and np.allclose(V1, V3, atol=1e-7)
            ok_mkl = V3 is not None and np.allclose(w2, w3, atol=1e-9) and np.allclose(V2, V3, atol=1e-7)
            c[(n, 'port==openblas', ok_np)] += 1
            c[(n, 'port==mkl', ok_mkl)] += 1
    for k in sorted(c): print(k, c[k])

def debug(n, zero=False, seed=0, count=3):
    import torch
    np.random.seed(seed)
    shown = 0
    for t in range(3000):
        A = np.random.randn(n, n); A = A + A.T
        w, V = np.linalg.eigh(A); w[0] = -1e-4 * np.random.rand()
        A = (V * w) @ V.T
        if zero:
            z = np.random.randint(n); A[z, :] = 0; A[:, z] = 0
        w1, V1 = np.linalg.eig(A)
        w3, wi3, V3, info = dgeev(A)
        ok = V3 is not None and np.allclose(w1, w3, atol=1e-9) and np.allclose(V1, V3, atol=1e-7)
        if not ok:
            print("A=", repr(A)); print("w_np", w1, "\nw_port", w3, wi3, info); print(V1); print(V3)
            shown += 1
            if shown >= count: break
