"""Plain-Python BN254 tower arithmetic used to check Execute-mode gadget outputs.

Independent of both the oracle and the product: integers mod p only.
Tower (ark-bn254): Fq2 = Fq[u]/(u^2+1); Fq6 = Fq2[v]/(v^3 - xi), xi = 9+u; Fq12 = Fq6[w]/(w^2 - v).
Wire order of an Fq12 (reference fq12.rs:17-24, fq6.rs:17-25, fq2.rs:33-41):
  c0.c0.c0, c0.c0.c1, c0.c1.c0, c0.c1.c1, c0.c2.c0, c0.c2.c1, c1.c0.c0, ... (12 x 254 bits, LSB first)
"""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 1 << 254
RINV = pow(R, -1, P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


XI = (9, 1)


def f2_mul_xi(a):
    return f2_mul(a, XI)


def f6_add(a, b):
    return tuple(f2_add(x, y) for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple(f2_sub(x, y) for x, y in zip(a, b))


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = f2_mul(a0, b0), f2_mul(a1, b1), f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(t2))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a2, b0)), t1)
    return (c0, c1, c2)


def f6_mul_v(a):  # multiply by v: (c0,c1,c2) -> (xi*c2, c0, c1)
    return (f2_mul_xi(a[2]), a[0], a[1])


def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0, t1 = f6_mul(a0, b0), f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_add(f6_mul(a0, b1), f6_mul(a1, b0))
    return (c0, c1)


def f12_scale(a, k):
    return tuple(tuple(tuple((x * k) % P for x in f2) for f2 in f6) for f6 in a)


def f12_flatten(a):
    return [x for f6 in a for f2 in f6 for x in f2]


def f12_unflatten(v):
    v = list(v)
    return tuple(tuple((v[6 * i + 2 * j], v[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2))


def f6_flatten(a):
    return [x for f2 in a for x in f2]


def f6_unflatten(v):
    v = list(v)
    return tuple((v[2 * j], v[2 * j + 1]) for j in range(3))


# ---- pairing (mirrors the reference's NATIVE helpers, src/gadgets/bn254/pairing.rs:30-133, and arkworks' Miller loop) ----
X_BN = 4965661367192848881
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP_COUNT = [0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0,
                  0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1]
assert sum(d << i for i, d in enumerate(ATE_LOOP_COUNT)) == 6 * X_BN + 2
HALF = pow(2, -1, P)
F12_ONE = (((1, 0), (0, 0), (0, 0)), ((0, 0), (0, 0), (0, 0)))


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_scale(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def f2_sq(a):
    return f2_mul(a, a)


def f2_pow(x, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, x)
        x = f2_mul(x, x)
        e >>= 1
    return r


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return ((a[0] * n) % P, (-a[1] * n) % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f12_pow(x, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, x)
        x = f12_mul(x, x)
        e >>= 1
    return r


COEFF_B_G2 = f2_mul((3, 0), f2_inv(XI))
TWIST_MUL_BY_Q_X = f2_pow(XI, (P - 1) // 3)
TWIST_MUL_BY_Q_Y = f2_pow(XI, (P - 1) // 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g2_double_in_place(r):  # pairing.rs:30-52 -> (new r, (c0, c1, c2))
    x, y, z = r
    a = f2_scale(f2_mul(x, y), HALF)
    b, c = f2_sq(y), f2_sq(z)
    e = f2_mul(COEFF_B_G2, f2_add(f2_add(c, c), c))
    f = f2_add(f2_add(e, e), e)
    g = f2_scale(f2_add(b, f), HALF)
    h = f2_sub(f2_sq(f2_add(y, z)), f2_add(b, c))
    i = f2_sub(e, b)
    j = f2_sq(x)
    e2 = f2_sq(e)
    nr = (f2_mul(a, f2_sub(b, f)), f2_sub(f2_sq(g), f2_add(f2_add(e2, e2), e2)), f2_mul(b, h))
    return nr, (f2_neg(h), f2_add(f2_add(j, j), j), i)


def g2_add_in_place(r, q):  # pairing.rs:54-73, q = (x, y) affine
    x, y, z = r
    theta = f2_sub(y, f2_mul(q[1], z))
    lam = f2_sub(x, f2_mul(q[0], z))
    c, d = f2_sq(theta), f2_sq(lam)
    e, f, g = f2_mul(lam, d), f2_mul(z, c), f2_mul(x, d)
    h = f2_sub(f2_add(e, f), f2_add(g, g))
    j = f2_sub(f2_mul(theta, q[0]), f2_mul(lam, q[1]))
    nr = (f2_mul(lam, h), f2_sub(f2_mul(theta, f2_sub(g, h)), f2_mul(e, y)), f2_mul(z, e))
    return nr, (lam, f2_neg(theta), j)


def g2_mul_by_char(q):  # pairing.rs:75-83
    return (f2_mul(f2_conj(q[0]), TWIST_MUL_BY_Q_X), f2_mul(f2_conj(q[1]), TWIST_MUL_BY_Q_Y))


def ell_coeffs(q):  # pairing.rs:88-126
    out = []
    r = (q[0], q[1], (1, 0))
    nq = (q[0], f2_neg(q[1]))
    for bit in list(reversed(ATE_LOOP_COUNT))[1:]:
        r, c = g2_double_in_place(r)
        out.append(c)
        if bit == 1:
            r, c = g2_add_in_place(r, q)
            out.append(c)
        elif bit == -1:
            r, c = g2_add_in_place(r, nq)
            out.append(c)
    q1 = g2_mul_by_char(q)
    q2 = g2_mul_by_char(q1)
    q2 = (q2[0], f2_neg(q2[1]))
    r, c = g2_add_in_place(r, q1)
    out.append(c)
    r, c = g2_add_in_place(r, q2)
    out.append(c)
    return out


def ell(f, coeffs, p):  # f * (c0 * p.y + c1 * p.x * w^3 + c2 * w^4): Fq12 element (c0', 0, 0) + (c1', c2, 0) w
    c0 = f2_scale(coeffs[0], p[1])
    c3 = f2_scale(coeffs[1], p[0])
    return f12_mul(f, ((c0, (0, 0), (0, 0)), (c3, coeffs[2], (0, 0))))


def multi_miller_loop(pairs):  # pairs: [(p affine G1 (x, y), q affine G2 (x, y))]; the reference's loop shape, pairing.rs:945-1007
    ells = [iter(ell_coeffs(q)) for _, q in pairs]
    f = F12_ONE
    n = len(ATE_LOOP_COUNT)
    for i in range(n - 1, 0, -1):
        if i != n - 1:
            f = f12_mul(f, f)
        for (p, _), it in zip(pairs, ells):
            f = ell(f, next(it), p)
        if ATE_LOOP_COUNT[i - 1] in (1, -1):
            for (p, _), it in zip(pairs, ells):
                f = ell(f, next(it), p)
    for _ in range(2):
        for (p, _), it in zip(pairs, ells):
            f = ell(f, next(it), p)
    return f


def final_exponentiation(f):  # final_exponentiation.rs:37-63
    conj = lambda x: f12_pow(x, P ** 6)
    inv = lambda x: f12_pow(x, P ** 12 - 2)
    frob = lambda x, i: f12_pow(x, P ** i)
    nx = lambda g: conj(f12_pow(g, X_BN))
    m = f12_mul
    u = m(inv(f), conj(f)); r = m(frob(u, 2), u)
    y0 = nx(r); y1 = m(y0, y0); y2 = m(y1, y1); y3 = m(y2, y1)
    y4 = nx(y3); y5 = m(y4, y4); y6 = nx(y5); y7 = conj(y3); y8 = conj(y6)
    y9 = m(y8, y4); y10 = m(y9, y7); y11 = m(y10, y1); y12 = m(y10, y4); y13 = m(y12, r)
    y14 = frob(y11, 1); y15 = m(y14, y13); y16 = frob(y10, 2); y17 = m(y16, y15); y18 = m(conj(r), y11)
    return m(frob(y18, 3), y17)


def g1_mul(k, p=(1, 2)):  # affine scalar multiplication on y^2 = x^3 + 3 (double-and-add, no point at infinity handling beyond None)
    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % P == 0: return None
            lam = (3 * a[0] * a[0]) * pow(2 * a[1], -1, P) % P
        else:
            lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
        x = (lam * lam - a[0] - b[0]) % P
        return (x, (lam * (a[0] - x) - a[1]) % P)
    r = None
    while k:
        if k & 1: r = add(r, p)
        p = add(p, p)
        k >>= 1
    return r


def g2_mul(k, q=G2_GEN):  # affine scalar multiplication on the twist y^2 = x^3 + 3/xi
    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if f2_add(a[1], b[1]) == (0, 0): return None
            lam = f2_mul(f2_scale(f2_sq(a[0]), 3), f2_inv(f2_add(a[1], a[1])))
        else:
            lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
        x = f2_sub(f2_sub(f2_sq(lam), a[0]), b[0])
        return (x, f2_sub(f2_mul(lam, f2_sub(a[0], x)), a[1]))
    r = None
    while k:
        if k & 1: r = add(r, q)
        q = add(q, q)
        k >>= 1
    return r
