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
