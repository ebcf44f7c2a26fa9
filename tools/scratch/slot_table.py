"""Count distinct (source word, destination word, rotation) classes of the per-symbol bit table."""
import collections

def unpunct(P, q):
    if P == 1: return 6 * (q // 4) + (5 if q % 4 == 3 else q % 4)
    if P == 2: return 4 * (q // 3) + q % 3
    return q

def deint(nb, j):
    R = 3 * nb; S = max(nb // 2, 1)
    c, u = divmod(j, R)
    return 16 * (S * (u // S) + ((u % S) + c) % S) + c

def bin_code(k):
    f = k if k < 32 else k - 64
    if f == 0 or f > 26 or f < -26: return -1
    if f in (-21, -7, 7): return -2
    if f == 21: return -3
    t = f + 26
    return t - (t > 5) - (t > 19) - (t > 26) - (t > 33) - (t > 47)

def row_layout():
    """row byte -> ('d', carrier) or ('p', sign) for the 52-byte slot row; returns list and per-lane offsets"""
    row, offs = [], []
    for l in range(8):
        offs.append(len(row))
        for m in range(8):
            c = bin_code(l + 8 * m)
            if c >= 0: row.append(('d', c))
            elif c <= -2: row.append(('p', c))
    return row, offs

for regime, (nb, P) in {1: (1, 0), 2: (1, 1), 3: (2, 0), 4: (2, 1), 5: (4, 0), 6: (4, 1), 7: (6, 2), 8: (6, 1)}.items():
    row, offs = row_layout()
    cls = collections.Counter()
    nbits = 0
    for rb, (kind, c) in enumerate(row):
        if kind != 'd': continue
        for t in range(nb):
            j = nb * c + t
            p = unpunct(P, deint(nb, j)); n = p >> 1
            src = ('B' if p & 1 else 'A', n >> 5); sbit = 31 - (n & 31)
            dpos = 8 * (rb % 4) + (nb - 1 - t); dw = rb // 4
            cls[(src, dw, (dpos - sbit) % 32)] += 1
            nbits += 1
    # message-word layout for comparison
    cls2 = collections.Counter()
    for j in range(48 * nb):
        p = unpunct(P, deint(nb, j)); n = p >> 1
        src = ('B' if p & 1 else 'A', n >> 5); sbit = 31 - (n & 31)
        cls2[(src, j >> 5, ((31 - (j & 31)) - sbit) % 32)] += 1
    print(regime, "bits", nbits, "slot-layout classes", len(cls), "msg-layout classes", len(cls2))
print(row_layout()[1], len(row_layout()[0]))
