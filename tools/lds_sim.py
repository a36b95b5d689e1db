#!/usr/bin/env python3
"""Bank-conflict model of the fused kernel's two LDS transpositions (N = 1024,
2048, 4096), per MI355X_MICROARCH.md §LDS:
  ds_write_b64 : 4 groups of 16 contiguous lanes, bank = (addr/4) % 32 (2 banks per lane)
  ds_read_b64  : 2 groups of 32 lanes,            bank = (addr/4) % 64
  ds_read_b128 : 4 groups {0-3,12-15,20-27},{4-11,16-19,28-31},(+32), bank % 64 (4 banks per lane)
Cycles for a group = max number of DISTINCT addresses (per bank) on one bank.
"""
import itertools
import sys


def rev16(s):
    return 4 * (s & 3) + (s >> 2)


def cycles(addrs_f2, lanes_groups, width_dw, modulo):
    """addrs_f2: per-lane address in float2 (8 B) units; width in dwords."""
    tot = 0
    for grp in lanes_groups:
        banks = {}
        for l in grp:
            a_dw = addrs_f2[l] * 2
            for w in range(width_dw):
                banks.setdefault((a_dw + w) % modulo, set()).add((a_dw + w))
        tot += max(len(v) for v in banks.values())
    return tot


W64 = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
R64 = [list(range(0, 32)), list(range(32, 64))]
R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
        [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
R128 = R128 + [[l + 32 for l in g] for g in R128]


def analyse(N, lds1, lds2, verbose=True):
    T, R3 = N // 16, N // 256
    J = 16 // R3
    res = {}
    for wave in range(T // 64):
        lanes = [wave * 64 + l for l in range(64)]
        # transposition 1 write: fixed s, lane t writes (rev16(s), t)
        w1 = sum(cycles([lds1(rev16(s), t) for t in lanes], W64, 2, 32) for s in range(16))
        # transposition 1 read: fixed r2, lane t=(q1,m2) reads (q1, R3*r2+m2)
        r1 = sum(cycles([lds1(t // R3, R3 * r2 + t % R3) for t in lanes], R64, 2, 64) for r2 in range(16))
        w2 = sum(cycles([lds2(t // R3, t % R3, rev16(s)) for t in lanes], W64, 2, 32) for s in range(16))
        r2c = sum(cycles([lds2(J * (t % R3) + j, m, t // R3) for t in lanes], R64, 2, 64)
                  for j in range(J) for m in range(R3))
        res[wave] = (w1, r1, w2, r2c)
    if verbose:
        for wave, (w1, r1, w2, r2c) in res.items():
            print("N=%d wave %d: write1 %d (ideal 64)  read1 %d (ideal 32)  write2 %d (64)  read2 %d (32)"
                  % (N, wave, w1, r1, w2, r2c))
    return res


if __name__ == "__main__":
    for N in (1024, 2048, 4096):
        T, R3 = N // 16, N // 256
        analyse(N, lambda q1, m1: q1 * (T + R3) + m1,
                lambda q1, m2, q2: q2 * (T + R3) + q1 * R3 + m2)


def search(N, max_f2):
    """Brute-force linear layouts for both transpositions and both pass-2 lane maps."""
    T, R3 = N // 16, N // 256
    J = 16 // R3
    best = []
    lanes_all = [[w * 64 + l for l in range(64)] for w in range(T // 64)]
    for pmap in ("A", "B"):
        def qm(t):
            return (t // R3, t % R3) if pmap == "A" else (t % 16, t // 16)
        # transposition 1: lds1 = q1*S1 + m1
        best1 = None
        for S1 in range(T, T + 33):
            c = 0
            for lanes in lanes_all:
                c += sum(cycles([rev16(s) * S1 + t for t in lanes], W64, 2, 32) for s in range(16))
                c += sum(cycles([qm(t)[0] * S1 + R3 * r2 + qm(t)[1] for t in lanes], R64, 2, 64)
                         for r2 in range(16))
            if best1 is None or c < best1[0]:
                best1 = (c, S1)
        # transposition 2: lds2 = q2*A + q1*B + m2*C ; reads either b64 per (j,m) or b128 per (j, m pair) when C == 1
        best2 = None
        for C in (1, 16, 17, 18, 20):
            for B in range(1, 40):
                for A in range(1, 160):
                    # must be injective and fit
                    mx = 15 * A + 15 * B + (R3 - 1) * C
                    if mx >= max_f2:
                        continue
                    seen = set()
                    ok = True
                    for q2 in range(16):
                        for q1 in range(16):
                            for m2 in range(R3):
                                a = q2 * A + q1 * B + m2 * C
                                if a in seen:
                                    ok = False
                                    break
                                seen.add(a)
                            if not ok:
                                break
                        if not ok:
                            break
                    if not ok:
                        continue
                    c = 0
                    for lanes in lanes_all:
                        c += sum(cycles([rev16(s) * A + qm(t)[0] * B + qm(t)[1] * C for t in lanes], W64, 2, 32)
                                 for s in range(16))
                        if C == 1 and R3 % 2 == 0:
                            c += sum(cycles([(t // R3) * A + (J * (t % R3) + j) * B + m for t in lanes], R128, 4, 64)
                                     for j in range(J) for m in range(0, R3, 2))
                        else:
                            c += sum(cycles([(t // R3) * A + (J * (t % R3) + j) * B + m * C for t in lanes], R64, 2, 64)
                                     for j in range(J) for m in range(R3))
                    if best2 is None or c < best2[0]:
                        best2 = (c, A, B, C)
        print("N=%d pass-2 map %s: transposition 1 best S1=%d cycles %d (ideal %d); transposition 2 best A=%d B=%d C=%d cycles %d (ideal %d)"
              % (N, pmap, best1[1], best1[0], 96 * (T // 64), best2[1], best2[2], best2[3], best2[0], 96 * (T // 64)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "search":
    search(1024, 16 * 68 + 2 + 64)


def search_b128(N, max_f2):
    """Layouts where every lane's 16 reads are contiguous (8 x ds_read_b128)."""
    T, R3 = N // 16, N // 256
    J = 16 // R3
    lanes_all = [[w * 64 + l for l in range(64)] for w in range(T // 64)]
    res1, res2 = [], []
    for P in range(0, 9):
        for S in range(R3 * (16 + P), R3 * (16 + P) + 40):
            if 15 * S + (R3 - 1) * (16 + P) + 15 >= max_f2:
                continue
            c_w = c_r = 0
            for lanes in lanes_all:
                c_w += sum(cycles([rev16(s) * S + (t % R3) * (16 + P) + t // R3 for t in lanes], W64, 2, 32) for s in range(16))
                c_r += sum(cycles([(t // R3) * S + (t % R3) * (16 + P) + 2 * i for t in lanes], R128, 4, 64) for i in range(8))
            res1.append((c_w + c_r, c_w, c_r, S, P))
    res1.sort()
    print("N=%d transposition 1 (b128 reads): best (total, write, read, S, P):" % N, res1[:4])
    for P in range(0, 9):
        for A in range(R3 * (16 + P), R3 * (16 + P) + 40):
            if 15 * A + (R3 - 1) * (16 + P) + 15 >= max_f2:
                continue
            c_w = c_r = 0
            for lanes in lanes_all:
                # writer lane (q1, m2) = (t//R3, t%R3); element (q1, m2, q2): q2*A + (q1//J)*(16+P) + (q1%J)*R3 + m2
                c_w += sum(cycles([rev16(s) * A + ((t // R3) // J) * (16 + P) + ((t // R3) % J) * R3 + t % R3 for t in lanes], W64, 2, 32)
                           for s in range(16))
                # reader lane (q2, g) = (t//R3, t%R3): q2*A + g*(16+P) + 2i
                c_r += sum(cycles([(t // R3) * A + (t % R3) * (16 + P) + 2 * i for t in lanes], R128, 4, 64) for i in range(8))
            res2.append((c_w + c_r, c_w, c_r, A, P))
    res2.sort()
    print("N=%d transposition 2 (b128 reads): best (total, write, read, A, P):" % N, res2[:4])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "b128":
    for N in (1024, 2048, 4096):
        search_b128(N, 16 * (N // 16 + N // 256) * 2)


def search_t1_general(N, max_f2):
    T, R3 = N // 16, N // 256
    lanes_all = [[w * 64 + l for l in range(64)] for w in range(T // 64)]
    res = []
    for Pp in range(16, 80):
        for S in range(16, 200):
            mx = 15 * S + (R3 - 1) * Pp + 15
            if mx >= max_f2:
                continue
            addrs = set()
            ok = True
            for q1 in range(16):
                for m2 in range(R3):
                    for r2 in range(16):
                        a = q1 * S + m2 * Pp + r2
                        if a in addrs:
                            ok = False
                        addrs.add(a)
            if not ok:
                continue
            c_w = c_r = 0
            for lanes in lanes_all:
                c_w += sum(cycles([rev16(s) * S + (t % R3) * Pp + t // R3 for t in lanes], W64, 2, 32) for s in range(16))
                c_r += sum(cycles([(t // R3) * S + (t % R3) * Pp + 2 * i for t in lanes], R128, 4, 64) for i in range(8))
            res.append((c_w + c_r, c_w, c_r, S, Pp, mx + 1))
    res.sort()
    print("N=%d transposition 1 general (total, write, read, S, P', f2 used):" % N, res[:6])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "t1":
    search_t1_general(1024, 1300)


def check_swizzled_1024():
    """The XOR-swizzled transposition-1 layout and the padded transposition-2 layout used at N=1024."""
    def h(q1):
        return 8 * ((q1 >> 2) & 1) + (q1 & 1)

    def lds1(q1, m1):
        m2, r2 = m1 & 3, m1 >> 2
        i, e = r2 >> 1, r2 & 1
        g = 16 * (i >> 2) + 8 * ((i >> 1) & 1) + 2 * m2 + (i & 1)
        return q1 * 64 + 2 * (g ^ h(q1)) + e

    def lds2(q1, m2, q2):
        return q2 * 72 + (q1 >> 2) * 18 + (q1 & 3) * 4 + m2

    assert len({lds1(q, m) for q in range(16) for m in range(64)}) == 1024
    assert len({lds2(q1, m2, q2) for q1 in range(16) for m2 in range(4) for q2 in range(16)}) == 1024
    lanes = list(range(64))
    w1 = sum(cycles([lds1(rev16(s), t) for t in lanes], W64, 2, 32) for s in range(16))
    r1 = sum(cycles([lds1(t >> 2, 4 * (2 * i) + (t & 3)) for t in lanes], R128, 4, 64) for i in range(8))
    for t in lanes:
        for i in range(8):
            assert lds1(t >> 2, 4 * (2 * i + 1) + (t & 3)) == lds1(t >> 2, 4 * (2 * i) + (t & 3)) + 1
    w2 = sum(cycles([lds2(t >> 2, t & 3, rev16(s)) for t in lanes], W64, 2, 32) for s in range(16))
    r2 = sum(cycles([lds2(4 * (t & 3) + (k >> 1), 2 * (k & 1), t >> 2) for t in lanes], R128, 4, 64) for k in range(8))
    print("N=1024 swizzled: write1 %d (64) read1 %d (32) write2 %d (64) read2 %d (32); f2 used %d / %d"
          % (w1, r1, w2, r2, 1 + max(lds1(q, m) for q in range(16) for m in range(64)),
             1 + max(lds2(a, b, c) for a in range(16) for b in range(4) for c in range(16))))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "check":
    check_swizzled_1024()


# ---- two virtual threads per lane (spectrum_fused_v2.hip): lane t owns vt = 2t, 2t+1 ----
W128 = [list(range(8 * g, 8 * g + 8)) for g in range(8)]     # ds_write_b128: 8 x 8 contiguous lanes, banks mod 32


def analyse_v2(N, S1, A2, P2, verbose=True):
    """lds1(q1, m1) = q1*S1 + m1; lds2(q1, m2, q2) = q2*A2 + (q1//J)*P2 + (q1%J)*R3 + m2.
    Every access is 16 bytes: the two virtual threads of a lane are adjacent in m1 / m2."""
    T, R3 = N // 16, N // 256
    J = 16 // R3
    TR = T // 2
    tot = {}
    for wave in range(max(1, TR // 64)):
        lanes = [wave * 64 + l for l in range(64)]
        # write 1: fixed s, lane t writes (rev16(s), 2t..2t+1)
        w1 = sum(cycles([rev16(s) * S1 + 2 * t for t in lanes], W128, 4, 32) for s in range(16))
        # read 1: fixed r2; vt = 2t + h: q1 = vt // R3, m2 = vt % R3 (h adjacent)
        r1 = sum(cycles([((2 * t) // R3) * S1 + R3 * r2 + (2 * t) % R3 for t in lanes], R128, 4, 64) for r2 in range(16))
        # write 2: fixed s (q2 = rev16(s)); (q1, m2) as above
        def l2(q1, m2, q2):
            return q2 * A2 + (q1 // J) * P2 + (q1 % J) * R3 + m2
        w2 = sum(cycles([l2((2 * t) // R3, (2 * t) % R3, rev16(s)) for t in lanes], W128, 4, 32) for s in range(16))
        # read 2: virtual thread (q2, g3) = (vt // R3, vt % R3) reads its 16 contiguous elements, 8 x b128, per h
        r2c = 0
        for h in range(2):
            for i in range(8):
                r2c += cycles([l2(J * ((2 * t + h) % R3), 0, (2 * t + h) // R3) + 2 * i for t in lanes], R128, 4, 64)
        tot[wave] = (w1, r1, w2, r2c)
        if verbose:
            print("N=%d v2 wave %d: write1 %d (ideal 128) read1 %d (64) write2 %d (128) read2 %d (64)"
                  % (N, wave, w1, r1, w2, r2c))
    return tot


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "v2":
    for N in (2048, 4096):
        T, R3 = N // 16, N // 256
        analyse_v2(N, T + R3, 18 * R3, 18)


# ---- spectrum_f64_fused.hip: double2 elements (16 B), N = 1024, one wavefront ----
def analyse_f64(row=68, grp=17, verbose=True):
    """Addresses in double2 (16 B) units -> float2 units x2.  Every access is a b128."""
    lanes = list(range(64))
    def f2(a):       # double2 index -> float2 index
        return 2 * a
    w1 = sum(cycles([f2(rev16(s) * row + t) for t in lanes], W128, 4, 32) for s in range(16))
    r1 = sum(cycles([f2((t // 4) * row + 4 * r2 + t % 4) for t in lanes], R128, 4, 64) for r2 in range(16))
    w2 = sum(cycles([f2(rev16(s) * row + (t >> 4) * grp + (t & 15)) for t in lanes], W128, 4, 32) for s in range(16))
    r2 = sum(cycles([f2((t // 4) * row + (t % 4) * grp + i) for t in lanes], R128, 4, 64) for i in range(16))
    if verbose:
        print("f64 N=1024 row=%d grp=%d: write1 %d (ideal 128) read1 %d (64) write2 %d (128) read2 %d (64)"
              % (row, grp, w1, r1, w2, r2))
    return w1, r1, w2, r2


def analyse_f64_n(N, verbose=True):
    """spectrum_f64_fused.hip at N = 2048 / 4096: rows of 17*R3 double2, reader groups of 17."""
    T, R3 = N // 16, N // 256
    J = 16 // R3
    row = 17 * R3
    res = []
    for wave in range(T // 64):
        lanes = [wave * 64 + l for l in range(64)]
        w1 = sum(cycles([2 * (rev16(s) * row + t) for t in lanes], W128, 4, 32) for s in range(16))
        r1 = sum(cycles([2 * ((t // R3) * row + R3 * r2 + t % R3) for t in lanes], R128, 4, 64) for r2 in range(16))
        w2 = sum(cycles([2 * (rev16(s) * row + ((t // R3) // J) * 17 + ((t // R3) % J) * R3 + t % R3) for t in lanes],
                        W128, 4, 32) for s in range(16))
        r2 = sum(cycles([2 * ((t // R3) * row + (t % R3) * 17 + i) for t in lanes], R128, 4, 64) for i in range(16))
        res.append((w1, r1, w2, r2))
        if verbose:
            print("f64 N=%d wave %d: write1 %d (128) read1 %d (64) write2 %d (128) read2 %d (64)" % (N, wave, w1, r1, w2, r2))
    return res


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "f64":
    analyse_f64()
    analyse_f64(64, 16)
    analyse_f64_n(2048)
    analyse_f64_n(4096)
