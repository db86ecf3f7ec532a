"""LDS bank-conflict estimate for the window reads, per MI355X_MICROARCH.md LDS table."""
import itertools
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 += [[l+32 for l in g] for g in G128]
G64 = [list(range(32)), list(range(32,64))]
def cycles(groups, addr_of_lane, ndw, nbanks):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for j in range(ndw):
                banks.setdefault((a + j) % nbanks, set()).add(a + j)
        tot += max(len(v) for v in banks.values())
    return tot
def eval_layout(pitch, reads, wave=0):
    # reads: list of (dword offset within window row, ndw, kind)
    total = ideal = 0
    for k in range(6):
        for off, ndw, kind in reads:
            def addr(l, k=k, off=off):
                tx, ty = l & 15, 4 * wave + (l >> 4)
                return (2 * ty + k) * pitch + 8 * tx + 6 + off
            if kind == 128:
                total += cycles(G128, addr, 4, 64); ideal += 4
            elif kind == 64:
                total += cycles(G64, addr, 2, 64); ideal += 2
            else:
                total += cycles(G64, addr, 1, 32); ideal += 2
    return total, ideal
cur = [(0, 2, 64), (2, 4, 128), (6, 4, 128), (10, 2, 64)]
all64 = [(2 * j, 2, 64) for j in range(6)]
for pitch in range(144, 176, 2):
    res = []
    if pitch % 4 == 0:
        res.append(("b64+2b128+b64", eval_layout(pitch, cur)))
    res.append(("6 x b64", eval_layout(pitch, all64)))
    print(pitch, res)

print("---- with a column pad every 64 columns: pos(col) = col + pad*(col>>6)")
def eval_layout2(pitch, reads, pad, wave=0):
    total = ideal = 0
    for k in range(6):
        for off, ndw, kind in reads:
            def addr(l, k=k, off=off):
                tx, ty = l & 15, 4 * wave + (l >> 4)
                col = 8 * tx + 6 + off
                return (2 * ty + k) * pitch + col + pad * (col >> 6)
            if kind == 128:
                total += cycles(G128, addr, 4, 64); ideal += 4
            else:
                total += cycles(G64, addr, 2, 64); ideal += 2
    return total, ideal
best = []
for pad in (2, 4, 8):
    for pitch in range(148, 200, 2):
        for name, reads in (("6xb64", all64), ("mixed", cur)):
            if name == "mixed" and (pitch % 4 or pad % 4):
                continue
            t, i = eval_layout2(pitch, reads, pad)
            worst = max(eval_layout2(pitch, reads, pad, w)[0] for w in range(4))
            best.append((worst, name, pad, pitch))
best.sort()
print(best[:12])

print("---- two-plane layout: even 16-B slots in [0,HALF), odd slots in [HALF, 2*HALF); 4 x b128 per row")
def pos(col, HALF):
    s = col >> 2
    return (s >> 1) * 4 + (col & 3) + (s & 1) * HALF
for HALF, pitch in ((80, 160), (72, 144), (76, 152), (96, 192)):
    worst = 0
    for wave in range(4):
        tot = 0
        for k in range(6):
            for rd in range(4):
                def addr(l, k=k, rd=rd):
                    tx, ty = l & 15, 4 * wave + (l >> 4)
                    return (2 * ty + k) * pitch + pos(8 * tx + 4 + 4 * rd, HALF)
                tot += cycles(G128, addr, 4, 64)
        worst = max(worst, tot)
    # fill stores: 64 consecutive units of one row-major sweep
    def saddr(l, plane):
        u = l  # consecutive units
        lr, lu = divmod(u, 18)
        return lr * pitch + pos(8 * lu + 4 * plane, HALF)
    st = sum(cycles([list(range(i, i + 8)) for i in range(0, 64, 8)], lambda l, pl=pl: saddr(l, pl), 4, 32) for pl in (0, 1))
    print("HALF", HALF, "pitch", pitch, "read cycles/strip", worst, "(ideal 96)", "store cycles (8x8 groups)", st, "(ideal 16)")
