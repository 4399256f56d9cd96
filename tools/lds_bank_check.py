#!/usr/bin/env python3
"""Bank-conflict model of fwd_fused2's latent staging image (MI355X_MICROARCH.md, LDS table) for the old and the new chunk
swizzle: extra LDS cycles per wave-instruction of the three access kinds (16-byte writes, 16-byte c^T fragment reads, 8-byte
drain reads).  python tools/lds_bank_check.py"""

B128_READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_READ_GROUPS += [[l + 32 for l in g] for g in B128_READ_GROUPS]


def extra_cycles(addrs, groups, nbytes, nbanks):
    """addrs[lane] = byte address; a group takes max over banks of the number of DISTINCT dword addresses on it."""
    extra = 0
    for g in groups:
        per_bank = {}
        for l in g:
            for dw in range(nbytes // 4):
                a = addrs[l] // 4 + dw
                per_bank.setdefault(a % nbanks, set()).add(a)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def swz_old(row):
    return row & 7


def swz_new(row):
    return (row & 7) ^ ((row >> 1) & 1) ^ ((row >> 4) & 1)


def report(name, swz):
    wr = rd = dr = 0
    for chunk in range(8):                   # writes / fragment reads: lane (arow, ah) -> chunk index 4 hf + 2 ks + ah
        for ah_pair in [(0, 1)]:
            addrs = [(l & 31) * 128 + ((((chunk & 6) | (l >> 5)) ^ swz(l & 31)) << 4) for l in range(64)]
            wr += extra_cycles(addrs, [list(range(8 * g, 8 * g + 8)) for g in range(8)], 16, 32)
            rd += extra_cycles(addrs, B128_READ_GROUPS, 16, 64)
    for p in range(4):                       # drain piece p: lane -> row 8 p + lane / 8, chunk 4 tl + 2 s + {0, 1}, half u
        for half in (0, 1):
            addrs = []
            for l in range(64):
                r, dch = 8 * p + (l >> 3), l & 7
                tl, s, u = dch >> 2, (dch & 3) >> 1, dch & 1
                addrs.append(r * 128 + (((4 * tl + 2 * s + half) ^ swz(r)) << 4) + 8 * u)
            dr += extra_cycles(addrs, [list(range(32)), list(range(32, 64))], 8, 64)
    print(f"{name}: extra LDS cycles summed over the 8 chunk positions: 16-byte writes {wr}, c^T 16-byte reads {rd}; over the 8 drain reads: {dr}")


report("swizzle (row & 7)                               [round 3]", swz_old)
report("swizzle (row & 7) ^ (row>>1 & 1) ^ (row>>4 & 1) [round 4]", swz_new)
