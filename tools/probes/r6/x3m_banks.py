"""Exhaustive LDS bank check of conv_maps_x3's halo layouts (csrc/igemm_x3m.hip): for every 32-row sub-tile, every one of the nine tap
shifts and both hardware lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}), the 16 lanes must touch 16 different
16-B bank groups.  Slot s holds 64 B; the 16-B piece p of a slot sits at s * 64 + ((p ^ ((s >> 2) & 3)) << 4).  Prints the worst multiplicity
per map width (1 = conflict-free).  CPU only."""
G1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G2 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]


def perm16(m):
    qd = m >> 2
    return ((bin(qd).count("1") & 1) << 4) | ((qd >> 1) << 2) | (m & 3)


def quads(p8):
    return lambda m: p8[m >> 2] * 4 + (m & 3)


LAYOUTS = {            # map width: (rows per tile and map, slot pitch, slots per map, MFMA row -> pixel)
    32: (8, 34, 340, lambda m: m),
    16: (16, 18, 324, perm16),
    8: (8, 12, 120, quads([0, 3, 5, 1, 6, 2, 4, 7])),
    4: (4, 6, 40, quads([0, 1, 3, 2, 5, 4, 6, 7])),
}


def worst(PW):
    PH, P, HS, perm = LAYOUTS[PW]
    w = 1
    for sub in range(0, 256, 32):
        for kh in range(3):
            for kw in range(3):
                for G in (G1, G2):
                    seen = {}
                    for lane in G:
                        pr = sub + perm(lane)
                        sp, q = divmod(pr, PW * PH)
                        py, px = divmod(q, PW)
                        sl = sp * HS + (py + kh) * P + px + kw
                        b = (sl & 3) * 4 + ((sl >> 2) & 3)          # (the k-piece XOR is a bijection of the class: left out)
                        seen[b] = seen.get(b, 0) + 1
                    w = max(w, max(seen.values()))
    return w


if __name__ == "__main__":
    for PW in (32, 16, 8, 4):
        print(PW, worst(PW))
