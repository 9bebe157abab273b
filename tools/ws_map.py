"""Offsets of the named regions of a workspace (mirror of csrc/xg_model.hip:carve) -- for diagnosis scripts that diff workspaces."""


def ws_map(B, K, R, A, E, V, C, H, T, sk_max_jobs=5, dsync_bytes=1024):
    off = 0
    out = []

    def take(name, n, size=4):
        nonlocal off
        off = (off + 255) & ~255
        out.append((name, off, n * size))
        off += n * size
    N, TB = B * K, T * B
    for m in range(2):
        for name, n in (("Z", N * R), ("X", N * R), ("PRE", N * 4 * R), ("Hs", N * R), ("Cs", N * R), ("G", N * 4 * R), ("GG", N * R),
                        ("Hprev", N * R), ("bn_mean", R), ("bn_var", R), ("dHs", N * R), ("dGG", N * R), ("dS", N * 4 * R), ("dX", N * R),
                        ("dCrec1", B * R)):
            take("%s[%d]" % (name, m), n)
    for name, n in (("zeroBR", B * R), ("S", B * 4 * R), ("S2", B * 4 * R), ("Y", N * 2 * R), ("Venc", N * R), ("dVw", N * R), ("dY", N * 2 * R),
                    ("vbar", B * R), ("vproj", N * A), ("Xe", TB * E), ("GP", TB * R), ("POSG", TB * R), ("PRE1", TB * 4 * R),
                    ("H1", (T + 1) * B * R), ("C1", (T + 1) * B * R), ("H2", (T + 1) * B * R), ("C2", (T + 1) * B * R),
                    ("G1", TB * 4 * R), ("G2", TB * 4 * R), ("P", TB * A), ("ALPHA", TB * K), ("AF", TB * R), ("LOGITS", TB * V),
                    ("HC", TB * H), ("CL", TB * C), ("LSE", TB), ("LSEC", TB), ("sums", 8), ("DH2OUT", TB * R), ("DHC", TB * H),
                    ("DCL", TB * C), ("DS1", TB * 4 * R), ("DS2", TB * 4 * R), ("DP", TB * A), ("DE", TB * K)):
        take(name, n)
    for j in range(4):
        take("dst1[%d]" % j, B * R)
    for name, n in (("DVPROJ", N * A), ("DV", N * R), ("DPOSG", TB * R), ("DH1X", TB * R), ("DGP", TB * R), ("DXe", TB * E),
                    ("state_tmp", 4 * B * R), ("AFU+ATS", B * R + ((B + 3) & ~3))):
        take(name, n)
    take("TOK", TB, 8); take("TOKLP", TB); take("UNF", TB); take("alive", 4)
    off = (off + 15) & ~15
    take("VPART", B * ((V + 31) // 32) * 4)
    take("dsync", dsync_bytes // 4)
    take("DAF", TB * R)
    return out


def locate(regions, byte_off):
    for name, o, n in regions:
        if o <= byte_off < o + n:
            return name, (byte_off - o) // 4
    return "?", byte_off
