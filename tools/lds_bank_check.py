"""The LDS images of prefix_attn32_kernel against the bank rule of MI355X_MICROARCH.md (ds_read_b128: four groups of 16
lanes; ds_read_b64_tr_b16: two groups of 32): every fragment read of the K image (piece c of row r at c ^ (r & 15)) and
every transposing read of the V image (piece c at c ^ ((r & 3) << 2)) must touch each bank once.  Run before the kernel's
first launch; SQ_LDS_BANK_CONFLICT = 0 on the hardware agrees (profiles/r6_prefix_attn32_pmc.txt)."""
# LDS bank rule check for the 32x32x16 prefix kernel's images (MI355X_MICROARCH.md, LDS table)
def conflicts_b128(addrs):   # 64 lane byte addresses, ds_read_b128: 4 groups of 16 lanes, bank = (a/4) mod 64, 4 banks each
    groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
    groups += [[l+32 for l in g] for g in groups]
    worst = 0
    for g in groups:
        use = {}
        for l in g:
            for d in range(4):
                b = (addrs[l]//4 + d) % 64
                use.setdefault(b, set()).add(addrs[l]//4 + d)
        worst = max(worst, max(len(v) for v in use.values()))
    return worst
def conflicts_tr(addrs):     # ds_read_b64_tr_b16: 2 groups of 32 lanes, 8 bytes per lane
    worst = 0
    for g in (range(32), range(32, 64)):
        use = {}
        for l in g:
            for d in range(2):
                b = (addrs[l]//4 + d) % 64
                use.setdefault(b, set()).add(addrs[l]//4 + d)
        worst = max(worst, max(len(v) for v in use.values()))
    return worst
# K: lane reads row 32kt + (lane&31), chunk (2ks+hi) ^ (row&15)
for kt in range(2):
    for ks in range(8):
        ad = [ (32*kt + (l&31))*256 + 16*(((2*ks + (l>>5)) ^ (l&15))) for l in range(64)]
        assert conflicts_b128(ad) == 1, (kt, ks, conflicts_b128(ad))
# V: group Gp = l>>4, q = (l&15)>>2, p = l&3: row = base + 4*hi + q; chunk = 4dt + 2(Gp&1) + (p>>1); swz chunk ^ ((row&3)<<2); +8(p&1)
for base in (0, 8, 16, 24, 32, 40, 48, 56):
    for dt in range(4):
        ad = []
        for l in range(64):
            Gp, q, p, hi = l >> 4, (l & 15) >> 2, l & 3, l >> 5
            row = base + 4*hi + q
            ch = 4*dt + 2*(Gp & 1) + (p >> 1)
            ad.append(row*256 + 16*(ch ^ ((row & 3) << 2)) + 8*(p & 1))
        assert conflicts_tr(ad) == 1, (base, dt, conflicts_tr(ad))
print("conflict-free")
