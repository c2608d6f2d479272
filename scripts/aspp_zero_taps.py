"""Host replay of the tap tables of conv_igemm_p8_kernel for the ASPP dilated 3x3 convolutions (aspp.py:17-24: 2048 -> 256, d = 12 / 24 / 36 on the
OS-8 map): how many of the EXECUTED K steps multiply an A operand that is entirely zero padding -

  * for the whole 256-pixel tile                (what a per-tile tap list can skip: the kernel already skips whole kernel ROWS this way),
  * for each 64-pixel quarter / 16-pixel group  (what a column-aware skip could at best remove: a quarter is the finest unit the wave tiling could
                                                  skip without restructuring; 16 pixels = one MFMA row block, the theoretical limit),

and for the weight gradient (conv_wgrad_p8_kernel: a tap is a k-column tile, the reduction runs over pixels): the share of 32-pixel steps whose
X rows are all padding for the tile's tap.  Tiles are 256 CONSECUTIVE output pixels of the [N][H][W] map (conv_p8.hip tile_info), K step = 64
channels of one tap; forward and data gradient have the same geometry (the data gradient is the forward with flipped taps).
Usage: python scripts/aspp_zero_taps.py [H=65] [N=32]   ->  stdout (committed as profiles/r06_aspp_zero_tap_share.txt)
"""
import sys

import numpy as np

H = int(sys.argv[1]) if len(sys.argv) > 1 else 65
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, BM, R = H, 256, 3
M = N * H * W
m = np.arange(M)
hd, wd = (m % (H * W)) // W, m % W
print(f"ASPP dilated 3x3, 2048 -> 256, map {H} x {W}, {N} images (M = {M}), tiles of {BM} consecutive pixels: {-(-M // BM)} tiles per launch")
print("nominal = 9 taps for every tile; executed = taps of the kernel rows that are live for the tile (the round-3 row skip)")
tot = {}
for d in (12, 24, 36):
    pad = d
    ntile = -(-M // BM)
    nominal = executed = 0
    zero_tile = zero_q64 = zero_g16 = 0.0      # in units of (tile, tap) steps
    for t in range(ntile):
        sl = slice(t * BM, min((t + 1) * BM, M))
        h, w = hd[sl], wd[sl]
        npx = h.size
        nominal += 9
        # live kernel rows of the tile (conv_p8.hip:141-156: a row is live when ANY pixel of the tile has a source row inside the image)
        rows = [r for r in range(R) if ((h + r * d - pad >= 0) & (h + r * d - pad < H)).any()]
        for r in rows:
            for s in range(R):
                ok = (h + r * d - pad >= 0) & (h + r * d - pad < H) & (w + s * d - pad >= 0) & (w + s * d - pad < W)
                executed += 1
                zero_tile += 0.0 if ok.any() else 1.0
                q = [ok[i:i + 64] for i in range(0, npx, 64)]
                zero_q64 += sum(0 if x.any() else 1 for x in q) / len(q)
                g = [ok[i:i + 16] for i in range(0, npx, 16)]
                zero_g16 += sum(0 if x.any() else 1 for x in g) / len(g)
    live = ((hd[:, None, None] + np.arange(3)[None, :, None] * d - pad >= 0) & (hd[:, None, None] + np.arange(3)[None, :, None] * d - pad < H)
            & (wd[:, None, None] + np.arange(3)[None, None, :] * d - pad >= 0) & (wd[:, None, None] + np.arange(3)[None, None, :] * d - pad < W))
    pix_share = live.mean()
    print(f"d = {d:2d}: executed {executed} of {nominal} nominal (tile, tap) steps = {executed / nominal:.3f} (row skip removes {1 - executed / nominal:.3f}); "
          f"in-range (pixel, tap) pairs: {pix_share:.3f} of nominal")
    print(f"        of the EXECUTED steps, A entirely padding for: the whole tile {zero_tile / executed:.4f} | 64-pixel quarters {zero_q64 / executed:.4f} | "
          f"16-pixel groups {zero_g16 / executed:.4f}   (useful share of executed MACs: {pix_share * nominal / executed:.3f})")
    # weight gradient: per tap (= per k-column tile row), 32-pixel steps with no in-range pixel
    z32 = n32 = 0
    for r in range(R):
        for s in range(R):
            ok = live[:, r, s]
            k = (M + 31) // 32
            okp = np.zeros(k * 32, bool)
            okp[:M] = ok
            any32 = okp.reshape(k, 32).any(1)
            z32 += (~any32).sum()
            n32 += k
    print(f"        weight gradient: 32-pixel steps whose X rows are all padding for the tile's tap: {z32 / n32:.4f} of the steps")
    tot[d] = (executed, nominal, zero_tile, zero_q64, zero_g16, z32 / n32)
ex = sum(v[0] for v in tot.values())
print(f"all three launches: executed {ex} steps; all-zero for the whole tile {sum(v[2] for v in tot.values()) / ex:.4f}, per 64-pixel quarter "
      f"{sum(v[3] for v in tot.values()) / ex:.4f}, per 16-pixel group {sum(v[4] for v in tot.values()) / ex:.4f}")
