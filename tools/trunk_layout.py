"""LDS layout of the split-precision SNP trunk kernel (k5_trunk_h3): checks every DS access pattern of the kernel against
the gfx950 bank model (tools/lds_sim.py) and prints the constant tables nc_cnn.hip embeds.  Design tool, not product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from lds_sim import cycles

lane = np.arange(64); g = lane >> 4; c16 = lane & 15
RX = 57                    # X: slots (16 B = one pixel) per padded image row
R1, PL1 = 44, 221          # A1: slots per row / per 8-channel chunk plane
R2, PL2 = 26, 119          # A2

# conv1 K groups: 4 taps (dy, dx) per group, lane group g reads tap g; (g0,g1) and (g2,g3) are the conflict-free pairs
C1_GROUPS = [
    [(2, 0), (2, 1), (2, 3), (2, 4)],
    [(0, 4), (2, 2), (1, 4), (3, 2)],
    [(0, 2), (0, 3), (1, 2), (1, 3)],
    [(4, 2), (4, 3), (4, 4), (4, 4)],
    [(0, 0), (0, 1), (1, 0), (1, 1)],
    [(3, 0), (3, 1), (3, 3), (3, 4)],
    [(4, 0), (4, 1), (4, 0), (4, 1)],
]
# conv3: lane -> output position (y, x) of the two 16-position tiles (hill-climbed, conflict-free with R2/PL2 above)
C3_T0 = [(2, 4), (1, 4), (1, 3), (0, 4), (1, 1), (1, 0), (2, 8), (0, 8), (0, 1), (2, 1), (0, 7), (2, 2), (0, 3), (2, 5), (2, 0), (2, 3)]
C3_T1 = [(1, 2), (1, 5), (1, 4), (0, 3), (0, 2), (2, 7), (1, 6), (0, 0), (2, 7), (1, 8), (0, 4), (0, 6), (0, 5), (1, 7), (2, 6), (1, 4)]


def conv2_pos(tile):
    if tile < 4:
        return np.full(16, tile), np.arange(16)
    c = np.arange(16)
    return c & 3, 16 + (c >> 2)


def main():
    tot = {}
    # conv1 reads
    n = 0
    for t in range(13):
        p = np.minimum(16 * t + c16, 204); h = p // 41; w = p % 41
        for grp in C1_GROUPS:
            dy = np.array([grp[i][0] for i in g]); dx = np.array([grp[i][1] for i in g])
            n += cycles("b128", ((h + dy) * RX + w + dx) * 16)
    tot["conv1 reads (one plane)"] = (n, 13 * 7 * 4)
    # conv1 writes: b64, chunk = 2b + (g>>1), half = g&1
    n = 0
    for t in range(13):
        p = 16 * t + c16; h = np.minimum(p // 41, 4); w = np.where(p < 205, p % 41, 41 + p - 205)
        n += cycles("w64", (((g >> 1) * PL1 + h * R1 + w) * 8 + (g & 1) * 4) * 2)
    tot["conv1 writes (per block, plane)"] = (n, 13 * 4)
    # conv2 reads
    n = 0
    for t in range(5):
        y, x = conv2_pos(t)
        for G in range(9):
            idx = 4 * G + g; tap = idx // 6; ch = idx % 6
            n += cycles("b128", (ch * PL1 + (y[c16] + tap // 3) * R1 + 2 * x[c16] + tap % 3) * 16)
    tot["conv2 reads (one plane, one tn)"] = (n, 5 * 9 * 4)
    n = 0
    for t in range(5):
        y, x = conv2_pos(t)
        n += cycles("w64", (((g >> 1) * PL2 + y[c16] * R2 + x[c16]) * 8 + (g & 1) * 4) * 2)
    tot["conv2 writes (plane)"] = (n, 5 * 4)
    n = 0
    for tl in (C3_T0, C3_T1):
        ys = np.array([a for a, b in tl]); xs = np.array([b for a, b in tl])
        for G in range(6):
            n += cycles("b128", (g * PL2 + (ys[c16] + G // 3) * R2 + 2 * xs[c16] + G % 3) * 16)
    tot["conv3 reads (one plane, one n-block)"] = (n, 2 * 6 * 4)
    # staging: thread = pixel, one b128 per plane at slot (h+2)*RX + w+2
    n = 0
    for wv in range(4):
        t = wv * 64 + lane; act = t < 205; tt = np.minimum(t, 204)
        n += cycles("w128", ((tt // 41 + 2) * RX + tt % 41 + 2) * 16, act)
    tot["staging writes (plane)"] = (n, 4 * 8)
    for k, (a, b) in tot.items():
        print("%-40s %5d cycles (conflict-free %d)" % (k, a, b))
    seen = set(); out = []
    for (y, x) in C3_T0 + C3_T1:
        out.append((y * R2 + 2 * x, (y * 9 + x) if (y, x) not in seen else -1)); seen.add((y, x))
    assert len(seen) == 27
    print("conv3 read slot :", ", ".join(str(a) for a, b in out))
    print("conv3 out index :", ", ".join(str(b) for a, b in out))
    print("conv1 tap slot offsets (dy*RX+dx) per group:", [[dy * RX + dx for dy, dx in grp] for grp in C1_GROUPS])


if __name__ == "__main__":
    main()
