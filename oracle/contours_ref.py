"""ORACLE — test infrastructure only.

Pure-Python restatement of cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE) as the reference calls it
(TreeDetection/prediction.py:232-234). OpenCV is un-vendored and absent here: this follows its published
algorithm (Suzuki-Abe border following + the icvFetchContour tracing loop + CHAIN_APPROX_SIMPLE point
emission; tree order = depth-first, most recently found sibling first). **Parity unpinned** against cv2 itself;
pinned by closed-form cases in tests/test_contours.py. Small images only (Python loops).
"""
from __future__ import annotations

from typing import List

import numpy as np

DX = (1, 1, 0, -1, -1, -1, 0, 1)
DY = (0, -1, -1, -1, 0, 1, 1, 1)


def find_contours(mask: np.ndarray) -> List[np.ndarray]:
    h, w = mask.shape
    F = np.zeros((h + 2, w + 2), dtype=np.int64)
    F[1:-1, 1:-1] = (np.asarray(mask) != 0)
    nodes = [None, {"parent": 0, "hole": True, "pts": [], "children": []}]
    nbd = 1
    for y in range(1, h + 1):
        lnbd = 1
        for x in range(1, w + 1):
            f = F[y, x]
            if f == 0:
                continue
            outer = f == 1 and F[y, x - 1] == 0
            hole = (not outer) and f >= 1 and F[y, x + 1] == 0
            if outer or hole:
                if hole and f > 1:
                    lnbd = int(f)
                nbd += 1
                ln = nodes[lnbd]
                parent = ln["parent"] if ln["hole"] == hole else lnbd
                if parent == 0:
                    parent = 1
                pts = []
                px, py = x - 1, y - 1
                cx, cy = x, y
                s_end = s = 0 if hole else 4
                while True:
                    s = (s - 1) & 7
                    i1 = (cx + DX[s], cy + DY[s])
                    if F[i1[1], i1[0]] != 0 or s == s_end:
                        break
                if F[i1[1], i1[0]] == 0:
                    F[cy, cx] = -nbd
                    pts.append((px, py))
                else:
                    i0 = (cx, cy)
                    i3 = i0
                    prev_s = s ^ 4
                    while True:
                        s_end = s
                        i4 = None
                        while s < 15:
                            s += 1
                            i4 = (i3[0] + DX[s & 7], i3[1] + DY[s & 7])
                            if F[i4[1], i4[0]] != 0:
                                break
                        s &= 7
                        if ((s - 1) & 0xFFFFFFFF) < s_end:
                            F[i3[1], i3[0]] = -nbd
                        elif F[i3[1], i3[0]] == 1:
                            F[i3[1], i3[0]] = nbd
                        if s != prev_s:
                            pts.append((px, py))
                            prev_s = s
                        px += DX[s]
                        py += DY[s]
                        if i4 == i0 and i3 == i1:
                            break
                        i3 = i4
                        s = (s + 4) & 7
                nodes.append({"parent": parent, "hole": hole, "pts": pts, "children": []})
                nodes[parent]["children"].append(len(nodes) - 1)
            fv = F[y, x]
            if fv != 1:
                lnbd = int(abs(fv))
    out = []
    stack = list(nodes[1]["children"])
    while stack:
        n = stack.pop()
        out.append(np.asarray(nodes[n]["pts"], dtype=np.int32).reshape(-1, 2))
        stack.extend(nodes[n]["children"])
    return out
