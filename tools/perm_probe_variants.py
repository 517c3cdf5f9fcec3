#!/usr/bin/env python3
"""perm_probe specs for the two layout-permuting schemes on random realistic tile sets D (8 index bits outside the fast
window P* = 3-6,12-15): today's in-place pass on D; variant 1 = read the tile at P*, store its bits to D (the next
tile's bits come to P*); variant 2 = read the tile from D, store it at P* (fast write window)."""
import random
import subprocess
import sys
from pathlib import Path

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
P = [3, 4, 5, 6, 12, 13, 14, 15]
pool = [b for b in range(7, n) if b not in P]
rng = random.Random(5)
specs = []
sets = [sorted(rng.sample(pool, 8)) for _ in range(10)]
sets += [[7, 8, 9, 10, 11, 16, 17, 18], [16, 17, 18, 19, 20, 21, 22, 23], [7, 8, 9, 10, 16, 17, 18, 19]]
for D in sets:
    d = ",".join(map(str, D))
    swap = ",".join(f"{a}>{b},{b}>{a}" for a, b in zip(P, D))
    specs.append(f"R={d};inplace;name=inplace        D={d}")
    specs.append(f"R={','.join(map(str, P))};P={swap};name=v1 read P* -> D  D={d}")
    specs.append(f"R={d};P={swap};name=v2 read D -> P*  D={d}")
    specs.append(f"R={d};P={swap};order=3;name=v2 + output-ordered tiles D={d}")
    specs.append(f"R={','.join(map(str, P))};P={swap};order=3;name=v1 + output-ordered tiles D={d}")
exe = Path(__file__).resolve().parent / "perm_probe"
subprocess.run([str(exe), str(n)] + specs, check=False)
