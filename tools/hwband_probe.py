#!/usr/bin/env python3
"""HW pair units inside the static band (Batch::solveHwBanded): one case per subprocess, so that a device fault names its case.
usage: hwband_probe.py            (the list below)
       hwband_probe.py m T k seed (one case; prints OK / MISMATCH)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) == 5:
    import edlib_amd
    from edlib_amd import synth
    from oracle.oracle import load_ref, load_oracle
    m, T, k, seed = (int(x) for x in sys.argv[1:])
    q = synth.random_dna(seed, m).tobytes(); t = synth.random_dna(seed + 1, T).tobytes()
    chk = load_ref() or load_oracle()
    g = edlib_amd.align_pairs([q], [t], mode="HW", task="locations", k=k, raw=True)[0]
    w = chk.align(q, t, "HW", "locations", k)
    print("OK" if all(g[f] == w[f] for f in ("status", "editDistance", "endLocations", "startLocations")) else "MISMATCH %r %r" % (g["editDistance"], w["editDistance"]))
    sys.exit(0)
for m, T, k in ((6000, 9000, 64), (6000, 9000, 256), (6000, 9000, -1), (6000, 7000, 64), (6000, 8100, 64), (6000, 8100, 256), (9000, 12000, 256),
                (3000, 4500, 64), (3000, 5100, 64), (1000, 1200, 20)):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(m), str(T), str(k), "5"], capture_output=True, text=True, timeout=300)
    print(m, T, k, "rc=%d" % p.returncode, p.stdout.strip()[-80:], p.stderr.strip().splitlines()[-1][-120:] if p.returncode else "")
