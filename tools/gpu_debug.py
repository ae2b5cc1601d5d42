"""Run the engine and the oracle on the same synthetic windows and report the first differences (GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import consent_amd as ca  # noqa: E402
from consent_amd.engine import synth_host  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    maxmsa = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    prm = ca.Params(9, 4, 8, 2, maxmsa)
    batch = synth_host(ca.SynthSpec.pacbio(nw, depth))
    eng = ca.Engine(prm)
    got = eng.run(batch)
    print("timings", eng.timings())
    info = eng.win_info(nw)
    exp, stats = oracle_lib.oracle_run(prm, batch)
    bad = 0
    for w in range(nw):
        g, e = got.consensus(w), exp.consensus(w)
        sg, se = got.solid_kmers(w), exp.solid_kmers(w)
        ok = g == e and got.status[w] == exp.status[w] and np.array_equal(sg, se)
        if not ok:
            bad += 1
            print(f"window {w}: status gpu {got.status[w]} cpu {exp.status[w]} len {len(g)} vs {len(e)} solid {len(sg)} vs {len(se)} solid_eq {np.array_equal(sg, se)} info {info[w][:13]}")
            if g != e:
                i = next((i for i in range(min(len(g), len(e))) if g[i] != e[i]), min(len(g), len(e)))
                print("   first diff at", i, "\n   gpu", g[max(0, i - 30) : i + 40], "\n   cpu", e[max(0, i - 30) : i + 40])
    print(f"{nw - bad}/{nw} windows identical; oracle stats/window:", {k: v / nw for k, v in stats.items()})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
