"""Replay one data set of tools/fuzz_pipeline.py: python tools/repro_pipeline_seed.py <master seed> <data-set seed>; prints, for the engine
under a few environment variants, which reads differ from the oracle pipeline and where.  GPU box only."""
import os
import pathlib
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def find(master, want):
    rng = random.Random(master)
    for _ in range(100000):
        seed = rng.getrandbits(30)
        rate = rng.choice([0.03, 0.08, 0.12, 0.12, 0.16])
        n_reads, glen = rng.choice([12, 24, 36]), rng.choice([4000, 7000])
        prm = dict(min_support=rng.choice([2, 3, 4]), max_support=rng.choice([5, 20, 1000]), window_size=rng.choice([300, 500, 500, 700]),
                   mer_size=rng.choice([7, 8, 9, 9, 10]), common_kmers=rng.choice([4, 8]), min_anchors=rng.choice([2, 10]),
                   solid_thresh=rng.choice([2, 4]), window_overlap=rng.choice([20, 50, 80]), max_msa=rng.choice([10, 50, 150]))
        trim = rng.random() < 0.7
        wpb = rng.choice([1, 64, 100000])
        if seed == want:
            return seed, rate, n_reads, glen, prm, trim, wpb
    raise SystemExit("seed not found in that stream")


if __name__ == "__main__":
    if len(sys.argv) > 3:  # child: one run under the caller's environment
        import json
        from consent_amd.pipeline import correct_reads
        fa, paf, prm, trim, wpb = json.loads(sys.argv[3])
        print(json.dumps(correct_reads(fa, paf, None, do_trim=trim, windows_per_batch=wpb, **prm)))
        sys.exit(0)
    import json
    from test_gpu_pipeline import make_dataset, oracle_pipeline
    seed, rate, n_reads, glen, prm, trim, wpb = find(int(sys.argv[1]), int(sys.argv[2]))
    d = pathlib.Path(tempfile.mkdtemp())
    fa, paf = make_dataset(d, seed, n_reads=n_reads, glen=glen, rate=rate)
    want = dict(oracle_pipeline(fa, paf, do_trim=trim, **prm))
    print("data set", seed, rate, n_reads, glen, prm, trim, wpb, "oracle reads", len(want))
    for env in ({}, {"CW_TIER_H": "2"}, {"CW_TIER_H": "1"}, {"CW_STITCH_NARROW": "1"}, {"CW_TIER_H": "2", "CW_STITCH_NARROW": "1"}):
        out = subprocess.run([sys.executable, __file__, "x", "y", json.dumps([fa, paf, prm, trim, wpb])], capture_output=True, text=True, env=dict(os.environ, CW_DRIVER_STATS="1", CW_DRIVER_TIMING="2", **env))
        if out.returncode != 0:
            print(env, "FAILED", out.stderr[-500:])
            continue
        got = dict(json.loads(out.stdout.splitlines()[-1]))
        if out.stderr.strip():
            print("    stderr:", out.stderr.strip()[-600:])
        bad = [n for n in sorted(set(got) | set(want)) if got.get(n) != want.get(n)]
        print(env, "reads", len(got), "different:", bad)
        for n in bad[:3]:
            a, b = got.get(n, ""), want.get(n, "")
            x = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            print("   ", n, "lengths", len(a), len(b), "first difference at", x, repr(a[max(0, x - 20):x + 30]), repr(b[max(0, x - 20):x + 30]))
