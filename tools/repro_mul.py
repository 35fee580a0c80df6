"""Ad-hoc (GPU box): smallest exercise of the variable-base scalar multiplication, growing in size."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from oracle import bbs
suite = bbs.SUITES[sys.argv[1] if len(sys.argv) > 1 else "bls12_381"]
c = suite.curve
rng = random.Random(1)
eng = pc.make_engine(c.name, pc.gens_for(suite, 3), b"x", None)
for n in (1, 3, 70):
    pts = [[c.g1_mul(c.g1, rng.randrange(1, 1 << 30))] for _ in range(n)]
    sc = [[rng.randrange(c.r)] for _ in range(n)]
    out, st = eng.g1_msm_batch([[0, 0, 0, 0]] * n, pts, sc)
    assert list(st) == [1] * n
    for i in range(min(n, 4)):
        assert out[i] == c.g1_mul(pts[i][0], sc[i][0]), (n, i)
    print("msm n=%d ok" % n, flush=True)
