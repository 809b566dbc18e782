"""Which launch shapes get the median of 'first tile = 1e6, rest U(100, 200)' wrong?  Prints got / expected / path."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "nvidia-resiliency-ext_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import oracle
from nvrx_straggler.backend import get_backend
be = get_backend()
def run(m, counts, kinds):
    s = torch.from_numpy(m).cuda(); c = torch.from_numpy(counts.astype(np.int32)).cuda(); k = torch.from_numpy(kinds).cuda()
    return be.row_stats(s, c, k).cpu().numpy()
for stride in (20000, 32768, 33000, 36000, 40000, 40960, 45000, 49152, 60000, 65536):
    rng = np.random.default_rng(stride)
    rows = []
    for first in (1024, 2048, 4096, 8192):
        for hi in (1e6, 1.0):
            x = rng.uniform(100.0, 200.0, stride).astype(np.float32); x[:first] = hi; rows.append(x)
    m = np.stack(rows); R = len(rows)
    counts = np.full(R, stride, np.uint32); kinds = np.zeros(R, np.uint8)
    got = run(m, counts, kinds); exp = oracle.rows_stats(m, counts, kinds)
    bad = [(r, float(got[r, 2]), float(exp[r, 2]), int(got[r, 7])) for r in range(R) if got[r, 2] != np.float32(exp[r, 2])]
    print(stride, "paths", [int(p) for p in got[:, 7]], "BAD" if bad else "ok", bad, flush=True)
