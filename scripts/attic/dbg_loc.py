import sys, numpy as np
sys.path.insert(0, '.')
from agile_grasp_amd import binding, synthetic
for n, seed in ((60_000, 7), (60_000, 8), (200_000, 9), (120_000, 7), (90_000, 8), (300_000, 11), (700_000, 21)):
    rc = synthetic.make_raw_cloud(n, seed)
    c = binding.Context(rc.cam_origins)
    nv = c.preprocess(rc.xyz, rc.size_left, rc.workspace)
    samples = np.sort(np.random.default_rng(0).permutation(nv)[:150]).astype(np.int32)
    try:
        h = c.find_hands(samples); print(n, seed, "ok", nv, len(h), c.neighbor_counts()[0].max())
    except Exception as e:
        print(n, seed, "ERR", nv, str(e)[:60])
