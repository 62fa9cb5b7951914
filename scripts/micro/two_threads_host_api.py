"""Two host threads, a context each, calling the HOST-buffer entry points (agh_set_cloud + agh_find_hands) on the C2 cloud at once --
what two HandSearch objects on two threads of a node would do -- against one thread.  Prints clouds per second for 1 and 2 threads.
    python scripts/micro/two_threads_host_api.py [calls]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from agile_grasp_amd import binding, synthetic  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = synthetic.config("C2")


def worker(ctx, n, out, k):
    for _ in range(n):
        ctx.set_cloud(sc.xyz, sc.cam)
        h = ctx.find_hands(sc.samples)
    out[k] = len(h)


for n_thr in (1, 2, 3):
    ctxs = [binding.Context(sc.cam_origins) for _ in range(n_thr)]
    for c in ctxs:
        for _ in range(3):
            c.set_cloud(sc.xyz, sc.cam)
            c.find_hands(sc.samples)
    out = [0] * n_thr
    th = [threading.Thread(target=worker, args=(ctxs[k], calls, out, k)) for k in range(n_thr)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.3f ms per cloud (%.0f clouds/s), %d hypotheses each" % (n_thr, dt / (calls * n_thr) * 1e3, calls * n_thr / dt, out[0]), flush=True)
    for c in ctxs:
        c.close()
