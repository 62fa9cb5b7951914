"""Throughput of a batch of C5 clouds in one context (one launch set for all of them) against the single-cloud step.
    python scripts/batch_bench.py [--clouds 1,2,4,8] [--steps 30] [--classify]
Prints one JSON line per batch size: ms per batch, ms per cloud, hypotheses/s, per-kernel ms per batch."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from agile_grasp_amd import binding, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--classify", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the HIP-event pass (for rocprofv3 runs)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sizes = [int(v) for v in args.clouds.split(",")]
    scs = [synthetic.config(f"C5_{k}") for k in range(max(sizes))]
    z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
    for C in sizes:
        ctx = binding.Context(scs[0].cam_origins, profile=0)
        if args.classify:
            ctx.load_svm(z["w"], float(z["rho"]))
        off = np.zeros(C + 1, np.int64)
        off[1:] = np.cumsum([s.n for s in scs[:C]])
        xyz_t = torch.from_numpy(np.concatenate([s.xyz for s in scs[:C]])).to(dev)
        cam_t = torch.from_numpy(np.concatenate([s.cam for s in scs[:C]])).to(dev)
        samples = np.concatenate([s.samples + off[k] for k, s in enumerate(scs[:C])]).astype(np.int32)
        s_t = torch.from_numpy(samples).to(dev)
        S = samples.size
        out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
        nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
        keep_t = torch.zeros(8 * S, dtype=torch.uint8, device=dev)
        tstream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        st = tstream.cuda_stream

        def step():
            ctx.set_cloud_batch_torch(xyz_t, cam_t, off, stream=st)
            ctx.find_hands_torch(s_t, out_t, nout_t, stream=st)
            if args.classify:
                ctx.classify_torch(keep_t, stream=st)

        # (two untimed steps first: a cloud that needs the larger capacity classes switches them on -- AGH_ERR_RETRY -- and the
        # steps that follow run the context's final configuration; bench.py's settle)
        for _ in range(2):
            step()
            torch.cuda.synchronize()
            try:
                ctx.synchronize()
            except binding.AghError as e:
                if e.code != binding.AGH_ERR_RETRY:
                    raise
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ctx.synchronize()
        n_hyp = int(nout_t.item())
        k_ms = {}
        if not args.no_events:
            ctx.set_profile(1)
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            k_ms = {k: v / args.steps for k, v in ctx.timing().items()}
        print(json.dumps({"clouds": C, "samples": S, "hypotheses": n_hyp, "ms_per_batch": dt * 1e3, "ms_per_cloud": dt * 1e3 / C,
                          "hypotheses_per_s": n_hyp / dt, "kernel_ms_per_batch": k_ms}))
        ctx.close()


if __name__ == "__main__":
    main()
