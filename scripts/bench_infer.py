#!/usr/bin/env python
"""BASELINE config 4: EDM-gated dynamic inference on one MI355X, bs=1, Cityscapes-shaped input.
Reports per-exit forward latency (early exit / final exit) and static all-exit forward latency (the CPU oracle's forward
is timed by tests/tools/time_oracle_infer.py: only tests/ may touch oracle/).
    python scripts/bench_infer.py [--height 1024 --width 2048] [--reps 20]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=1024); ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=20); ap.add_argument('--math', default='fp32')
    a = ap.parse_args()
    import addk
    from addk.modeling.ADD import ADD, EDM
    from bench import NETWORK_ARCH, C_INDEX, make_args
    addk.set_precision(a.math)
    dev = torch.device('cuda:0')
    g = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
    torch.manual_seed(1)
    m = ADD(NETWORK_ARCH, C_INDEX, g, 19, make_args(20), 0).to(dev).eval()
    edm = EDM().to(dev).eval()
    x = torch.randn(1, 3, a.height, a.width, device=dev)
    res = {}
    with torch.no_grad():
        for name, thr in (('early_exit', 1e9), ('final_exit', -1e9)):
            for _ in range(3):
                m.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)
            ts = []
            for _ in range(a.reps):
                y, ee, secs, conf = m.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)
                ts.append(secs)
            res[name] = {'ms_median': 1e3 * float(np.median(ts)), 'ms_min': 1e3 * float(np.min(ts)), 'exit': int(ee)}
        for _ in range(3):
            m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps):
            m(x)
        torch.cuda.synchronize()
        res['static_all_exits'] = {'ms_mean': 1e3 * (time.perf_counter() - t0) / a.reps}
    out = {'metric': 'per-exit forward ms, EDM-gated dynamic inference, bs=1', 'input': [1, 3, a.height, a.width], 'dtype': a.math,
           'results': res}
    print(json.dumps(out))

if __name__ == '__main__':
    main()
