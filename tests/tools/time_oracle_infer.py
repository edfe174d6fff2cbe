#!/usr/bin/env python
"""CPU oracle forward latency at bs=1 (config 4), the number quoted beside scripts/bench_infer.py.  Lives under tests/
because it runs the oracle (test infrastructure).    python tests/tools/time_oracle_infer.py [--height 1024 --width 2048]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--height', type=int, default=1024); ap.add_argument('--width', type=int, default=2048)
    a = ap.parse_args()
    import oracle
    from bench import NETWORK_ARCH, C_INDEX, make_args
    g = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
    torch.manual_seed(1)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    mo = oracle.ADD(NETWORK_ARCH, C_INDEX, g, 19, make_args(20), 0).eval()
    x = torch.randn(1, 3, a.height, a.width)
    with torch.no_grad():
        mo(x); t0 = time.perf_counter(); mo(x)
        print(json.dumps({'cpu_oracle_static_all_exits_ms': 1e3 * (time.perf_counter() - t0), 'input': [1, 3, a.height, a.width]}))


if __name__ == '__main__':
    main()
