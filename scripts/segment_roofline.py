#!/usr/bin/env python
"""North-star target: fraction of the MI355X HBM roofline reached by the "ASPP + cell" forward segment of ADD
(config 2: F=20, C=2, 1024x2048, bs=2).  Times the forward launch list per segment with HIP events on the launch stream
and divides SURVEY §8(d)'s algorithmic bytes by it.

    python scripts/segment_roofline.py [--mode eval|train] [--batch 2]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch

# SURVEY §8(d), per image, fp32, 1024x2048: inference algorithmic bytes / training-forward extra / forward FLOPs
ALG_GB = {'stem': 0.630, 'cell': 1.275 + 0.144 + 0.598 + 0.271 + 0.880, 'low': 0.033, 'aspp': 0.069, 'decoder': 0.358}
TRAIN_EXTRA_GB = {'stem': 0.671, 'cell': 2.098, 'low': 0.013, 'aspp': 0.168, 'decoder': 0.268}
GFLOP = {'stem': 59.8, 'cell': 120.75, 'low': 0.63, 'aspp': 104.7, 'decoder': 169.75}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='eval'); ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--height', type=int, default=1024); ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=10)
    a = ap.parse_args()
    import addk, addk.plan as P
    from addk.modeling.ADD import ADD
    from bench import NETWORK_ARCH, C_INDEX, make_args
    dev = torch.device('cuda:0')
    g0 = np.load(os.path.join(ROOT, 'searched_arch', 'autodeeplab', 'genotype.npy'))
    torch.manual_seed(1)
    m = ADD(NETWORK_ARCH, C_INDEX, g0, 19, make_args(20), 0).to(dev)
    m.train(a.mode == 'train')
    x = torch.randn(a.batch, 3, a.height, a.width, device=dev)
    g = P.Graph(dev, a.mode == 'train', False, None)
    act, inref = g.input_nchw(x)
    inref.bind(x)
    m.emit(g, act)
    g.finalize()
    st = torch.cuda.current_stream().cuda_stream
    g.run(g.fwd, st); g.run(g.fwd, st); torch.cuda.synchronize()
    # contiguous runs of one tag
    runs, cur = [], None
    for i, c in enumerate(g.fwd):
        if cur is None or c.tag != cur[0]:
            cur = [c.tag, i, i + 1]; runs.append(cur)
        else:
            cur[2] = i + 1
    tot = {}
    for tag, i0, i1 in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(a.reps):
            g.run(g.fwd[i0:i1], st)
        e1.record(); torch.cuda.synchronize()
        tot[tag] = tot.get(tag, 0.0) + e0.elapsed_time(e1) / a.reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.reps):
        g.run(g.fwd, st)
    e1.record(); torch.cuda.synchronize()
    whole = e0.elapsed_time(e1) / a.reps
    n = a.batch
    # the same forward as the model runs it: launch list re-ordered by dependency level, same-level launches batched, two streams
    g2 = P.Graph(dev, a.mode == 'train', False, None)
    act2, inref2 = g2.input_nchw(x)
    inref2.bind(x)
    m.emit(g2, act2)
    g2.reorder = True
    g2.finalize(2)
    for _ in range(2):
        g2.run_parallel(g2.fwd, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.reps):
        g2.run_parallel(g2.fwd, None)
    e1.record(); torch.cuda.synchronize()
    whole2 = e0.elapsed_time(e1) / a.reps
    out = {'mode': a.mode, 'input': [n, 3, a.height, a.width], 'forward_ms_total': whole, 'launches': len(g.fwd),
           'forward_ms_total_level_batched_2streams': whole2, 'launches_level_batched': len(g2.fwd), 'segments': {}}
    for tag, ms in tot.items():
        key = tag if tag in ALG_GB else None
        if key is None:
            out['segments'][tag] = {'ms': ms}
            continue
        gb = n * (ALG_GB[key] + (TRAIN_EXTRA_GB[key] if a.mode == 'train' else 0.0))
        out['segments'][tag] = {'ms': ms, 'alg_GB': gb, 'GBps': gb / ms * 1e3, 'frac_hbm_8TBps': gb / ms * 1e3 / 8000.0,
                                'TFLOPs': n * GFLOP[key] / ms, 'frac_mfma_f32_157TF': n * GFLOP[key] / ms / 157.3}
    ms = tot.get('cell', 0) + tot.get('aspp', 0)
    gb = n * (ALG_GB['cell'] + ALG_GB['aspp'] + ((TRAIN_EXTRA_GB['cell'] + TRAIN_EXTRA_GB['aspp']) if a.mode == 'train' else 0.0))
    gf = n * (GFLOP['cell'] + GFLOP['aspp'])
    out['aspp_plus_cell'] = {'ms': ms, 'alg_GB': gb, 'GBps': gb / ms * 1e3, 'frac_hbm_8TBps': gb / ms * 1e3 / 8000.0,
                             'TFLOPs': gf / ms, 'frac_mfma_f32_157TF': gf / ms / 157.3,
                             'note': 'segment is MFMA-limited at fp32: %.1f GFLOP need >= %.2f ms at 157.3 TF/s' % (gf, gf / 157.3)}
    print(json.dumps(out))

if __name__ == '__main__':
    main()
