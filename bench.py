#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): Cityscapes-shaped 1024x2048 images/sec, forward+backward+SGD of ADD
(searched-dense C=2, F=20, all exits active) at bs=2 per GPU, on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Synthetic data (N(0,1) images, uniform labels with 5 % ignore), random-init
weights of the named architecture (kaiming-scaled normals keyed on parameter names, so the CPU oracle gets the SAME
weights and the first-step loss can be asserted against it), fp32 storage; the wide k x k contractions run in the
arithmetic `--math` names (library default otherwise: f16x3 = fp32 operands as two fp16 terms under an exact power-of-two scale, three
product terms on the fp16 matrix pipe, fp32 accumulation — split error below the rounding noise of an fp32 accumulation chain, see
DESIGN.md §4.1; bf16x6, the default of rounds 2-5 = three bf16 terms, six product terms, is timed beside it as a second value).

Besides the contract's fields the line carries, measured in this run on rank 0 at N=1:
  roofline           dominant kernel (decoder 3x3) AND `segment`: the north-star "ASPP + cell forward" segment, eval and
                     train mode, as fractions of the HBM roofline (SURVEY §8(d) algorithmic bytes) and of the MFMA peak
  step_frac_mfma     whole-step algorithmic FLOP/s over the fp32 matrix peak
  per_exit_ms        config 4: EDM-gated dynamic inference, early / final exit, 1024x2048 and 1025x2049, and the mean latency
                     at 0 / 50 / 100 % early exits
  drop_in            the same step through the reference's own call pattern (model(x); loss.backward(); torch.optim.SGD.step())
  ddp_path_world1    the N > 1 code path (SyncBN exchanges + bucketed gradient all-reduce through RCCL) timed at world_size 1
  cpu_baseline       the CPU oracle on the same inputs (bounded sample), whose first-step loss the GPU loss is asserted against
The four measurements beside the headline (segment, per_exit_ms, drop_in, ddp_path_world1) run in a child process of this file
(--extras-child): whatever happens to them, the contract line is printed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

NETWORK_ARCH = [1, 2, 2, 2, 3, 2, 2, 1, 1, 1, 1, 2]      # train.py:75-79 (searched-dense, C=2)
C_INDEX = [5]
PEAK_MFMA_F32_TFLOPS = 157.3                              # MI355X_MICROARCH.md: fp32 matrix peak (spec)
PEAK_MFMA_BF16_TFLOPS = 2500.0                            # dense bf16 peak (spec)
SUSTAINED_MFMA_BF16_TFLOPS = 1950.0                       # measured on this pool: register-resident 32x32x16 bf16 loop, toggling operands (profiles/r03_mfma_clock_probe.txt)
PEAK_HBM_GBS = 8000.0
WEIGHT_SEED = 1001
# SURVEY §8(d), per image, fp32, 1024x2048: inference algorithmic bytes / training-forward extra / forward conv FLOPs per segment
ALG_GB = {'stem': 0.630, 'cell': 1.275 + 0.144 + 0.598 + 0.271 + 0.880, 'low': 0.033, 'aspp': 0.069, 'decoder': 0.358}
TRAIN_EXTRA_GB = {'stem': 0.671, 'cell': 2.098, 'low': 0.013, 'aspp': 0.168, 'decoder': 0.268}
GFLOP = {'stem': 59.8, 'cell': 120.75, 'low': 0.63, 'aspp': 104.7, 'decoder': 169.75}


def make_args(F=20, B=5, sync_bn=False):
    from types import SimpleNamespace
    return SimpleNamespace(F=F, B=B, sync_bn=sync_bn)


def synthetic_batch(n, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, 3, h, w), generator=g)
    t = torch.randint(0, 19, (n, h, w), generator=g)
    t[torch.rand((n, h, w), generator=g) < 0.05] = 255
    return x.to(device), t.to(device)


def init_weights(model):
    """Random init of the architecture, keyed on parameter names (addk.synth.fill_params): conv ~ N(0, 2/fan_in) as
    kaiming_normal_ gives, BN gamma ~ 1, beta ~ 0 with a small spread.  Identical in the oracle child."""
    from addk.synth import fill_params
    return fill_params(model, WEIGHT_SEED)


def cpu_baseline(n, h, w):
    """The CPU oracle (PyTorch-CPU restatement pinned to the reference by tests/golden) timed on this box's host
    cores on the same workload: full fwd+bwd+SGD steps at bs=n (a bounded sample: the reference needs ~14 s per step on 8
    cores).  Runs in a child process with a time limit; if the full-size step does not finish the sample is shrunk to a
    quarter-size image and scaled by pixel count (said so in `sample`)."""
    import subprocess
    for hh, ww, limit in ((h, w, 150), (h // 2, w // 2, 100), (h // 4, w // 4, 60)):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', '--batch', str(n), '--height', str(hh),
                                '--width', str(ww)], capture_output=True, text=True, timeout=limit)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode == 0 and line:
                d = json.loads(line[-1])
                if (hh, ww) != (h, w):
                    d['value'] *= (hh * ww) / float(h * w)
                    d['sample'] += ' (scaled by pixel count to %dx%d)' % (h, w)
                    d.pop('first_step_loss', None)
                return d
        except subprocess.TimeoutExpired:
            sys.stderr.write('[bench] cpu baseline at %dx%d exceeded %ds\n' % (hh, ww, limit))
            continue
    return {'value': None, 'unit': 'images/sec', 'cores': None, 'kind': 'port', 'sample': 'cpu baseline did not finish in time'}


def _cpu_baseline_child(genotype, n, h, w):
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    cores = max(1, min(cores, 16))          # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    m = oracle.ADD(NETWORK_ARCH, C_INDEX, genotype, 19, make_args(), 0)
    init_weights(m)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)
    x, t = synthetic_batch(n, h, w, 1, 'cpu')
    losses = []

    def step():
        ys = m(x)
        loss = oracle.cross_entropy_mean_exits(ys, t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    t0 = time.perf_counter()
    step()                                   # untimed warm-up (allocator, oneDNN primitive caches); its loss is step 0's
    warm = time.perf_counter() - t0
    k = max(2, min(5, int(16.0 / max(warm, 1e-3))))      # ~10-30 s of timed CPU work
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    dt = (time.perf_counter() - t0) / k
    res = {'value': n / dt, 'unit': 'images/sec', 'cores': cores, 'kind': 'port', 'first_step_loss': losses[0], 'second_step_loss': losses[1],
           'sample': '%d steps fwd+bwd+SGD after 1 warm-up step, bs=%d %dx%d, torch-CPU oracle, %.1f s per step' % (k, n, h, w, dt)}
    if (h, w) == (1024, 2048):
        # config 4 beside it: the oracle's own EDM-gated dynamic inference, early and final exit, bs = 1 (the early exit is the
        # SLOWER one in the reference itself: get_feature / dynamic_inference up-sample the gated feature 4x per side before ASPP,
        # SURVEY Q5, so ASPP runs on 16x the pixels)
        edm = oracle.EDM()
        from addk.synth import fill_params
        fill_params(edm, 701)
        m.eval(); edm.eval()
        x1 = x[:1]
        tt = {}
        with torch.no_grad():
            for name, thr in (('early', 1e9), ('final', -1e9)):
                m.dynamic_inference(x1, threshold=thr, confidence='edm', edm=edm)
                t0 = time.perf_counter()
                m.dynamic_inference(x1, threshold=thr, confidence='edm', edm=edm)
                tt[name + '_exit_s'] = time.perf_counter() - t0
        res['dynamic_inference_cpu'] = tt
    return res


def time_launch(cmd, reps=20):
    """Average device time of ONE launch of a plan command (the launch lists are re-ordered by dependency level after
    emission, so the command object, not its index, identifies it), with HIP events on the launch stream."""
    name, fn, args = cmd
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        fn(*args, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn(*args, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def _events_ms(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def segment_roofline(model, x, mode, reps=8):
    """North-star segment: the forward launch list of `model` in `mode` ('eval' | 'train'), timed per tagged segment with HIP
    events on the launch stream (single stream: true per-segment durations), plus the whole forward as the model runs it
    (level-ordered, batched, two streams).  Fractions: SURVEY §8(d) algorithmic bytes / 8 TB/s and conv FLOPs / fp32 matrix peak."""
    import addk.plan as P
    dev = x.device
    n = x.shape[0]
    model.train(mode == 'train')
    g = P.Graph(dev, mode == 'train', False, None)
    act, inref = g.input_nchw(x)
    inref.bind(x)
    model.emit(g, act)
    g.finalize()
    st = torch.cuda.current_stream().cuda_stream
    g.run(g.fwd, st); g.run(g.fwd, st)
    runs, cur = [], None
    for i, c in enumerate(g.fwd):
        if cur is None or c.tag != cur[0]:
            cur = [c.tag, i, i + 1]; runs.append(cur)
        else:
            cur[2] = i + 1
    tot = {}
    for tag, i0, i1 in runs:
        tot[tag] = tot.get(tag, 0.0) + _events_ms(lambda: g.run(g.fwd[i0:i1], st), reps)
    whole = _events_ms(lambda: g.run(g.fwd, st), reps)
    nl = len(g.fwd)
    del g
    g2 = P.Graph(dev, mode == 'train', False, None)
    act2, inref2 = g2.input_nchw(x)
    inref2.bind(x)
    model.emit(g2, act2)
    g2.reorder = True
    g2.finalize(int(os.environ.get('ADDK_STREAMS', '2')))
    for _ in range(2):
        g2.run_parallel(g2.fwd, None)
    whole2 = _events_ms(lambda: g2.run_parallel(g2.fwd, None), reps)
    # the SAME segment on the list the model really runs (level-ordered, batched, two streams): the cell + ASPP commands of that list,
    # re-scheduled on their own (their inputs — stem outputs, frozen-BN coefficients, packed weights — are resident from the runs above)
    seg_tags = {'cell', 'aspp'}
    tags_of = lambda c: getattr(c, 'tags', None) or {c.tag}
    segc = [c for c in g2.fwd if tags_of(c) & seg_tags]
    # a level-merged launch carries the tags of ALL its members: the segment figure is only the segment's if no selected launch holds work of
    # another segment (stem / low-level / decoder) and no cell / ASPP member hides in a launch that was dropped
    mixed = [(c.name, sorted(t for t in tags_of(c) if t)) for c in segc if not tags_of(c) <= seg_tags]
    assert not mixed, 'segment_roofline: merged launches mix segments: %r' % mixed[:4]
    for c in segc:
        c.event = None
    P.schedule(segc, g2.nstreams)
    for c in segc:
        c.event = torch.cuda.Event() if c.event else None
    for _ in range(2):
        g2.run_parallel(segc, None)
    seg2 = _events_ms(lambda: g2.run_parallel(segc, None), reps)
    extra = TRAIN_EXTRA_GB if mode == 'train' else dict.fromkeys(ALG_GB, 0.0)
    ms = tot.get('cell', 0.0) + tot.get('aspp', 0.0)
    gb = n * (ALG_GB['cell'] + ALG_GB['aspp'] + extra['cell'] + extra['aspp'])
    gf = n * (GFLOP['cell'] + GFLOP['aspp'])
    out = {'mode': mode, 'aspp_plus_cell_ms': ms, 'alg_GB': gb, 'achieved_GBps': gb / ms * 1e3, 'frac_hbm': gb / ms * 1e3 / PEAK_HBM_GBS,
           'alg_GFLOP': gf, 'achieved_TFLOPs': gf / ms, 'frac_mfma_f32': gf / ms / PEAK_MFMA_F32_TFLOPS,
           'aspp_plus_cell_ms_two_streams_batched': seg2, 'frac_hbm_two_streams_batched': gb / seg2 * 1e3 / PEAK_HBM_GBS,
           'segment_launches': sum(i1 - i0 for tag, i0, i1 in runs if tag in ('cell', 'aspp')), 'segment_launches_batched': len(segc),
           'forward_ms_single_stream': whole, 'forward_ms_two_streams_batched': whole2, 'launches': nl, 'launches_batched': len(g2.fwd),
           'segments_ms': {k: round(v, 4) for k, v in tot.items() if k}}
    del g2
    torch.cuda.empty_cache()
    return out


def per_exit_latency(model, dev, reps=12):
    """Config 4 (eval.py:195-230): EDM-gated dynamic inference at bs=1; per-exit latency as the reference measures it
    (synchronize + perf_counter around the whole call), at both shapes, and the mean at 0 / 50 / 100 % early exits."""
    from addk.modeling.ADD import EDM
    from addk.synth import fill_params
    edm = EDM()
    fill_params(edm, 701)
    edm.to(dev).eval()
    model.eval()
    out = {}
    with torch.no_grad():
        for h, w in ((1024, 2048), (1025, 2049)):
            x = synthetic_batch(1, h, w, 31, dev)[0]          # seeded (the gate is forced either way; the latency does not depend on the data)
            res = {}
            for name, thr in (('early', 1e9), ('final', -1e9)):
                for _ in range(4):
                    model.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)
                ts = [model.dynamic_inference(x, threshold=thr, confidence='edm', edm=edm)[2] for _ in range(reps)]
                res[name + '_exit_ms'] = 1e3 * float(np.median(ts))
            e, f = res['early_exit_ms'], res['final_exit_ms']
            res['mean_ms_at_early_exit_fraction'] = {'0%': f, '50%': 0.5 * (e + f), '100%': e}
            res['fps_at_early_exit_fraction'] = {'0%': 1e3 / f, '50%': 2e3 / (e + f), '100%': 1e3 / e}
            out['%dx%d' % (h, w)] = res
            plans = model._plans()
            for k in [k for k in plans if k[0] == 'dyn']:
                del plans[k]
            torch.cuda.empty_cache()
    return out


def drop_in_step(model, x, t, steps=5):
    """The reference's own call pattern (train.py:227-240) on the addk modules: forward through nn.Module.__call__, the
    drop-in CrossEntropyLoss per exit, autograd backward, torch.optim.SGD.  No TrainStep, no whole-step graph."""
    from addk.loss import CrossEntropyLoss
    model.train()
    crit = CrossEntropyLoss(ignore_index=255)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)

    def step():
        ys = model(x)
        loss = sum(crit(y, t) for y in ys) / len(ys)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    for _ in range(4):          # two eager calls, the third captures the plan's forward / backward hipGraphs (module.Plan), the fourth replays
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {'ms_per_step': dt * 1e3, 'images_per_sec': x.shape[0] / dt, 'steps': steps,
            'what': 'outputs = model(image); loss = mean_i CE(outputs[i], target); loss.backward(); torch.optim.SGD.step()'}


def ddp_path_world1(genotype, a, x, t, dev, steps=10):
    """The N > 1 code path timed on this one GPU: SynchronizedBatchNorm2d + every statistics exchange and the bucketed gradient
    all-reduce issued through RCCL at world_size 1 (zero link time: what remains is the launch structure the exchanges force
    on the step), captured in the hipGraph like the local step.  The driver's multi-GPU runs measure the real thing."""
    import torch.distributed as dist
    from addk import parallel
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    dist.init_process_group(backend='nccl', init_method='env://')
    ts, out = None, {}
    try:
        # twice on one process group: the statistics exchanges through the small-message mailboxes (csrc/comm.hip, the default when its
        # start-up self-test against the stock collective passes) and through stock RCCL all_reduce (ADDK_COMM_SMALL=0)
        for key, small in (('mailbox', '1'), ('rccl', '0')):
            os.environ['ADDK_COMM_SMALL'] = small
            comm = parallel.init_sync_bn(force=True)
            m = ADD(NETWORK_ARCH, C_INDEX, genotype, 19, make_args(a.F, sync_bn=True), 0)
            init_weights(m)
            m.to(dev)
            ts = TrainStep(m, tuple(x.shape), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, sync_comm=comm)
            ts.load_batch(x, t)
            for _ in range(3):
                ts.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ts.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            comm.check()
            nall = sum(getattr(c, 'members', 1) for c in ts.g.fwd + ts.g.bwd if getattr(c, 'name', '') in ('allreduce', 'allreduce_packed'))
            ncmd = sum(1 for c in ts.g.fwd + ts.g.bwd if getattr(c, 'name', '') in ('allreduce', 'allreduce_packed', 'grad_allreduce'))
            res = {'ms_per_step': dt * 1e3, 'images_per_sec': x.shape[0] / dt, 'hip_graph': ts.graph is not None,
                   'collectives_per_step': ncmd, 'exchanged_statistics_vectors': nall, 'loss': float(ts.loss.item()),
                   'statistics_exchange': 'addk_comm_allreduce (mailboxes, %d launches so far)' % comm.small.calls if comm.small is not None else 'torch.distributed all_reduce (RCCL)'}
            if key == 'mailbox':
                out = dict(res, what='world_size 1, exchanges forced: SyncBN statistics all-reduces (one per dependency level) + bucketed gradient all-reduce')
            else:
                out['stock_all_reduce'] = res
            ts.close()
            ts = None
            del m
            torch.cuda.synchronize()
            parallel.disable_sync_bn()
            torch.cuda.empty_cache()
        return out
    finally:
        # teardown in dependency order: the captured graph holds RCCL kernel nodes of this communicator and the
        # gradient buckets hold Work handles on it — release those, drain the device, THEN destroy the group
        os.environ.pop('ADDK_COMM_SMALL', None)
        if ts is not None:
            ts.close()
        ts = None
        torch.cuda.synchronize()
        parallel.disable_sync_bn()
        dist.destroy_process_group()


def timed_region(ts, steps, warmup, world, rank, barrier, max_over_ranks, contract_line, json_out, no_graph=False):
    """The contract's measurement and the whole N > 1 control flow, independent of what `ts` is (tests/test_parallel_gloo.py drives it
    with a stub step on two gloo ranks): W untimed warm-up steps, EXACTLY K steps between barrier + device synchronisation on both
    sides, the MAXIMUM over ranks.  With world > 1 the step runs on the eager launch list (the safe default: no RCCL call inside a
    capture has been seen on a multi-GPU box by the builder); the whole-step capture — 40 vs 45 ms at world 1 — is then TRIED under a
    watchdog: a refused capture falls back collectively inside TrainStep; if the captured steps do not finish within the limit, rank 0
    prints the eager line with `capture_hang: true` and every rank leaves with exit code 3 (a hang is not a successful run).
    Returns (seconds for K steps, {mode: ms per step}, first two warm-up losses)."""
    def timed(k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            ts.step()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    losses = []
    for i in range(warmup):
        ts.step()
        if i < 2:
            losses.append(float(ts.loss.item()))
    dt = timed(steps)
    modes = {('hip_graph' if ts.graph is not None else 'eager_list'): dt / steps * 1e3}
    try_cap = os.environ.get('ADDK_BENCH_TRY_CAPTURE', '1')
    if (world > 1 or try_cap == 'force') and ts.graph is None and ts.has_coll and not no_graph and try_cap != '0':
        import threading
        eager_line = contract_line(dt, False, 'whole-step capture with RCCL collectives did not finish in time: eager launch list')
        phase = ['capture']

        def bail():
            # a hang is NOT a successful run: the eager measurement is valid and is printed, but the line says so in machine-readable
            # fields and every rank leaves with a distinct exit code (3) while the stuck capture / replay is still in flight
            if rank == 0:
                eager_line['capture_hang'] = True
                eager_line['capture_phase'] = phase[0]
                json_out.write(json.dumps(eager_line) + '\n')
                json_out.flush()
            sys.stderr.write('[bench] rank %d: capture watchdog fired in phase %r\n' % (rank, phase[0])); sys.stderr.flush()
            os._exit(3)
        timer = threading.Timer(float(os.environ.get('ADDK_BENCH_CAPTURE_LIMIT', '120')), bail)
        timer.daemon = True
        timer.start()
        try:
            ts.enable_capture()
            ts.step()
            if ts.graph is not None:
                for i in range(2):
                    phase[0] = 'replay warm-up step %d' % i
                    ts.step()
                phase[0] = 'timed replay of %d steps' % steps
                dt2 = timed(steps)
                modes['hip_graph'] = dt2 / steps * 1e3
                if dt2 < dt:
                    dt = dt2
                else:
                    ts.use_graph = False
            else:
                modes['hip_graph'] = 'capture refused: eager launch list on every rank'
        except Exception as e:
            sys.stderr.write('[bench] capture attempt failed: %r\n' % (e,))
        timer.cancel()
    return dt, modes, losses


def alt_math_step(mode, genotype, a, x, t, dev, steps=10):
    """Second value beside the headline (NOT the headline): the same step in another arithmetic of the wide contractions — `bf16x6`, the
    six-term split-bf16 form that was the default through round 5, beside today's `f16x3`."""
    import addk
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    prev = addk.get_precision()
    try:
        addk.set_precision(mode)
        m = ADD(NETWORK_ARCH, C_INDEX, genotype, 19, make_args(a.F), 0)
        init_weights(m)
        m.to(dev)
        ts = TrainStep(m, tuple(x.shape), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True)
        ts.load_batch(x, t)
        losses = []
        for i in range(3):
            ts.step()
            if i < 2:
                losses.append(float(ts.loss.item()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ts.close()
        return {'ms_per_step': dt * 1e3, 'images_per_sec': x.shape[0] / dt, 'first_step_losses': losses,
                'what': 'the same step with --math %s; fp32 storage and accumulation' % mode}
    finally:
        addk.set_precision(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--F', type=int, default=20)
    ap.add_argument('--genotype', default='autodeeplab/genotype', help="cell genotype under searched_arch/ (config 5: --F 40 --genotype 40_5e_38_lr/genotype_1)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the segment roofline / per-exit latency / drop-in measurements')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--math', choices=['fp32', 'f16x3', 'bf16x6', 'bf16x3', 'tail_x3'], default=None,
                    help='arithmetic of the wide k x k contractions (default: the library default, see addk.get_precision())')
    ap.add_argument('--cpu-baseline-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--extras-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--force-sync', action='store_true', help='rehearse the N>1 path (RCCL SyncBN + gradient all-reduce) at world_size 1')
    a = ap.parse_args()
    genotype = np.load(os.path.join(ROOT, 'searched_arch', a.genotype + '.npy'))
    if a.cpu_baseline_child:
        print(json.dumps(_cpu_baseline_child(genotype, a.batch, a.height, a.width)))
        return

    # exactly ONE line on stdout: RCCL prints a version banner to stdout when a communicator is created, so the process's
    # stdout is pointed at stderr for the whole run and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(out_fd, 'w')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import addk
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    from addk import parallel
    if a.math:
        addk.set_precision(a.math)
    math = addk.get_precision()
    comm = None
    if world > 1 or a.force_sync:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group(backend='nccl', init_method='env://')
        comm = parallel.init_sync_bn(force=a.force_sync)
    model = ADD(NETWORK_ARCH, C_INDEX, genotype, 19, make_args(a.F, sync_bn=comm is not None), 0)
    init_weights(model)
    model.to(dev)
    parallel.broadcast_params(model)
    n, h, w = a.batch, a.height, a.width
    if a.extras_child:
        # the measurements beside the headline number, in a process of their own (see `extras` below): one JSON object on the saved stdout
        import faulthandler
        faulthandler.enable(all_threads=True)         # a fatal signal leaves the Python stacks of every thread on stderr, which the parent keeps
        x, t = synthetic_batch(n, h, w, 1 + rank, dev)
        res = {}

        def guarded(name, fn):
            sys.stderr.write('[bench extras] begin %s\n' % name); sys.stderr.flush()     # which extra was running, should the process die
            try:
                res[name] = fn()
            except Exception as e:
                res[name] = {'error': str(e).splitlines()[0] if str(e) else repr(e)}
            sys.stderr.write('[bench extras] end %s\n' % name); sys.stderr.flush()
        with torch.no_grad():
            guarded('segment_eval', lambda: segment_roofline(model, x, 'eval'))
            guarded('segment_train', lambda: segment_roofline(model, x, 'train'))
        guarded('per_exit_ms', lambda: per_exit_latency(model, dev))
        guarded('drop_in', lambda: drop_in_step(model, x, t))
        if math == 'f16x3':
            guarded('bf16x6', lambda: alt_math_step('bf16x6', genotype, a, x, t, dev))
        json_out.write(json.dumps(res) + '\n')
        json_out.flush()                               # before the last, least proven extra: a crash there keeps the others
        guarded('ddp_path_world1', lambda: ddp_path_world1(genotype, a, x, t, dev))
        json_out.write(json.dumps({'ddp_path_world1': res['ddp_path_world1']}) + '\n')
        json_out.flush()
        return
    ts = TrainStep(model, (n, 3, h, w), lr=0.05, momentum=0.9, weight_decay=4e-5, nesterov=True, sync_comm=comm,
                   use_graph=False if a.no_graph else None)
    x, t = synthetic_batch(n, h, w, 1 + rank, dev)
    ts.load_batch(x, t)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(d):
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d = float(tt.item())
        return d

    def contract_line(d, graph, note=None):
        line = {'metric': 'Cityscapes 1024x2048 images/sec fwd+bwd @ bs=2/GPU', 'value': world * n * a.steps / d, 'unit': 'images/sec',
                'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': d / a.steps * 1e3, 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32' if math == 'fp32' else 'f32 storage and accumulation; k x k contractions as %s %s MFMA products' % (math, 'split-fp16' if math == 'f16x3' else 'split-bf16'),
                'data': 'synthetic',
                'config': {'workload': 'ADD F=%d searched_arch/%s C=2 all exits, %dx%d bs=%d/GPU fwd+CE+bwd+SGD(nesterov)' % (a.F, a.genotype, h, w, n),
                           'global_batch': world * n, 'parallelism': 'dp%d' % world, 'sync_bn': comm is not None,
                           'sync_bn_exchange': None if comm is None else ('addk mailboxes (csrc/comm.hip)' if comm.small is not None else 'torch.distributed all_reduce (RCCL)'),
                           'hip_graph': bool(graph), 'math': math}}
        if note:
            line['note'] = note
        return line

    dt, modes, losses = timed_region(ts, a.steps, a.warmup, world, rank, barrier, max_over_ranks, contract_line, json_out, no_graph=a.no_graph)
    loss = float(ts.loss.item())
    if comm is not None:
        comm.check()              # a timed-out mailbox exchange raises here (bounded polls: csrc/comm.hip), on every rank
    if rank != 0:
        ts.close()
        torch.cuda.synchronize()
        parallel.disable_sync_bn()
        torch.distributed.destroy_process_group()
        return
    ms = dt / a.steps * 1e3
    value = world * n * a.steps / dt

    # roofline of the dominant kernel: the heaviest dense-conv launch (decoder 3x3 304->256) on the halo-patch kernel
    convs = [m for m in ts.g.meta if m['kind'] == 'conv_fwd']
    top = max(convs, key=lambda m: m['flops'])
    tk = time_launch(top['cmd'])
    fwd_flops = sum(m['flops'] for m in convs)
    halo = bool(top.get('halo'))
    terms = {'fp32': 1, 'bf16x6': 6, 'f16x3': 3, 'tail_x3': 3}[math] if halo else 1          # (the dominant launch is a decoder conv: a tail launch)
    if terms == 1:
        peak, kern = PEAK_MFMA_F32_TFLOPS, ('conv3_kernel' if halo else 'conv_kernel') + ' (v_mfma_f32_16x16x4_f32)'
        peak_note = 'fp32 matrix peak'
    else:
        peak = PEAK_MFMA_BF16_TFLOPS / terms
        el = 'bf16' if terms == 6 else 'f16'
        kern = 'conv3b_kernel (v_mfma_f32_32x32x16_%s, %d %s product terms per fp32 product)' % (el, terms, 'bf16' if terms == 6 else 'fp16')
        peak_note = 'dense %s MFMA peak %.0f TFLOP/s / %d matrix instructions per fp32 multiply-add' % (el, PEAK_MFMA_BF16_TFLOPS, terms)
    ach = top['flops'] / tk / 1e12
    # HBM-side traffic of that launch: PMC passes cannot run inside this process; the committed rocprofv3 --pmc measurement of the
    # same kernel and shape (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes: scripts/refresh_profiles_r04.sh) is quoted, labelled
    traffic, tsrc = None, None
    tname = 'pmc_traffic_decoder_conv3b_f16x3' if math == 'f16x3' else 'pmc_traffic_decoder_conv3b'
    tpath = next((q for q in (os.path.join(ROOT, 'profiles', 'r0%d_%s.json' % (r, tname)) for r in (5, 4, 3, 2)) if os.path.exists(q)),
                 os.path.join(ROOT, 'profiles', 'r02_pmc_traffic_decoder_conv3b.json'))
    if halo and math in ('bf16x6', 'f16x3') and (n, h, w, a.F, a.genotype) == (2, 1024, 2048, 20, 'autodeeplab/genotype') and os.path.exists(tpath):
        try:
            traffic, tsrc = json.load(open(tpath)).get('traffic_bytes_per_launch'), 'profiles/%s (rocprofv3 --pmc passes of the same kernel and shape; not measured in this run)' % os.path.basename(tpath)
        except Exception:
            pass
    roof = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
            'traffic_source': tsrc,
            'peak_note': peak_note, 'executed_matrix_TFLOPs': ach * terms, 'frac_of_fp32_matrix_peak': ach / PEAK_MFMA_F32_TFLOPS,
            'kernel': kern, 'launch_ms': tk * 1e3,
            'launch_shape_NHWCinCoutKSD': list(top['shape']), 'algorithmic_gflop_per_launch': top['flops'] / 1e9,
            'algorithmic_bytes_per_launch': top['bytes']}
    if terms > 1:
        # what this part SUSTAINS on a register-resident v_mfma_f32_32x32x16_bf16 loop (no memory traffic, every SIMD busy): the shader
        # clock drops from 2.4 to ~1.95-2.05 GHz under that load (s_memtime / s_memrealtime, scripts/mfma_clock_probe.hip) — reported
        # beside `frac`, which stays priced against the guide's nominal peak
        roof['measured_sustained_matrix_TFLOPs'] = SUSTAINED_MFMA_BF16_TFLOPS
        roof['frac_of_measured_sustained'] = ach * terms / SUSTAINED_MFMA_BF16_TFLOPS
        roof['sustained_source'] = 'profiles/r03_mfma_clock_probe.txt (1.90-2.07 PFLOP/s at 1.95-2.05 GHz; operands toggling every iteration: 1.93)'
    step_tflops = 3 * fwd_flops / (dt / a.steps) / 1e12
    out = contract_line(dt, ts.graph is not None and ts.use_graph)
    out.update({'ms_per_step_by_mode': modes,
           'loss': loss, 'first_step_losses': losses,
           'step_algorithmic_tflop': 3 * fwd_flops / 1e12,
           'step_tflops_per_gpu': step_tflops, 'step_frac_mfma': step_tflops / PEAK_MFMA_F32_TFLOPS,
           'plan_device_gb': ts.nbytes / 1e9,
           'roofline': roof})
    default_cfg = (n, h, w, a.F, a.genotype) == (2, 1024, 2048, 20, 'autodeeplab/genotype')
    extras = world == 1 and not a.no_extras and comm is None and default_cfg
    if extras:
        # Segment roofline, per-exit latency, the drop-in path and the N > 1 code path at world 1 run in a CHILD process: they capture
        # hipGraphs of their own, start RCCL and replay captured collectives — one run in ~30 of round 3's bench ended without its JSON line
        # while they still ran in-process (no stderr kept).  Round 4 fixed the one ordering fault the teardown paths had (communicator
        # destroyed under a live captured graph and live Work handles: TrainStep.close); the child stays because a long
        # side measurement has no business in the process that owns the contract line, and its rc / last marker / stderr tail are now
        # fields of that line
        ts.close()
        del ts
        torch.cuda.empty_cache()
        import subprocess
        merged = {}
        try:
            cmd = [sys.executable, os.path.abspath(__file__), '--extras-child', '--batch', str(n), '--height', str(h), '--width', str(w),
                   '--F', str(a.F), '--genotype', a.genotype, '--math', math]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            for line in r.stdout.strip().splitlines():
                try:
                    merged.update(json.loads(line))
                except ValueError:
                    pass
            # the child's fate goes INTO the line (negative rc = killed by that signal); its stderr carries the begin/end marker of every
            # extra and, after a fatal signal, faulthandler's stacks
            out['extras_rc'] = r.returncode
            marks = [l for l in r.stderr.splitlines() if l.startswith('[bench extras]')]
            out['extras_last_marker'] = marks[-1] if marks else None
            if r.returncode != 0:
                out['extras_stderr_tail'] = r.stderr[-2000:]
                merged.setdefault('ddp_path_world1', {'error': 'extras child exited with code %d' % r.returncode})
                sys.stderr.write(r.stderr[-6000:])
        except subprocess.TimeoutExpired as e:
            merged = {'error': 'extras child exceeded 900 s'}
            out['extras_rc'] = 'timeout'
            err = e.stderr.decode(errors='replace') if isinstance(e.stderr, bytes) else (e.stderr or '')
            out['extras_stderr_tail'] = err[-2000:]
        if 'segment_eval' in merged and 'segment_train' in merged and 'error' not in merged['segment_eval'] and 'error' not in merged['segment_train']:
            roof['segment'] = {'name': 'ASPP + cell forward, 1024x2048 bs=2 (BASELINE north_star: >= 0.60 of the HBM roofline)',
                               'eval': merged['segment_eval'], 'train': merged['segment_train'], 'bound': 'hbm', 'peak_GBps': PEAK_HBM_GBS,
                               'frac': merged['segment_eval']['frac_hbm']}
        else:
            roof['segment'] = {'error': str(merged.get('segment_eval', merged.get('error', 'extras child produced no output')))}
        for key in ('per_exit_ms', 'drop_in', 'ddp_path_world1'):
            out[key] = merged.get(key, {'error': merged.get('error', 'extras child produced no output')})
        if isinstance(merged.get('bf16x6'), dict) and 'ms_per_step' in merged['bf16x6']:
            out['ms_per_step_by_mode']['bf16x6'] = merged['bf16x6']['ms_per_step']          # a second value; `value` / `dtype` are the f16x3 ones
            out['bf16x6'] = merged['bf16x6']
    if not a.no_cpu_baseline and world == 1 and default_cfg:
        cb = cpu_baseline(n, h, w)
        out['cpu_baseline'] = cb
        if cb.get('dynamic_inference_cpu') and 'per_exit_ms' in out:
            d = cb.pop('dynamic_inference_cpu')
            out['per_exit_ms']['cpu_oracle_1024x2048'] = {'early_exit_ms': 1e3 * d['early_exit_s'], 'final_exit_ms': 1e3 * d['final_exit_s'],
                                                         'note': 'the reference arithmetic itself: its early exit costs more than its final exit (SURVEY Q5: 4x up-sampling per side before ASPP)'}
        ref = cb.get('first_step_loss')
        if ref is not None and losses:
            rel = abs(losses[0] - ref) / abs(ref)
            out['first_step_loss_vs_cpu_oracle'] = {'gpu': losses[0], 'cpu_oracle': ref, 'rel_diff': rel}
            assert rel <= 1e-3, 'first-step loss %.7f differs from the CPU oracle %.7f on the same inputs (rel %.2e)' % (losses[0], ref, rel)
            if len(losses) > 1 and cb.get('second_step_loss') is not None:
                # the second step's loss has been through one whole backward pass + SGD update of every parameter: the full-size
                # backward / optimizer check of the headline shape.  Bound from the reference's OWN arithmetic at this shape
                # (tests/golden/grads64.npz `full_train_sentinels`, the real reference in fp32 and in double at 2x1024x2048, train mode): its
                # fp32 gradients sit 4.4e-2..6.5e-2 (max-abs) / 4.8e-2..5.0e-2 (rms) from its fp64 ones on the stems and early cells —
                # train-mode BatchNorm amplifies rounding by 10^4..10^5 — so two fp32 realisations of the step, each correct to rounding,
                # disagree by ~7 % of what the update does to the loss: 0.07 x 0.053 (the loss moves 3.30 -> 3.25) / 3.25 = 1.2e-3 relative
                # (measured over builds 6.8e-5 .. 1.27e-3).  Asserted at 2.5e-3 = twice that spread; a missing or wrong update is 1.6e-2
                # away (the loss does not move without the step).  tests/test_gpu_round5.py holds the gradients themselves to that fixture
                ref2 = cb['second_step_loss']
                rel2 = abs(losses[1] - ref2) / abs(ref2)
                out['first_step_loss_vs_cpu_oracle']['second_step'] = {'gpu': losses[1], 'cpu_oracle': ref2, 'rel_diff': rel2}
                assert rel2 <= 2.5e-3, 'second-step loss %.7f differs from the CPU oracle %.7f (rel %.2e): backward / SGD parity' % (losses[1], ref2, rel2)
    json_out.write(json.dumps(out) + '\n')
    json_out.flush()
    if comm is not None:
        if 'ts' in locals():
            ts.close()
        torch.cuda.synchronize()
        parallel.disable_sync_bn()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
