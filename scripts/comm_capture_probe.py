#!/usr/bin/env python
"""Which ingredient aborts the capture of a step that holds mailbox exchanges (tests/test_gpu_train.py::test_syncbn_captured_step_world1...)?
One variant per child process:  python scripts/comm_capture_probe.py            (driver)
                                python scripts/comm_capture_probe.py child MODE (MODE: small|rccl  x  tl|global  x  grad|nograd)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def child(mode):
    import faulthandler; faulthandler.enable()
    import torch, torch.distributed as dist
    small, cap, probe = mode.split('-')
    os.environ['ADDK_COMM_SMALL'] = '1' if small == 'small' else '0'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29655', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1)
    dev = torch.device('cuda:0')
    import addk
    from addk import parallel
    if probe == 'kernel':
        # no network: the mailbox kernel + one RCCL all_reduce in one captured graph
        sc = parallel.SmallComm.create(device=dev)
        v = torch.ones(64, device=dev, dtype=torch.float64); w = torch.ones(1 << 16, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local' if cap == 'tl' else 'global'):
            sc.allreduce(v, torch.cuda.current_stream().cuda_stream)
            work = dist.all_reduce(w, async_op=True); work.wait()
            sc.allreduce(v, torch.cuda.current_stream().cuda_stream)
        g.replay(); torch.cuda.synchronize()
        print('ok', sc.check())
        return
    from addk.modeling.ADD import ADD
    from addk.train import TrainStep
    from _util import ARCH_C2, GENOTYPE_AUTODEEPLAB, fill_params, make_args
    m = ADD(ARCH_C2['network_arch'], ARCH_C2['C_index'], GENOTYPE_AUTODEEPLAB, 19, make_args(4, sync_bn=True), 0)
    fill_params(m, 600); m.to(dev)
    comm = parallel.init_sync_bn(force=True)
    ts = TrainStep(m, (2, 3, 65, 129), sync_comm=comm)
    if cap == 'global':
        ts.has_coll = False            # _capture then uses the global error mode (and would re-raise a refused capture)
    x = torch.randn(2, 3, 65, 129, device=dev); t = torch.randint(0, 19, (2, 65, 129), device=dev)
    ts.load_batch(x, t)
    l = [ts.step().item() for _ in range(3)]
    print('ok', l, ts.graph is not None, comm.small is not None)
    ts.close(); torch.cuda.synchronize(); parallel.disable_sync_bn(); dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'child':
        child(sys.argv[2]); sys.exit(0)
    for mode in ('rccl-tl-step', 'small-tl-kernel', 'small-global-kernel', 'small-tl-step', 'small-global-step'):
        env = dict(os.environ)
        if mode.startswith('small') and mode.endswith('step'):
            env['AMD_LOG_LEVEL'] = '1'
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', mode], capture_output=True, text=True, timeout=300, env=env)
        print('=== %s: rc %d' % (mode, r.returncode)); print(r.stdout[-600:]); print(r.stderr[-2500:] if r.returncode else '')
        sys.stdout.flush()
