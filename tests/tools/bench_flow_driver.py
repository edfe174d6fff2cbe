"""One rank of a world-2 gloo rehearsal of bench.py's N > 1 control flow (bench.timed_region) with a STUB step — no GPU, no kernels:
what is rehearsed is the flow the first real multi-GPU run will take (eager timing, the capture trial under the watchdog, the refused
capture, the hang) and what each rank prints and exits with.  Started twice by tests/test_parallel_gloo.py with RANK / WORLD_SIZE / MASTER_* set.
    python tests/tools/bench_flow_driver.py {capture_ok|capture_refused|capture_hang}"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch                                  # noqa: E402
import torch.distributed as dist              # noqa: E402

import bench                                  # noqa: E402


class StubStep:
    """The TrainStep surface timed_region uses: step(), loss, graph, has_coll, use_graph, enable_capture()."""

    def __init__(self, mode, rank):
        self.mode, self.rank = mode, rank
        self.graph, self.has_coll, self.use_graph = None, True, False
        self.loss = torch.tensor([3.25])
        self.want_capture = False
        self.steps = 0

    def enable_capture(self):
        self.want_capture = True

    def step(self):
        self.steps += 1
        t = torch.ones(1)
        dist.all_reduce(t)                     # every step is collective, as the SyncBN / gradient exchanges make the real one
        if self.want_capture and self.graph is None:
            if self.mode == 'capture_ok':
                self.graph, self.use_graph = object(), True
            elif self.mode == 'capture_refused':
                self.want_capture = False      # TrainStep._capture decided collectively: eager on every rank
            elif self.mode == 'capture_hang':
                time.sleep(60)                 # a capture that never returns (the watchdog's limit is 1 s in the test)
        time.sleep(0.004 if self.graph is not None else 0.006)
        return self.loss


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ts = StubStep(mode, rank)

    def barrier():
        dist.barrier()

    def max_over_ranks(d):
        t = torch.tensor([d], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def contract_line(d, graph, note=None):
        line = {'metric': 'stub', 'value': world * 2 * 5 / d, 'n_gpus': world, 'steps': 5, 'config': {'hip_graph': bool(graph)}}
        if note:
            line['note'] = note
        return line
    dt, modes, losses = bench.timed_region(ts, 5, 2, world, rank, barrier, max_over_ranks, contract_line, sys.stdout)
    if rank == 0:
        out = contract_line(dt, ts.graph is not None and ts.use_graph)
        out['ms_per_step_by_mode'] = modes
        out['first_step_losses'] = losses
        sys.stdout.write(json.dumps(out) + '\n')
        sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
