"""bench.py's stage agreement over two gloo processes: a stage that fails on one rank makes every
rank give the path up, and the ranks stay in step for the collectives that follow."""
import os
import tempfile

import torch.multiprocessing as mp


def _rank_main(rank, world, rdzv, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    import bench

    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)

    class StageError(Exception):
        pass

    log = []

    def fine():
        log.append("fine")

    def fails_on_rank_1():
        if rank == 1:
            raise StageError("rank 1 could not map a mailbox")
        log.append("would go on")

    r1 = bench.stage_agreed(dist, fine, StageError)
    r2 = bench.stage_agreed(dist, fails_on_rank_1, StageError)
    # still in step: a collective after the failed stage completes on both ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(repr((r1, r2, t.item(), log)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failed_stage_is_given_up_by_every_rank():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, args=(2, os.path.join(d, "rdzv"), d), nprocs=2, join=True)
        got = [eval(open(os.path.join(d, f"r{r}.txt")).read()) for r in range(2)]
    assert got[0][0] == (True, None) and got[1][0] == (True, None)
    assert got[0][1] == (False, None)                                   # told by the agreement alone
    assert got[1][1] == (False, "rank 1 could not map a mailbox")      # its own error
    assert got[0][2] == 3.0 and got[1][2] == 3.0
    assert got[0][3] == ["fine", "would go on"] and got[1][3] == ["fine"]
