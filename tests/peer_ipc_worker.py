"""One rank of tests/test_peer_ipc_gpu.py: a separate PROCESS that shares GPU 0 with the other ranks.  The process group is gloo
(ranks on one GPU cannot form an RCCL group); the camera partial sums travel through the peer-store exchange: hipIpc-mapped
mailboxes, direct stores, arrival polls inside the kernels."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    rank, world, port, out_dir, n_sweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    os.environ.setdefault('GBP_PEER_TIMEOUT_MS', '8000')
    import torch
    import torch.distributed as dist
    from gbp_amd.sharded import ShardedBA
    from gbp_amd.synthetic import make_synthetic
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    p = make_synthetic(n_cams=int(sys.argv[6]), n_lmks=int(sys.argv[7]), obs_per_lmk=10, seed=2)
    if os.environ.get('GBP_TEST_SELFTEST_ONLY'):             # test_peer_selftest_names_the_pair_that_failed: rank 1 never sends its probe row
        from gbp_amd._capi import GbpError
        from gbp_amd.engine import BAEngine
        e = BAEngine.from_problem(p)
        handles = [None] * world
        dist.all_gather_object(handles, e.peer_export(world))
        e.peer_connect(rank, handles)
        dist.barrier()
        if rank == 0:
            try:
                e.peer_selftest(300)
                print('SELFTEST-PASSED')
            except GbpError as ex:
                print('SELFTEST-FAILED', ex)
        else:
            print('SELFTEST-SKIPPED')
        dist.barrier()
        e.close()
        dist.destroy_process_group()
        return
    fused = {'0': False, '1': True}.get(os.environ.get('GBP_TEST_FUSED', ''), None)      # (None: the library picks the sweep)
    g = ShardedBA(p, device=0, exchange='peer', fused=fused)
    assert g.library_loop and g.exchange == 'peer', (g.library_loop, g.exchange)
    g.generate_priors_var(50.0)
    g.update_all_beliefs()
    dist.barrier()
    g.iterate(n_sweeps)
    g.sync()                                                  # raises if a finish wave timed out
    ce, cl = g.camera_beliefs()
    rng, le, ll = g.local_landmark_beliefs()
    info = g.comm_info()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), ce=ce, cl=cl, le=le, ll=ll, lo=rng[0], hi=rng[1], are=g.are(), F=g.F,
             kind=info['kind'], n_ranks=info['n_ranks'])
    dist.barrier()
    g.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
