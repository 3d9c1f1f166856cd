"""Test double of gbp_amd.sharded._HipShard on CPU tensors + the C oracle (TEST INFRASTRUCTURE: used by
tests/test_sharded_gloo.py and by the launch-path dry run of bench.py in tests/test_bench_launch.py)."""
import numpy as np
import torch


class OracleAdapter:
    """What gbp_amd.sharded._HipShard is on the GPU, on CPU tensors + the C oracle."""

    def __init__(self, problem):
        from oracle import oracle
        self.engine = oracle.OracleShard.from_problem(problem)
        self.partial_doubles = oracle.OracleShard.PARTIAL_DOUBLES

    def new_buffer(self, n):
        return torch.empty(n, dtype=torch.float64)

    def begin(self, partial, with_messages, robustify, local_relin):
        partial.copy_(torch.from_numpy(self.engine.shard_begin_host(with_messages, robustify, local_relin)))

    def end(self, gathered, world):
        self.engine.shard_end_host(gathered.numpy(), world)

    def to_tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a))

    def sync(self):
        pass


def factory(problem):
    return OracleAdapter(problem)
