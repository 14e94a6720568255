"""Worker for tests/test_parallel.py: one rank of a world-size-2 gloo job (CPU, oracle-backed engine)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist

    from gpras_amd import gpr, parallel
    from gpras_amd.synth import make_regression
    from test_host_logic import OracleBackend

    gpr.Engine = OracleBackend  # the HIP engine cannot run in the CPU container; host logic under test
    from oracle import kmeans as okm

    gpr.kmeans_centers = lambda x, m, device=0: okm.kmeans_centers(x, m)[0]  # likewise the device Lloyd iterations
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out_dir = sys.argv[1]
    x, y, xs = make_regression(80, 3, n_outputs=3, n_test=17, config=8, unit=0)
    g = parallel.ShardedGPRAS("Matern32")
    g.fit(x, y, 8, "kmeans", "adam", max_iter=4)
    mean, var = g.predict(xs)
    params = np.array([[m.variance, m.lengthscales, m.noise] for m in g.models])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), mean=mean, var=var, params=params, z=np.stack([m.Z for m in g.models]),
             owned=np.array(parallel.shard_units(3, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
