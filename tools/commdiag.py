"""RCCL through the C ABI in a process without torch: which libraries end up mapped (development aid)."""
import numpy as np
from gpras_amd.comm import Communicator
try:
    c = Communicator.bootstrap(0, rank=0, world=1)
    print("OK gathered", c.all_gather(np.arange(3.0)))
    c.close()
except Exception as e:  # noqa: BLE001
    print("FAILED", e)
print("loaded:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if any(k in ln for k in ("hsa", "amdhip", "rccl"))}))
