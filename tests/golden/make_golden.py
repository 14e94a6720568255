"""Generate the golden vectors under tests/golden/ from the CPU oracle.

The reference (gpflow/tensorflow) cannot be imported in this image, so
these vectors pin *this repository's restatement*: a later edit of oracle/ or of the HIP path that
moves any number is caught.  Inputs are regenerated from seeds (gpras_amd.synth), so the files
hold only parameters and expected outputs.

    python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gpras_amd.synth import make_regression  # noqa: E402
from oracle import exact, gpras_oracle, sgpr  # noqa: E402
from oracle import kernels as kn  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
N, D, NT = 256, 4, 64
W_VAR, W_LEN, W_NOISE = 0.3, 0.15, -1.2
W_LEN_ARD = np.array([0.15, -0.2, 0.4, 0.0])


def main():
    x, y, xs = make_regression(N, D, n_outputs=1, n_test=NT, config=1, unit=0)
    yc = y[:, 0]
    out = {"n": N, "d": D, "n_test": NT, "config": 1, "unit": 0, "w_var": W_VAR, "w_len": W_LEN, "w_noise": W_NOISE, "w_len_ard": W_LEN_ARD}
    z_km = gpras_oracle.create_inducing(x, 32, "kmeans")
    z_grid = gpras_oracle.create_inducing(x, 32, "grid")
    out["z_kmeans32"] = z_km
    out["z_grid32"] = z_grid
    for kernel in kn.KERNEL_NAMES:
        for tag, wl in (("iso", W_LEN), ("ard", W_LEN_ARD)):
            # sparse, M = 32 (k-means centres) and M = N (Z = X)
            for mtag, z in (("m32", z_km), ("m256", x)):
                loss, g = sgpr.loss_and_grad(kernel, x, yc, z, W_VAR, wl, W_NOISE)
                key = f"sgpr_{kernel}_{tag}_{mtag}"
                out[key + "_loss"] = loss
                out[key + "_g_var"] = g["variance"]
                out[key + "_g_len"] = g["lengthscales"]
                out[key + "_g_noise"] = g["noise"]
                if mtag == "m32":
                    out[key + "_g_Z"] = g["Z"]
                from oracle import transforms as tr

                v, l, s = tr.constrain(W_VAR, wl, W_NOISE)
                mean, var = sgpr.predict(kernel, x, yc, z, float(v), l if np.ndim(l) else float(l), float(s), xs)
                out[key + "_mean"] = mean
                out[key + "_var"] = var
            loss, g = exact.loss_and_grad(kernel, x, yc, W_VAR, wl, W_NOISE)
            key = f"exact_{kernel}_{tag}"
            out[key + "_loss"] = loss
            out[key + "_g_var"] = g["variance"]
            out[key + "_g_len"] = g["lengthscales"]
            out[key + "_g_noise"] = g["noise"]
            v, l, s = tr.constrain(W_VAR, wl, W_NOISE)
            mean, var = exact.predict(kernel, x, yc, float(v), l if np.ndim(l) else float(l), float(s), xs)
            out[key + "_mean"] = mean
            out[key + "_var"] = var
    np.savez_compressed(os.path.join(HERE, "gp_golden_n256_d4.npz"), **out)
    print("wrote", os.path.join(HERE, "gp_golden_n256_d4.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
