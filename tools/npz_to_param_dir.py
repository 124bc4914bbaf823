#!/usr/bin/env python3
"""tests/golden/resnet20_cifar10.npz -> a directory of the reference's parameter text files (one number per line),
the input of `tests/cpp/cnn_test bench <dir>` and of cnn.import_parameters_cifar10."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
with np.load(os.path.join(root, "tests", "golden", "resnet20_cifar10.npz"), allow_pickle=False) as z:
    for k in z.files:
        np.savetxt(os.path.join(out, k + ".txt"), np.asarray(z[k], dtype=np.float64).ravel(), fmt="%.9e")
print(out)
