#!/usr/bin/env python3
"""Writes tests/golden/resnet20_cifar10.npz from the reference's pretrained ResNet-20 parameter files
(/root/reference/pretrained_parameters/resnet20_new/*.txt, the data examples/cnn.cu loads through
import_parameters_cifar10): the same numbers as float32 arrays under the same names.  Data only; run in the
container that holds the reference, the GPU box uses the committed file."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phantom_fhe_amd import cnn

# optional: <layer_num> <output.npz> for the deeper networks (not committed: tests/golden/_local/ is git-ignored)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 20
src = "/root/reference/pretrained_parameters/resnet%d_new" % layers
params = cnn.import_parameters_cifar10(src, layers)
lossless = all(np.array_equal(v.astype(np.float32).astype(np.float64), v) for v in params.values())
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resnet20_cifar10.npz")
np.savez_compressed(out, **{k: v.astype(np.float32) for k, v in params.items()})
print(out, os.path.getsize(out), "bytes;", sum(v.size for v in params.values()), "parameters; float32 lossless:", lossless)
