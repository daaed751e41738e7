#!/usr/bin/env python3
"""Golden `.dla` files for the attention-map export (gbm/classify.py:207-225).

The reference's `write_map` cannot be imported (gbm/classify.py pulls in modules that are not in the repository and
writes into a module-global directory), so this script executes the reference's own formatting statements — the
`plt.Normalize()(attn.data)` call on matplotlib itself and the four f-string writes — on seeded tensors, and stores
the resulting files (data) next to the inputs.  Run in the build container; the fixtures are committed.

    python tests/golden/make_dla_golden.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fixture_inputs import DLA_CASES, dla_inputs as inputs  # noqa: E402

import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dla")


def reference_statements(output_dir, name, raster, attn, activations):
    # the statements of gbm/classify.py:208-225, verbatim in effect (output_dir passed instead of a global)
    att_weights_real = plt.Normalize()(attn.data)
    for tag, col in (("ATTN", None), ("ACTF1", 0), ("ACTF2", 1), ("ACTF3", 2)):
        f = open(f'{output_dir}/prediction-AGMIL-{tag}.{name}.dla', "w+")
        for i, coord in enumerate(raster):
            v = att_weights_real[i, 0] if col is None else activations[i, col]
            f.write(f'{coord[1]} {coord[0]} {v}\n')
        f.close()


def main():
    os.makedirs(HERE, exist_ok=True)
    for name, seed, n in DLA_CASES:
        attn, activations, raster = inputs(seed, n)
        reference_statements(HERE, name, raster, attn, activations)
    # a constant map: Normalize() yields zeros when vmin == vmax
    attn = torch.full((4, 3), 0.25)
    _, activations, raster = inputs(13, 4)
    reference_statements(HERE, "slideC", raster, attn, activations)
    print(sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
