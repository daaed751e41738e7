#!/usr/bin/env python3
"""Golden `.dla` files for the attention-map export (gbm/classify.py:207-225).

The reference module cannot be imported (gbm/classify.py pulls in modules that are not in the repository), but its
`write_map` is self-contained: this script parses the reference file with `ast`, compiles ONLY the `write_map` function
definition, and runs it — the reference's own code, matplotlib's own `plt.Normalize` — on seeded tensors with the
module-global `output_dir` it writes into pointed at tests/golden/dla.  The resulting files (data) are committed; the
reference source is read at generation time only and never stored.  Runs in the build container (needs /root/reference):

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


REFERENCE = "/root/reference/gbm/classify.py"
# sha256 of the reference file whose `write_map` was read (lines 207-225: four formatting statements, no imports, no default
# arguments that evaluate code) before it was first executed here.  The function is executed only when the file still has this
# digest: an upstream snapshot that differs must be re-read by a person first (it runs with this process's rights).
REFERENCE_SHA256 = "595db461dd505c2de48b6dd8e61964cb28f3af970218029b69f8709d07ef5359"


def reference_write_map(output_dir):
    """The reference's own `write_map(meta, epoch, raster, attn, activations)` (gbm/classify.py:207-225), compiled from its
    source file: only that FunctionDef is executed, in a namespace holding what it refers to (`plt`, `output_dir`)."""
    import ast
    import hashlib
    src = open(REFERENCE, "rb").read()
    if hashlib.sha256(src).hexdigest() != REFERENCE_SHA256:
        raise SystemExit(f"{REFERENCE} differs from the vetted snapshot (sha256 {REFERENCE_SHA256}): read write_map again, then update the digest")
    tree = ast.parse(src.decode(), filename=REFERENCE)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "write_map"]
    assert len(fn) == 1, "write_map not found in the reference"
    ns = {"plt": plt, "output_dir": output_dir}
    exec(compile(ast.Module(body=fn, type_ignores=[]), REFERENCE, "exec"), ns)
    return ns["write_map"]


def reference_statements(output_dir, name, raster, attn, activations):
    reference_write_map(output_dir)({"basename": name}, 0, raster, attn, activations)


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
