"""The oracle against its committed golden fixture (tests/golden/oracle_tiny.npz, written by
tests/golden/gen_oracle_goldens.py): freezes the restated arithmetic, including the synthetic
input generator, so that neither can drift unnoticed."""
import importlib.util
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_matches_golden():
    spec = importlib.util.spec_from_file_location("gen_oracle_goldens", os.path.join(G, "gen_oracle_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.run()
    ref = np.load(os.path.join(G, "oracle_tiny.npz"))
    assert sorted(ref.files) == sorted(got)
    for k in ref.files:
        a, b = got[k], ref[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype.kind == "f":
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            # transcendental functions come from the host libm; allow a vanishing number of last-ulp flips
            assert (~same).sum() <= 1e-4 * a.size, (k, int((~same).sum()))
        else:
            assert np.array_equal(a, b), k
    # disparities are plausible: inside [1/max_depth, 1/min_depth] wherever defined
    d = ref["plain_l0"]
    v = d[np.isfinite(d)]
    assert v.size > 0.5 * d.size and v.min() > 0 and v.max() <= 2.0 + 1e-3
