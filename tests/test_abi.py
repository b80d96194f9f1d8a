"""The C-ABI shared library loads without a GPU and exports every symbol include/derp_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "derp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(derp_[a-z0-9_]+)\s*\(", text)))


def test_exports(built):
    from facebook360_dep_amd import derp

    lib = derp.lib()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libderp_hip.so does not export %s" % n
    assert sorted(derp.EXPORTS) == names, set(derp.EXPORTS) ^ set(names)


def test_no_cpu_fallback(built):
    """Without a HIP device derp_create must fail loudly (no CPU compute path exists)."""
    import torch

    from facebook360_dep_amd import derp, synth

    if torch.cuda.is_available():
        return
    rig = synth.make_rig(4, 64)
    try:
        derp.Derp(rig["cameras"])
    except derp.DerpError as e:
        assert "no HIP device" in str(e) or "gfx950" in str(e)
    else:
        raise AssertionError("derp_create succeeded without a GPU")


def test_product_does_not_touch_oracle():
    """The product package and the CLI sources never import / link the oracle."""
    pkg = os.path.join(ROOT, "facebook360_dep_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "derp_oracle" not in text and "libderp_oracle" not in text, f


def test_restated_libstdcxx_algorithms(built):
    """gcc_algos.h (what the device code runs) vs this image's libstdc++ through the oracle library."""
    import numpy as np

    from facebook360_dep_amd import derp
    from oracle import oracle_lib as O

    rng = np.random.default_rng(1)
    for t in range(4000):
        n = int(rng.integers(1, 33))
        if t % 3 == 0:
            p = rng.integers(0, 5, size=(n, 2)).astype(np.float32)  # many ties
        elif t % 3 == 1:
            p = np.sort(rng.random((n, 2)).astype(np.float32), axis=0)  # sorted input
        else:
            p = rng.random((n, 2)).astype(np.float32)
        nth = max(1, n - 2)
        assert np.array_equal(O.nth_element_pairs(p, nth), derp.host_nth_element_pairs(p, nth)), (n, p)
    for seed in (0, 1, 7, 12345, 2147483647, 2147483646):
        ref = O.minstd_uniform(seed, 64, 0.25, 1.75)
        got = np.array([derp.host_minstd_uniform(seed, i, 0.25, 1.75) for i in range(64)], dtype=np.float32)
        assert np.array_equal(ref, got), seed
    # jump-ahead far into the stream
    ref = O.minstd_uniform(99, 100001, 0.0, 1.0)
    assert ref[100000] == np.float32(derp.host_minstd_uniform(99, 100000, 0.0, 1.0))


def test_block_bias_rounding_identity():
    """k_random_proposals sums the 3x3 colour-bias boxes from the 4x4 texel block in fp32 and rounds with
    trunc((s + 4) * fl(1/9)) (derp_kernels.h, DERP_RANDOM_BLOCK_BIAS); k_reproject_bias, like cv::blur on CV_16UC3
    (DerpUtil.cpp:208-210), rounds the integer sum with (s + 4) / 9. The two agree for EVERY possible sum of nine
    16-bit texels, and every such sum (+ 4) is exact in fp32."""
    import numpy as np

    s = np.arange(0, 9 * 65535 + 1, dtype=np.int64)
    f = s.astype(np.float32)
    assert np.array_equal(f.astype(np.int64), s) and np.array_equal((f + np.float32(4)).astype(np.int64), s + 4)
    q = np.trunc((f + np.float32(4)) * np.float32(1.0 / 9.0))
    assert q.dtype == np.float32 and np.array_equal(q.astype(np.int64), (s + 4) // 9)
