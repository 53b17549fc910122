"""The device generator of the synthetic workloads (csrc/synth.hip) against its numpy twin (fastqdedup_amd/synth.py),
uniform and skewed model."""
import numpy as np
import pytest

@pytest.mark.gpu
@pytest.mark.parametrize("skew", [False, True])
@pytest.mark.parametrize("L,umi", [(32, 32), (50, 8), (100, 12)])
def test_device_generator_equals_numpy_twin(skew, L, umi):
    import torch
    import fastqdedup_amd as F
    from fastqdedup_amd.synth import SKEW, synth_keys_range
    ctx = F.Context(0)
    n_total, start, count = 3_000_000, 1_234_567, 200_000
    dev = torch.empty(count * L, dtype=torch.uint8, device="cuda:0")
    ctx.synth_keys(dev, n_total, start, count, L, umi, 1003, skew=SKEW if skew else None)
    want = synth_keys_range(n_total, start, count, L, umi, 1003, skew=SKEW if skew else None)
    assert np.array_equal(dev.cpu().numpy().reshape(count, L), want)


def test_skewed_model_has_what_it_promises():
    """A hot key, a crowded segment, a ladder that is one component (host generator alone)."""
    from fastqdedup_amd.synth import SKEW, synth_keys
    n, L = 400_000, 32
    keys = synth_keys(n, L, L, 5, sub_rate=0.0, n_rate=0.0, skew=SKEW)
    u, c = np.unique(keys.view(f"S{L}").ravel(), return_counts=True)
    assert c.max() >= 0.02 * n                                       # the hot key
    seg0 = np.unique(np.array([k[:16] for k in u]), return_counts=True)[1]
    assert seg0.max() >= 0.005 * n                                   # the ladder's keys share segment 0 ...
    assert np.sort(seg0)[-2] >= 50                                   # ... and the poly-A keys theirs
