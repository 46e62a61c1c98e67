import pytest


def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from flate_amd import default_engine
    return default_engine()
