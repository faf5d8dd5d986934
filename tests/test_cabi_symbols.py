"""The C-ABI library loads on a CPU-only box and exports every symbol include/hlvae_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hlvae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hlvae_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from hlvae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hlvae_hip.h but not exported"
    # the binding knows all of them too, and the struct layouts agree with the header
    assert set(syms) == set(_lib.EXPORTED_SYMBOLS)
    _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hl-vae_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), fn


def test_cpu_tensors_raise():
    import torch
    from hlvae_amd import synthetic
    from hlvae_amd.HLVAE import HLVAE
    src = synthetic.make_tabular(n_rows=8, T=4, seed=1)
    m = HLVAE([src.cov_dim_ext, [16], 4, [16], 5], src.types_info, src.n_variables, conv=False)
    with pytest.raises(RuntimeError):
        m(torch.tensor(src.data), torch.tensor(src.mask), None, src.types_info)
    with pytest.raises(ValueError):      # conv=True (the default, as in the reference) needs 36 x 36 = 1296 variables
        HLVAE([src.cov_dim_ext, [16], 4, [16], 5], src.types_info, src.n_variables)
    d4 = synthetic.make_d4(n_subjects=1, T=2, seed=0)
    mc = HLVAE([d4.cov_dim_ext, [16], 4, [16], 5], d4.types_info, d4.n_variables)       # convolutional model, CPU tensors
    assert {"conv1.weight", "conv2.bias", "deconv_layer.2.weight", "Decoder_Conv_layer.0.bias",
            "representation_layer.0.weight"} <= set(mc.state_dict())
    with pytest.raises(RuntimeError):
        mc(torch.tensor(d4.data), torch.tensor(d4.mask), None, d4.types_info)
