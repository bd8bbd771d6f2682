"""CPU: torch.ops.mm_native.* registrations — schemas, fake (meta) shapes, and that a CPU tensor has
no kernel to land on (the ops are HIP-only; nothing falls back)."""
import pytest
import torch

import matchmaker_amd.torch_ops  # noqa: F401  (defines torch.ops.mm_native.*)


def test_ops_are_registered_with_the_documented_schemas():
    ns = torch.ops.mm_native
    assert str(ns.maxsim.default._schema) == \
        ("mm_native::maxsim(Tensor q, Tensor d, Tensor? q_mask, Tensor? d_mask, SymInt pairs_per_query=1, "
         "bool sim_round=False, bool sum_round=False) -> Tensor")
    for name, n_args in (("maxsim_inbatch", 7), ("kernel_pool", 11), ("tkl_window_pool", 10)):
        assert len(getattr(ns, name).default._schema.arguments) == n_args
    assert len(ns.tkl_window_pool.default._schema.returns) == 2


def test_fake_tensor_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q, d = torch.empty(4, 32, 128, dtype=torch.bfloat16, device="cuda"), torch.empty(4000, 180, 128, dtype=torch.bfloat16, device="cuda")
        s = torch.ops.mm_native.maxsim(q, d, None, None, 1000)
        assert s.shape == (4000,) and s.dtype == torch.float32
        s = torch.ops.mm_native.maxsim_inbatch(q, None, d[:7], None, False)
        assert s.shape == (4, 7)
        z = torch.empty(11, device="cuda")
        s = torch.ops.mm_native.kernel_pool(q.float(), d.float(), None, None, z, z, z, z, 1000, None, 1e-10)
        assert s.shape == (4000,)
        sc, win = torch.ops.mm_native.tkl_window_pool(q.float(), torch.empty(30, 50, 128, device="cuda"), torch.empty(30, 50, device="cuda"),
                                                      torch.empty(30, dtype=torch.int32, device="cuda"), torch.empty(4, 32, device="cuda"),
                                                      torch.empty(100, device="cuda"), 4, 52, 11, "embedding")
        assert sc.shape == (4,) and win.shape == (4, (52 * 40 - 30) // 2 + 1)


def test_cpu_tensors_have_no_kernel():
    q, d = torch.zeros(1, 4, 8), torch.zeros(1, 5, 8)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.mm_native.maxsim(q, d, None, None, 1)
